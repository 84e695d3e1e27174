#!/bin/bash
# per-kernel average durations of the headline pair for several builds, from rocprofv3 kernel traces: bash tools/prof_variants.sh <lib>...
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/pv_$lib
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv_$lib -o out -- python3 $GRAFT_REPO_ROOT/tools/pair_time.py --one $lib > /tmp/pv_$lib.log 2>&1 || { tail -5 /tmp/pv_$lib.log; exit 1; }
  f=$(find /tmp/pv_$lib -name "*kernel_stats.csv" | head -1)
  echo "== $lib"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    import os
    pat = os.environ.get("PROF_KERNELS", "level_coefs,resize,spline").split(",")
    if any(q in n for q in pat):
        print("   %-50s calls %5s avg %9.1f us" % (n[:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
