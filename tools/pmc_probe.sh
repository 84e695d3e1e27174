#!/bin/bash
# one SQ counter pass over the headline pair's kernels (measurement helper): bash tools/pmc_probe.sh <lib> <counters...>
cd /tmp && export TMPDIR=/tmp
lib=$1; shift
rm -rf /tmp/pmc_$lib
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$lib -o p -- python3 $GRAFT_REPO_ROOT/tools/pair_time.py --one $lib > /tmp/pmc_$lib.log 2>&1 || { tail -5 /tmp/pmc_$lib.log; exit 1; }
f=$(find /tmp/pmc_$lib -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:44]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    d = (k, r["Dispatch_Id"])
    if d not in seen:
        seen.add(d); n[k] += 1
for k in acc:
    if "level" in k or "resize" in k or "ysolve" in k:
        print(k, n[k], " ".join("%s=%.3g" % (c, v / n[k]) for c, v in sorted(acc[k].items())))
PY
