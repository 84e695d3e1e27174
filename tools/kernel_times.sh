#!/bin/bash
# Per-kernel average durations of a short bench run (development aid; run on the GPU box from the repo root).
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/st
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 5 $BENCH_ARGS > /tmp/bench_line.json 2>/dev/null
f=$(ls /tmp/st/*/bench_kernel_stats.csv /tmp/st/bench_kernel_stats.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-62s %5s %9.1f us %6s%%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
cut -c1-190 /tmp/bench_line.json
