#!/bin/bash
# Runs on the GPU box from the repository root: the collection that tools/make_profiles.py turns into profiles/.
# Usage: bash tools/collect_profiles.sh gpurun_out/<dir>
set -e
O=$GRAFT_REPO_ROOT/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $B --no-cpu-baseline > $O/bench_under_rocprof.json
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 $B --steps 2 --warmup 1 --no-cpu-baseline > /dev/null
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 $B --steps 2 --warmup 1 --no-cpu-baseline > /dev/null
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py > $O/bench_default.json
timeout -k 10 300 python3 bench_paths.py > $O/bench_paths.txt
timeout -k 10 200 python3 bench.py --no-cpu-baseline --with-transfer > $O/bench_with_transfer.json
# rocprofv3 nests its files one directory down (host name): flatten
for d in stats fetch write; do f=$(ls -d $O/$d/*/ 2>/dev/null | head -1); if [ -n "$f" ]; then mv $f* $O/$d/; fi; done
ls $O $O/stats
