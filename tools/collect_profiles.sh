#!/bin/bash
# Runs on the GPU box from the repository root: the collection that tools/make_profiles.py turns into profiles/.
# Usage: bash tools/collect_profiles.sh gpurun_out/<dir>
# Every rocprofv3 run is on its own (counters never share a run with the stats pass; at most 8 SQ counters a pass).
set -e
O=$GRAFT_REPO_ROOT/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
P=$GRAFT_REPO_ROOT/bench_paths.py
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $B --no-cpu-baseline --no-in-flight > $O/bench_under_rocprof.json
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- python3 $B --steps 2 --warmup 1 --no-cpu-baseline --no-in-flight > /dev/null
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- python3 $B --steps 2 --warmup 1 --no-cpu-baseline --no-in-flight > /dev/null
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/sq -o p -- python3 $B --steps 2 --warmup 1 --no-cpu-baseline --no-in-flight > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/paths_stats -o paths -- python3 $P --reps 5 > /dev/null
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/paths_fetch -o p -- python3 $P --reps 2 --cfg3-size 0 > /dev/null
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/paths_write -o p -- python3 $P --reps 2 --cfg3-size 0 > /dev/null
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $O/paths_sq -o p -- python3 $P --reps 2 --cfg3-size 0 > /dev/null
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json
timeout -k 10 300 python3 bench_paths.py --reps 5 > $O/bench_paths.txt
timeout -k 10 200 python3 bench.py --no-cpu-baseline --with-transfer > $O/bench_with_transfer.json
timeout -k 10 300 python3 bench.py --batch 64 --steps 2 --warmup 1 > $O/bench_batch64.json
timeout -k 10 300 python3 bench.py --batch 64 --steps 2 --warmup 1 --in-flight 1 > $O/bench_batch64_one_in_flight.json
timeout -k 10 300 python3 bench.py --batch 64 --steps 2 --warmup 1 --force-dist > $O/bench_batch64_rccl_world1.json
timeout -k 10 200 python3 tools/ls_modes_time.py 2048 1024 4096 > $O/ls_modes.txt
# rocprofv3 may nest its files one directory down (host name): flatten
for d in stats fetch write sq paths_stats paths_fetch paths_write paths_sq; do f=$(ls -d $O/$d/*/ 2>/dev/null | head -1); if [ -n "$f" ]; then mv $f* $O/$d/; fi; done
ls $O
