#!/bin/bash
# Builds opticalflow-ri_amd/libofri_trace.so: the library with the Horn-Schunck pipeline's workgroup trace compiled in
# (tools/hs_wg_trace.py).  First argument: library name (default libofri_trace.so).  Extra flags for the pipeline's translation unit may follow, e.g. -DSOMETHING=1.
set -e
cd "$(dirname "$0")/../opticalflow-ri_amd/csrc"
make -s
OUT=${1:-libofri_trace.so}; [ $# -gt 0 ] && shift
B=/tmp/ofri_trace_build_$$; mkdir -p $B
for f in *.hip; do
  if [ $f = ofri_hs_stream.hip ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -fno-slp-vectorize -DOFRI_SL_TRACE "$@" -c $f -o $B/${f%.hip}.o
  else cp build/${f%.hip}.o $B/; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o ../$OUT $B/*.o
