#!/bin/bash
# Builds a variant of the library with extra compiler flags for one translation unit (measurement builds):
#   tools/build_variant.sh <library name> <file.hip> [flags...]     e.g.  tools/build_variant.sh libofri_x.so ofri_hs_stream.hip -DSL_PF=8
# The other objects are taken from the regular build.
set -e
cd "$(dirname "$0")/../opticalflow-ri_amd/csrc"
make -s
OUT=$1; SRC=$2; shift 2
B=/tmp/ofri_variant_build_$$; mkdir -p $B
for f in *.hip; do
  if [ $f = $SRC ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -fno-slp-vectorize "$@" -c $f -o $B/${f%.hip}.o
  else cp build/${f%.hip}.o $B/; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o ../$OUT $B/*.o
rm -rf $B
