#!/bin/bash
# Builds the library of a git revision (default HEAD) beside the working tree's, as cortex_amd/lib/libcortex_hip_<name>.so (default base), for
# same-box A/B runs (CORTEX_HIP_LIB=... selects it).  Not part of the product.
set -e
REV=${1:-HEAD}; NAME=${2:-base}; W=/tmp/cx_base_build; rm -rf $W; mkdir -p $W
cd "$(dirname "$0")/.."
git archive $REV cortex_amd/csrc include | tar -x -C $W
make -C $W/cortex_amd/csrc -j8 > $W/build.log 2>&1 || { tail -20 $W/build.log; exit 1; }
cp $W/cortex_amd/lib/libcortex_hip.so cortex_amd/lib/libcortex_hip_$NAME.so
ls -la cortex_amd/lib/
