#!/bin/bash
# Experiment build of the library: tools/prof/build_variant.sh <name> [extra hipcc flags...]
# -> gpuwork/lib<name>.so with only the plain B=1 / B=4 run kernels (HX_MINIMAL_BUILD, ~4x faster
# to compile).  gpuwork/ is git-ignored but travels to the GPU box; load it with
# hector_amd.Core(..., lib_path=...).  Not a product path.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; shift
SRC=${HX_SRC_DIR:-$R/hector_amd/csrc}
mkdir -p $R/gpuwork
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed -mllvm -disable-machine-licm -DHX_MINIMAL_BUILD "$@" \
  -I$SRC -shared -o $R/gpuwork/lib$NAME.so $SRC/hx_kernels.hip $SRC/ensemble_core.cpp $(ls $SRC/hx_fleet.cpp 2>/dev/null) $SRC/hx_scenario.cpp $SRC/hx_abi.cpp -ldl
