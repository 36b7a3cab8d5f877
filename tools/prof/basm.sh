#!/bin/bash
# Device assembly of one biome count's kernels only: tools/prof/basm.sh <biomes> <out.s> [flags]
# (0: the looped kernels) -- then tools/isa_stats.py <out.s> --kernel hx_run_kernel
R=$(cd "$(dirname "$0")/../.." && pwd)
B=$1; OUT=$2; shift; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wall -Wno-unused-function -Wno-pass-failed \
  -mllvm -disable-machine-licm -DHX_MINIMAL_BUILD -DHX_ONLY_B=$B "$@" -I$R/hector_amd/csrc \
  --cuda-device-only -S -o $OUT $R/hector_amd/csrc/hx_kernels.hip 2>&1 | grep -v hip-link
python3 $R/tools/isa_stats.py $OUT --kernel hx_run_kernel
