#!/bin/bash
# Device assembly of the plain one-biome kernels only (hx_run_kernel<1,0,0,0> and the
# two-wavefront flavour <HX_B1W2,0,0,0>): tools/prof/w2asm.sh <out.s> [extra flags] -- ~25 s,
# then tools/isa_stats.py <out.s> --kernel Li101E
R=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wall -Wno-unused-function -Wno-pass-failed \
  -mllvm -disable-machine-licm -DHX_MINIMAL_BUILD -DHX_W2_ONLY "$@" -I$R/hector_amd/csrc \
  --cuda-device-only -S -o $OUT $R/hector_amd/csrc/hx_kernels.hip 2>&1 | grep -v hip-link
python3 $R/tools/isa_stats.py $OUT --kernel hx_run_kernel
