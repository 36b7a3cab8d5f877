#!/bin/bash
# What a round runs on the GPU box after the kernels changed, in one gpurun call:
#   gpurun --timeout 3000 -- 'bash tools/prof/gpu_round.sh r02'
# 1. the whole -m gpu suite (log: gpurun_out/<tag>_pytest_gpu.log; tests/test_gpu_fullsize.py also
#    writes gpurun_out/parity_fullsize_*.json),
# 2. the rocprofv3 evidence of the four bench configurations (tools/prof/collect.sh),
# 3. kernel times by ensemble size.
# Back in the container: python tools/prof/summarize.py <tag> 65536x1 65536x4 1024x1 32768x1 131072x1
# turns gpurun_out/ into profiles/ (and profiles/pmc_index.json, which bench.py and
# tests/test_zz_profiles_current.py key on the hash of the kernel sources).
TAG=${1:-r02}
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; tail -4 gpurun_out/${TAG}_pytest_gpu.log
bash tools/prof/collect.sh $TAG "65536 1" "65536 4" "1024 1" "32768 1" "131072 1" > gpurun_out/collect_${TAG}.log 2>&1; tail -c 150 gpurun_out/collect_${TAG}.log
python tools/prof/time_variants.py prod=hector_amd/lib/libhector_amd.so --configs=65536x1,65536x4,1024x1,32768x1,131072x1,262144x1 2>&1 | grep -v "^$\|amdgpu.ids" | tee gpurun_out/${TAG}_sizes.txt
python tools/prof/tracking_times.py 2>&1 | grep "^|" | tee gpurun_out/${TAG}_tracking.md
python bench.py --steps 20 --warmup 3 2>/dev/null | grep "^{" | tail -1 > gpurun_out/${TAG}_bench_default_line.json
