#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline object, on the GPU box:
#   gpurun --timeout 1500 -- 'bash tools/prof/collect.sh r02 "65536 1" "65536 4" "1024 1"'
# For every "<members> <biomes>" configuration: kernel-trace statistics of the bench command and
# five PMC passes (FETCH_SIZE and WRITE_SIZE apart -- they do not fit one pass -- and three SQ
# passes; never together with any --sys/--runtime trace).  Raw output goes to
# gpurun_out/<tag>_<members>x<biomes>/; tools/prof/summarize.py turns it into profiles/.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  set -- $cfg; M=$1; B=$2
  D=$R/gpurun_out/${TAG}_${M}x${B}; mkdir -p $D
  ARGS="--members $M --biomes $B --no-cpu-baseline --no-other-configs --sustain 0"
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -o s -- python $R/bench.py --steps 5 --warmup 1 $ARGS > $D/bench_under_rocprof.json 2> $D/stats.err
  one() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $D/$1 -o p -- python $R/bench.py --steps 1 --warmup 0 $ARGS > /dev/null 2> $D/$1.err; }
  one fetch "FETCH_SIZE"
  one write "WRITE_SIZE"
  one sq_a "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"
  one sq_b "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT"
  one sq_c "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_BRANCH SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES"
  python $R/bench.py --steps 5 --warmup 1 $ARGS > $D/bench.json 2> $D/bench.err
  tail -c 400 $D/bench.json
done
