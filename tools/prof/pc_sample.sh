#!/bin/bash
# PC sampling of one configuration's year-loop launch (rocprofv3 beta feature; gfx950: host_trap
# and stochastic).  Aggregated on the box: samples per instruction (code-object offset) ->
# gpurun_out/<tag>_pcs_<method>.json, small enough to travel.
#   tools/prof/pc_sample.sh <tag> <members> <biomes> [lib] [method ...]
tag=$1; n=$2; b=$3; lib=${4:-hector_amd/lib/libhector_amd.so}; shift 4
methods=${@:-host_trap stochastic}
export TMPDIR=/tmp
root=$(pwd)
for m in $methods; do
  out=/tmp/pcs_${tag}_$m
  rm -rf $out
  if [ $m = stochastic ]; then unit=cycles; iv=${PCS_INTERVAL:-1048576}; else unit=time; iv=${PCS_INTERVAL:-1}; fi
  ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 240 rocprofv3 --pc-sampling-beta-enabled \
     --pc-sampling-method $m --pc-sampling-unit $unit --pc-sampling-interval $iv \
     --kernel-trace --output-format csv -d $out -- python tools/prof/one_run.py $n $b $lib 3 \
     > gpurun_out/${tag}_pcs_${m}.log 2>&1
  echo "rc $?" >> gpurun_out/${tag}_pcs_${m}.log
  find $out -type f | head -20 >> gpurun_out/${tag}_pcs_${m}.log
  python3 tools/prof/pc_aggregate.py $out gpurun_out/${tag}_pcs_${m}.json >> gpurun_out/${tag}_pcs_${m}.log 2>&1
done
