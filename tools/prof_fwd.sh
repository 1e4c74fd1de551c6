#!/bin/bash
# dev tool: per-kernel times of the bf16 forward at cfg4's size and at 8x the batch (rocprofv3 kernel trace)
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for shape in "8 150" "64 150"; do
  tag=$(echo $shape | tr ' ' 'x')
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fwd_$tag -- python3 $R/tools/time_forward.py $shape > $R/gpurun_out/prof_fwd_$tag.log 2>&1
  f=$(ls $R/gpurun_out/prof_fwd_$tag/*/*kernel_stats.csv | head -1)
  cp $f $R/gpurun_out/prof_fwd_${tag}_kernel_stats.csv
  grep "bf16\|ms / forward" $R/gpurun_out/prof_fwd_$tag.log | tail -3
  head -12 $f | cut -d, -f1-6
done
