#!/bin/bash
# dev tool: per-kernel time of the bf16 stack for several builds of the library (SWN_HIP_LIB override)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  export SWN_HIP_LIB=$R/shallow_wavenet_amd/libswn_$v.so
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/var_$v -o p -- python3 $R/tools/time_forward.py 64 150 > $R/gpurun_out/var_$v.log 2>&1
  echo "$v: $(grep bf16_layer $R/gpurun_out/var_$v/p_kernel_stats.csv | cut -d, -f2-7 | tail -1) | $(grep '^bf16 vs' $R/gpurun_out/var_$v.log)"
done
