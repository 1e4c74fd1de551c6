#!/bin/bash
# dev tool (GPU box): per-kernel durations of the REF6 teacher-forced forward (fp32 parity kernels and the bf16 GEMM stack)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/fwd_ref6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o f -- python3 $R/tools/time_forward_ref6.py 8 150 > $O/run.log 2>&1
find $O -name "*kernel_trace.csv" -delete
f=$(find $O -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -d, -f1-4 "$f" | sed "s/(anonymous namespace):://g" | cut -c1-110 | head -14
tail -3 $O/run.log
