#!/bin/bash
# dev tool (GPU box): per-kernel statistics of the BL6 training step.  usage: tools/prof_train.sh TAG B [chain]
R=$PWD; T=${1:-x}; B=${2:-8}; O=$R/gpurun_out/prof_train_$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 $R/tools/time_train.py $B 150 bf16 bl6 opt $3 > $O/log.txt 2>&1
find $O -name "*_kernel_trace.csv" -delete
f=$(find $O -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 $f | head -${4:-24}
tail -1 $O/log.txt
