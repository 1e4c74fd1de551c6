#!/bin/bash
# dev tool (GPU box): per-kernel statistics of the dropout-mode training step (do_prob 0.5, as run.sh trains).
# usage: tools/prof_drop.sh TAG bl6|ref6 [B]
R=$PWD; T=${1:-x}; NET=${2:-bl6}; B=${3:-8}; O=$R/gpurun_out/prof_drop_${NET}_$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 $R/tools/time_train.py $B 150 bf16 $NET opt drop > $O/log.txt 2>&1
find $O -name "*_kernel_trace.csv" -delete
f=$(find $O -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 $f | head -${4:-30}
tail -1 $O/log.txt
