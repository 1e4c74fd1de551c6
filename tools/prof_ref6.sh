#!/bin/bash
# dev tool (GPU box): per-kernel statistics of the REF6 mixed-precision training step.  usage: tools/prof_ref6.sh TAG [extra time_train flags]
R=$PWD; T=${1:-x}; O=$R/gpurun_out/prof_ref6_$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 $R/tools/time_train.py 8 150 bf16 ref6 opt $2 $3 > $O/log.txt 2>&1
find $O -name "*_kernel_trace.csv" -delete
f=$(find $O -name "*kernel_stats.csv" | head -1)
python3 $R/tools/kstats.py $f 7 14
tail -1 $O/log.txt
