#!/bin/bash
# dev tool (GPU box): the rocprofv3 evidence of a round, one pass per counter set (never --pmc together with tracing
# domains other than --kernel-trace).  usage: tools/prof_round.sh r02  -> gpurun_out/prof_<tag>/
R=$PWD; T=${1:-r02}; O=$R/gpurun_out/prof_$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# (1) every leg of the default bench line: per-kernel durations
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.log 2>&1
echo "bench trace done"
# (2) headline decode kernel: HBM traffic counters, separate passes
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-legs --no-cpu-baseline > $O/pmc_$c.log 2>&1
  echo "pmc $c done"
done
# (3) training step incl. Adam + re-pack, BL6 and REF6, mixed precision
for s in bl6 ref6; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_$s -o t -- python3 $R/tools/time_train.py 8 150 bf16 $s opt > $O/train_$s.log 2>&1
  echo "train $s done"
done
# (4) REF6 decode: stepped vs cluster
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ref6 -o d -- python3 $R/tools/try_cluster.py time > $O/ref6.log 2>&1
echo "ref6 decode done"
# the per-dispatch traces of the launch-chain decodes are tens of MB: keep the per-kernel statistics and the counters
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*.csv" | head -40; du -sh $O
