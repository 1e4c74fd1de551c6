#!/bin/bash
# dev tool (GPU box): REF6 training-step parity tests, then the unprofiled step times (plain and dropout) and the per-kernel
# statistics of both.  usage: tools/ab_ref6.sh TAG [notest]
R=$PWD; T=${1:-x}; O=$R/gpurun_out/ab_ref6_$T; mkdir -p $O
if [ "$2" != "notest" ]; then
  python -m pytest tests/test_gpu_ref6_teacher_forced.py tests/test_gpu_dropout_parity.py tests/test_gpu_train_bf16.py tests/test_gpu_backward_parity.py -x -q -m gpu > $O/tests.log 2>&1
  rc=$?; tail -3 $O/tests.log; [ $rc -ne 0 ] && exit $rc
fi
python3 tools/time_train.py 8 150 bf16 ref6 opt | tee $O/plain.txt
python3 tools/time_train.py 8 150 bf16 ref6 opt drop | tee $O/drop.txt
cd /tmp && export TMPDIR=/tmp
for m in plain drop; do
  x=""; [ $m = drop ] && x=drop
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o t -- python3 $R/tools/time_train.py 8 150 bf16 ref6 opt $x > $O/prof_$m.log 2>&1
  find $O/prof_$m -name "*_kernel_trace.csv" -delete
  f=$(find $O/prof_$m -name "*kernel_stats.csv" | head -1)
  echo "== $m"; cut -d, -f1-4 $f | head -14 | sed 's/(anonymous namespace):://g' | cut -c1-150
done
