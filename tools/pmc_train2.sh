#!/bin/bash
# dev tool (GPU box): HBM traffic counters of the training-step kernels, one pass per counter (--pmc only with --kernel-trace).
# usage: tools/pmc_train2.sh bl6|ref6 B  -> gpurun_out/pmc2_<shape>_b<B>/{FETCH_SIZE,WRITE_SIZE}.csv (per-kernel means)
R=$PWD; S=${1:-bl6}; B=${2:-8}; O=$R/gpurun_out/pmc2_${S}_b$B
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/raw_$c -o p -- python3 $R/tools/time_train.py $B 150 bf16 $S opt > $O/$c.log 2>&1
  f=$(find $O/raw_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$c" > $O/$c.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]:
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
print("Kernel_Name,Dispatches,Counter_Name,Mean_Value,Min_Value,Max_Value")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if "at::native" in k or "rocclr" in k: continue
    print('"%s",%d,%s,%.3f,%.3f,%.3f' % (k, len(v), sys.argv[2], sum(v) / len(v), min(v), max(v)))
PY
  rm -rf $O/raw_$c
  echo "pass $c done"
done
head -12 $O/FETCH_SIZE.csv | cut -c1-170
