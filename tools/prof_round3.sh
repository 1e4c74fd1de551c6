#!/bin/bash
# dev tool (GPU box): the rocprofv3 evidence of round 3, one pass per counter set (never --pmc together with tracing
# domains other than --kernel-trace).  usage: tools/prof_round3.sh  -> gpurun_out/prof_r03/
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/prof_r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ "$1" != "pmc" ]; then
# (1) every leg of the default bench line: per-kernel durations
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err
echo "bench trace done"
fi
# (2) headline decode kernel: HBM traffic counters, separate passes
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_decode_$c -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-legs --no-cfg5 --no-cpu-baseline --detail "" > $O/pmc_decode_$c.log 2>&1
  echo "pmc decode $c done"
done
# (3) bf16 forward at cfg4's own size and at 8x the batch: kernel times, then HBM traffic of the fused stack kernel
for shape in "8 150" "64 150"; do
  tag=$(echo $shape | tr ' ' 'x')
  [ "$1" != "pmc" ] && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwd_$tag -o f -- python3 $R/tools/time_forward.py $shape > $O/fwd_$tag.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_fwd_${tag}_$c -o p -- python3 $R/tools/time_forward.py $shape > $O/pmc_fwd_${tag}_$c.log 2>&1
  done
  echo "forward $tag done"
done
# (4) stepped decode at the run.sh geometry, 64 utterances (tiles) and 8 (one workgroup per pair and utterance)
for b in 64 8; do
  [ "$1" = "pmc" ] && continue
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ref6_b$b -o d -- python3 $R/tools/time_ref6_batch.py $b > $O/ref6_b$b.log 2>&1
  echo "ref6 decode B=$b done"
done
# the per-dispatch traces of the launch-chain decodes are tens of MB: keep the per-kernel statistics and the counters
python3 - <<PY
import csv, glob, collections, os
O = "$O"
out = open(os.path.join(O, "pmc_summary.csv"), "w")
out.write("pass,kernel,counter,launches,mean_per_launch\n")
for d in sorted(glob.glob(O + "/pmc_*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        out.write(f"{os.path.basename(d)},{k},{c},{len(v)},{sum(v)/len(v):.1f}\n")
out.close()
print(open(os.path.join(O, "pmc_summary.csv")).read()[:3000])
PY
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
du -sh $O
