#!/bin/bash
# dev tool (GPU box): HBM traffic counters of the headline decode launch alone, separate passes.  -> gpurun_out/pmc_headline/
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/pmc_headline; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-legs --no-cfg5 --no-cpu-baseline --detail "" > $O/$c.log 2>&1
done
for s in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS"; do
  t=$(echo $s | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $s --output-format csv -d $O/$t -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-legs --no-cfg5 --no-cpu-baseline --detail "" > $O/$t.log 2>&1
done
python3 - <<PY
import csv, glob, collections, os
O = "$O"
out = open(os.path.join(O, "pmc_summary.csv"), "w")
out.write("pass,kernel,counter,launches,mean_per_launch\n")
for d in sorted(glob.glob(O + "/*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        if "decode" in k: out.write(f"{os.path.basename(d)},{k},{c},{len(v)},{sum(v)/len(v):.1f}\n")
out.close()
print(open(os.path.join(O, "pmc_summary.csv")).read())
PY
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
