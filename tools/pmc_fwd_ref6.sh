#!/bin/bash
# dev tool (GPU box): counters of the REF6 bf16 forward's kernels (separate passes, --kernel-trace only): HBM traffic and MFMA busy
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/pmc_fwd_ref6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $c | tr ' ' '+')
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$tag -o p -- python3 $R/tools/time_forward_ref6.py 8 150 > $O/$tag.log 2>&1
  echo "pass $tag done"
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
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
            if "bf16g" in k: acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        out.write(f"{os.path.basename(d)},{k},{c},{len(v)},{sum(v)/len(v):.1f}\n")
out.close()
print(open(os.path.join(O, "pmc_summary.csv")).read())
PY
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
