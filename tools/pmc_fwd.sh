#!/bin/bash
# dev tool (GPU box): SQ counter passes over the bf16 forward (tools/time_forward.py B Tf); per-kernel means of the layer kernel
# usage: tools/pmc_fwd.sh B Tf [libvariant]  -> gpurun_out/pmc_fwd_<tag>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; B=${1:-64}; TF=${2:-150}; V=${3:-main}
lib=$R/shallow_wavenet_amd/libswn_hip_$V.so; [ "$V" = main ] && lib=$R/shallow_wavenet_amd/libswn_hip.so
export SWN_HIP_LIB=$lib
O=$R/gpurun_out/pmc_fwd_${V}_${B}x${TF}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p$i -- python3 $R/tools/time_forward.py $B $TF > $O/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "bf16_layer" in k or "bf16_stack_fused" in k:
            acc[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = open("$R/gpurun_out/pmc_fwd_${V}_${B}x${TF}.txt", "w")
for k, d in acc.items():
    for c, v in sorted(d.items()):
        line = f"{k} {c} launches={len(v)} mean={sum(v)/len(v):.1f}"
        print(line); out.write(line + "\n")
PY
