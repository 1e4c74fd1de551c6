#!/bin/bash
# dev tool (GPU box): rocprofv3 passes over one mixed-precision training step shape; counters in separate passes
# usage: tools/pmc_train.sh ref6|bl6   -> gpurun_out/pmc_train_<shape>/
set -e
R=$PWD; S=${1:-ref6}; O=$R/gpurun_out/pmc_train_$S
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/time_train.py 8 150 bf16 $S > $O.trace.log 2>&1
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -- python3 $R/tools/time_train.py 8 150 bf16 $S > $O.$n.log 2>&1
  echo "pass $n done"
done
