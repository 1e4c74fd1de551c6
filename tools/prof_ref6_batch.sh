#!/bin/bash
# dev tool: per-kernel times of the stepped decode at the run.sh geometry, B utterances (rocprofv3 kernel trace)
R=${GRAFT_REPO_ROOT:-/root/repo}; B=${1:-64}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ref6_b$B -- python3 $R/tools/time_ref6_batch.py $B > $R/gpurun_out/prof_ref6_b$B.log 2>&1
f=$(ls $R/gpurun_out/prof_ref6_b$B/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/prof_ref6_b${B}_kernel_stats.csv
head -14 $f | cut -d, -f1-4 | cut -c1-150
