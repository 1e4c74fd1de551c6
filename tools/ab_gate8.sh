#!/bin/bash
# dev tool (GPU box): parity + timing of the REF6 forward after a change to the gated-layer kernels
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/gate8_${1:-x}; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_bf16_stack.py tests/test_gpu_ref6_teacher_forced.py tests/test_gpu_dropout_parity.py tests/test_gpu_train_bf16.py -x -q -m gpu > $O/tests.log 2>&1
rc=$?; tail -4 $O/tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python3 tools/time_forward_ref6.py 8 150 2>&1 | tail -3 | tee $O/fwd.txt
timeout -k 10 120 python3 tools/time_train.py 8 150 bf16 ref6 opt | tee $O/plain.txt
timeout -k 10 120 python3 tools/time_train.py 8 150 bf16 ref6 opt drop | tee $O/drop.txt
bash tools/prof_fwd_ref6.sh | head -8
