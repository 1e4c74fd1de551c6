#!/bin/bash
# dev tool (GPU box): REF6 bf16 forward time with the regular library and with diagnostic variants (SWN_HIP_LIB)
echo "== regular"; python3 tools/time_forward_ref6.py 8 150 2>&1 | grep "^bf16:"
for v in "$@"; do
  echo "== $v"; SWN_HIP_LIB=$PWD/shallow_wavenet_amd/libswn_hip_$v.so python3 tools/time_forward_ref6.py 8 150 2>&1 | grep "^bf16:"
done
