R=${GRAFT_REPO_ROOT:-/root/repo}
for v in main old; do
  lib=$R/shallow_wavenet_amd/libswn_hip_$v.so; [ "$v" = main ] && lib=$R/shallow_wavenet_amd/libswn_hip.so
  for shape in "4 150" "16 150" "32 150" "64 150"; do
    echo -n "$v $shape: "; SWN_HIP_LIB=$lib python3 $R/tools/time_forward.py $shape 2>&1 | grep "^bf16:"
  done
done
