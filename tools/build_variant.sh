#!/bin/bash
# dev tool: build a diagnostic variant of the library (never shipped): tools/build_variant.sh NAME -DFLAG [-DFLAG2 ...]
# -> shallow_wavenet_amd/libswn_hip_NAME.so ; select with SWN_HIP_LIB=<path>.  Only the named sources are rebuilt with
# the flags (VSRC, default swn_stack_bf16.hip), the rest is linked from the objects of the regular build.
set -e
name=$1; shift
cd $(dirname $0)/../shallow_wavenet_amd/csrc
VSRC=${VSRC:-swn_stack_bf16.hip}
objs=""
for f in *.o; do
  src=${f%.o}.hip; [ -f "$src" ] || src=${f%.o}.cpp
  if [[ " $VSRC " == *" $src "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-function "$@" -c $src -o /tmp/var_${name}_$f
    objs="$objs /tmp/var_${name}_$f"
  else objs="$objs $f"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libswn_hip_$name.so $objs
echo built ../libswn_hip_$name.so
