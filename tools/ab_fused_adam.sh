#!/bin/bash
# dev tool (GPU box): the training step with torch's foreach Adam and with fused=True
for s in "bl6" "ref6"; do
  for f in "" "fused_adam" "" "fused_adam"; do
    echo -n "[$s ${f:-foreach}] "; python3 tools/time_train.py 8 150 bf16 $s opt $f | cut -c1-110
  done
done
