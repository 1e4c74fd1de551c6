#!/bin/bash
# dev tool (GPU box): the round's closing check - whole GPU suite, smoke, then the unprofiled default bench line
bash tools/full_check.sh > /dev/null 2>&1; tail -2 gpurun_out/s3/gputests.log
bash tools/final_bench.sh > /dev/null
python3 - <<'PY'
import json
b = json.load(open("gpurun_out/final_bench.json"))
print(b["value"], b["us_per_sample_step"], b["roofline"]["frac"])
for k in ("cfg4_bl6_step_bf16", "cfg4_ref6_step_bf16", "cfg4_ref6_fwd_bf16", "cfg4_bl6_step_bf16_dropout", "cfg4_ref6_step_bf16_dropout", "ref6_cfg2", "ref6_cfg5_share"):
    print(k, b["legs"][k])
PY
