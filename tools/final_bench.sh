#!/bin/bash
# dev tool (GPU box): the unprofiled default bench line of the final build -> gpurun_out/final_bench.json (+ detail)
mkdir -p gpurun_out
python bench.py --detail gpurun_out/final_detail.json > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
tail -c 1500 gpurun_out/final_bench.json
