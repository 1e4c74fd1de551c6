mkdir -p gpurun_out/s3
python -m pytest tests -x -q -m gpu > gpurun_out/s3/gputests.log 2>&1; rc=$?; tail -5 gpurun_out/s3/gputests.log; [ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/s3/bench.json 2> gpurun_out/s3/bench.err; tail -c 4000 gpurun_out/s3/bench.json
