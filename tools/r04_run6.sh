#!/bin/bash
# full gpu suite + the bench line + profiles of the round-4 build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -q > gpurun_out/r04/gputest_7.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/gputest_7.log
python bench.py > gpurun_out/r04/bench_n1_c.json 2> gpurun_out/r04/bench_n1_c.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_n1_20steps.json 2>/dev/null
python bench.py --config c2 --no-other-configs > gpurun_out/r04/bench_c2.json 2>/dev/null
tail -4 gpurun_out/r04/gputest_7.log
