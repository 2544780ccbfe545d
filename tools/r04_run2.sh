#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python tools/h2d_numa_probe.py > gpurun_out/r04/h2d_numa_probe.log 2>&1
python -m pytest tests -m gpu -q -x > gpurun_out/r04/gputest_2.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/gputest_2.log
python bench.py --no-cpu-baseline > gpurun_out/r04/bench_n1_b.json 2> gpurun_out/r04/bench_n1_b.err
tail -8 gpurun_out/r04/gputest_2.log; cat gpurun_out/r04/h2d_numa_probe.log
