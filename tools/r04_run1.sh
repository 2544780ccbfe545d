#!/bin/bash
# round-4 first GPU pass: full gpu suite, default bench line, C2 trajectory ensembles
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputest_1.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/gputest_1.log
python bench.py > gpurun_out/r04/bench_n1_a.json 2> gpurun_out/r04/bench_n1_a.err
python tools/trajectory.py --config c2 --steps 600 --dtypes fp32,bf16 --ensemble 4 --out gpurun_out/r04/traj_c2_600_ens4.json > gpurun_out/r04/traj_c2_600_ens4.log 2>&1
python tools/trajectory.py --config c2 --steps 2400 --dtypes fp32,bf16 --ensemble 2 --out gpurun_out/r04/traj_c2_2400_ens2.json > gpurun_out/r04/traj_c2_2400_ens2.log 2>&1
tail -5 gpurun_out/r04/gputest_1.log
