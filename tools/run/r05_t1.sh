#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_t1; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_bench_contract_gpu.py tests/test_hip_ops.py -x -q -m gpu -k "two_ranks or split" > $O/tests.log 2>&1; tail -5 $O/tests.log
bash tools/timeline.sh; head -60 gpurun_out/critical_path.txt | cut -c1-180; head -30 gpurun_out/concurrency.txt | cut -c1-180
