#!/bin/bash
# the bench lines of the closing pass alone (after a bench.py-only change): contract test, driver's invocation, default, C2
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04final; mkdir -p $O
python -m pytest tests/test_bench_contract_gpu.py -m gpu -q > $O/gputest_contract.log 2>&1; tail -2 $O/gputest_contract.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_20steps.json 2> $O/bench_n1_20steps.err ) 2> $O/bench_n1_20steps.time
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --config c2 --no-other-configs > $O/bench_c2.json 2>/dev/null
cat $O/bench_n1_20steps.time
