#!/bin/bash
# last check of the tree as committed: full gpu suite, smoke, the driver's bench invocation
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04verify; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_20steps.json 2> $O/bench.err
tail -3 $O/gputest.log; tail -2 $O/smoke.log; cut -c1-330 $O/bench_n1_20steps.json
