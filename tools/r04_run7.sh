#!/bin/bash
# fan-in fusion check: gpu suite, bench line, kernel stats (counts at::native launches), step critical chain
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
python bench.py --no-other-configs > $O/bench_n1.json 2> $O/bench_n1.err
bash tools/prof.sh r04g --steps 40 --settle-s 0 --no-other-configs --no-input-ab; mv gpurun_out/prof_r04g.csv $O/kernel_stats.csv
bash tools/timeline.sh; mv gpurun_out/timeline.txt $O/timeline.txt; mv gpurun_out/concurrency.txt $O/concurrency.txt; mv gpurun_out/critical_path.txt $O/critical_chain.txt
tail -5 $O/gputest.log; cat $O/bench_n1.json
