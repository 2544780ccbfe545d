#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
python -m pytest tests/test_hip_ops.py tests/test_hip_path.py -m gpu -q -x > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
python bench.py --no-other-configs --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err
bash tools/timeline.sh; for f in timeline timeline_native concurrency critical_path; do mv gpurun_out/$f.txt $O/$f.txt; done
tail -4 $O/gputest.log; cut -c1-300 $O/bench_n1.json; head -3 $O/critical_path.txt
