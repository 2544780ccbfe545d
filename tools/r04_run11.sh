#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k; mkdir -p $O
python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "batchnorm or bn or replay" > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
bash tools/timeline.sh; for f in timeline timeline_native concurrency critical_path; do mv gpurun_out/$f.txt $O/$f.txt; done
tail -4 $O/gputest.log; head -3 $O/critical_path.txt; grep bn_small $O/timeline.txt
