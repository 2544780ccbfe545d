#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
python tools/native_ops.py c3 > $O/native_ops.txt 2> $O/native_ops.err
python -m pytest tests/test_hip_path.py tests/test_hip_ops.py tests/test_bench_shapes.py tests/test_distributed_gpu.py -m gpu -q -x > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
python bench.py --no-other-configs --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err
bash tools/timeline.sh; for f in timeline timeline_native concurrency critical_path; do mv gpurun_out/$f.txt $O/$f.txt; done
tail -4 $O/gputest.log; cut -c1-400 $O/bench_n1.json; tail -5 $O/native_ops.err; cat $O/native_ops.txt
