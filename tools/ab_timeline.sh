#!/bin/bash
# one-step kernel timelines of the builds under ab/<name> and of the working tree ("new"), on the same box
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@" new; do
  d=$R/ab/$v; [ $v = new ] && d=$R
  rm -rf $O/tl
  rocprofv3 --kernel-trace --output-format csv -d $O/tl -o t -- python3 $d/bench.py --no-cpu-baseline --no-roofline --steps 12 --warmup 4 --no-other-configs --no-input-ab --settle-s 0 > /dev/null 2>&1
  f=$(find $O/tl -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/trace_timeline.py "$f" > $O/timeline_$v.txt
  python3 $R/tools/trace_concurrency.py "$f" > $O/concurrency_$v.txt
  python3 $R/tools/trace_critical_path.py "$f" > $O/critical_$v.txt
  rm -rf $O/tl
done
head -3 $O/critical_*.txt
