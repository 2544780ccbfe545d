#!/bin/bash
# usage (GPU box): tools/timeline.sh -> gpurun_out/timeline.txt (one step of the bench as a per-queue kernel timeline)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -o t -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 12 --warmup 4 --no-other-configs --no-input-ab --settle-s 0 > /dev/null 2>&1
f=$(find $R/gpurun_out/tl -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_timeline.py "$f" > $R/gpurun_out/timeline.txt
NAMELEN=150 python3 $R/tools/trace_timeline.py "$f" 2>/dev/null | head -40 > $R/gpurun_out/timeline_head.txt
# the launches that are not this library's (torch elementwise / fill / cat), with their neighbours
NAMELEN=260 python3 $R/tools/trace_timeline.py "$f" | grep -B1 -A1 "at::native" > $R/gpurun_out/timeline_native.txt
python3 $R/tools/trace_concurrency.py "$f" > $R/gpurun_out/concurrency.txt
python3 $R/tools/trace_critical_path.py "$f" > $R/gpurun_out/critical_path.txt
rm -rf $R/gpurun_out/tl
