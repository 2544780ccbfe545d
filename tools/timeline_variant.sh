#!/bin/bash
# usage (GPU box): tools/timeline_variant.sh TAG [bench args...]   (env passes through) -> gpurun_out/timeline_TAG.txt
R=$GRAFT_REPO_ROOT
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tl_$tag
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$tag -o t -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 12 --warmup 4 "$@" > /dev/null 2>&1
f=$(find $R/gpurun_out/tl_$tag -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_timeline.py "$f" > $R/gpurun_out/timeline_$tag.txt
rm -rf $R/gpurun_out/tl_$tag
grep -n "step:\|wpatch_kernel<32, 16, 4>\|wpatch_kernel<8, 32, 4>\|refine_combine_bwd\|adam_dev" $R/gpurun_out/timeline_$tag.txt | sed "s/^/$tag: /"
