#!/bin/bash
# usage (GPU box): tools/prof_any.sh <tag> <script.py> [args]  -> gpurun_out/pa_<tag>.csv per-kernel stats
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pa_$tag -o r01 -- python3 $R/"$@" > $R/gpurun_out/pa_$tag.log 2>&1
f=$(find $R/gpurun_out/pa_$tag -name "*kernel_trace.csv" | head -1)
cp $f $R/gpurun_out/pa_$tag.trace.csv
rm -rf $R/gpurun_out/pa_$tag
