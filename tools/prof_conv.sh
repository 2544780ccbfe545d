#!/bin/bash
# usage (GPU box): tools/prof_conv.sh <tag> <bench_conv args...>  -> gpurun_out/pc_<tag>.csv (per-kernel stats)
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pc_$tag -o r01 -- python3 $R/tools/bench_conv.py "$@" > $R/gpurun_out/pc_$tag.log 2>&1
python3 $R/tools/rocpd_stats.py $R/gpurun_out/pc_$tag/r01_results.db $R/gpurun_out/pc_$tag.csv
rm -rf $R/gpurun_out/pc_$tag
