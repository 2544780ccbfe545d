#!/bin/bash
# usage (on the GPU box): tools/prof.sh <tag> [bench args]; writes gpurun_out/prof_<tag>.csv
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o r01 -- python3 $R/bench.py --no-cpu-baseline --no-roofline "$@" > $R/gpurun_out/prof_$tag.log 2>&1
python3 $R/tools/rocpd_stats.py $R/gpurun_out/prof_$tag/r01_results.db $R/gpurun_out/prof_$tag.csv
rm -rf $R/gpurun_out/prof_$tag
