#!/bin/bash
# A/B on one box: fused un-pooling passes and the number of weight-gradient side streams (bench.py headline config)
mkdir -p gpurun_out
run() { echo "== $1"; shift; env "$@" python bench.py --steps 100 --warmup 15 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
{
run "baseline (no fused pool passes)" CSMRI_NO_POOL_ACT_FUSED=1
run "fused pool passes" X=1
run "fused + 2 wgrad streams" CSMRI_WGRAD_STREAMS=2
run "fused + 3 wgrad streams" CSMRI_WGRAD_STREAMS=3
run "baseline again" CSMRI_NO_POOL_ACT_FUSED=1
run "fused again" X=1
run "fused + 2 streams again" CSMRI_WGRAD_STREAMS=2
} 2>&1 | tee gpurun_out/ab_pool_streams.log
