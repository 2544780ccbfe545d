#!/bin/bash
# A/B on one box: where the VGG branch's backward is issued (bench.py headline config)
mkdir -p gpurun_out
run() { echo "== $1"; shift; env "$@" python bench.py --steps 100 --warmup 15 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
{
run "late (inside the generator backward)" CSMRI_VGG_BWD=late
run "early (end of segment 1)" CSMRI_VGG_BWD=early
run "seg3 (first in segment 3, joined by a hook)" CSMRI_VGG_BWD=seg3
run "late again" CSMRI_VGG_BWD=late
run "seg3 again" CSMRI_VGG_BWD=seg3
} 2>&1 | tee gpurun_out/ab_vgg_bwd.log
