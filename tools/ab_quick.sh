#!/bin/bash
# quick same-box bench of the headline config under a few env settings: tools/ab_quick.sh "NAME=VAL ..." ...
mkdir -p gpurun_out
run() { echo "== $*"; env "$@" python bench.py --steps 100 --warmup 15 --no-cpu-baseline --no-roofline --no-other-configs --settle-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
{
for cfg in "$@"; do run $cfg; done
} 2>&1 | tee gpurun_out/ab_quick.log
