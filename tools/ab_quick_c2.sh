#!/bin/bash
mkdir -p gpurun_out
for i in 1 2; do python bench.py --config c2 --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-other-configs --settle-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', d['value'], d['ms_per_step'])"; done
