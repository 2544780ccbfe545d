#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_h2; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench_shapes.py -x -q -m gpu -k "disc1 or unet_e0a or recnet" > $O/shapes.log 2>&1; tail -3 $O/shapes.log
bash tools/prof_conv.sh h2 disc_first wgrad; head -5 gpurun_out/pc_h2.csv | cut -c1-140
