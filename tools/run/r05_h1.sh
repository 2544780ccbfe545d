#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_h1; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench_shapes.py -x -q -m gpu -k "unet_head or disc_final" > $O/shapes.log 2>&1; tail -3 $O/shapes.log
bash tools/prof_conv.sh h1 unet_head wgrad; head -6 gpurun_out/pc_h1.csv | cut -c1-140
