#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_w1; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench_shapes.py -x -q -m gpu -k "unet" > $O/shapes.log 2>&1; tail -12 $O/shapes.log
timeout 600 python tools/bench_conv.py unet32 unet64 unet_cat u32x64 wgrad > $O/bench_conv.log 2>&1; grep -v amdgpu.ids $O/bench_conv.log
