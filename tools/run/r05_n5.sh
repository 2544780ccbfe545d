#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_n5; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench_shapes.py -x -q -m gpu -k unet > $O/shapes.log 2>&1; tail -3 $O/shapes.log
timeout 600 python tools/bench_conv.py u128 d1cat d1up wgrad > $O/bench_conv.log 2>&1; grep -v amdgpu.ids $O/bench_conv.log
