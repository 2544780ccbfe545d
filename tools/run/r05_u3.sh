#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_u3; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bench_shapes.py -x -q -m gpu > $O/shapes.log 2>&1; tail -3 $O/shapes.log
timeout 600 python tools/bench_conv.py unet32 unet64 unet_cat u32x64 vgg1_2 fwd dgrad > $O/bench_conv.log 2>&1; grep -v amdgpu.ids $O/bench_conv.log
{
for a in "32 32 4 256 8" "64 64 4 128 8" "64 32 4 256 8" "64 64 3 256 16 zero"; do
  timeout 120 python tools/stamp_uconv.py $a
done
} > $O/stamps.log 2>&1
grep -v amdgpu.ids $O/stamps.log
