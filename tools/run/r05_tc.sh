#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_tc; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_bench_shapes.py tests/test_hip_ops.py -x -q -m gpu --deselect tests/test_bench_shapes.py::test_dispatch_variants_are_all_exercised > $O/tests.log 2>&1; tail -3 $O/tests.log
for lib in new r04; do
  echo "== $lib"
  d=$PWD; [ $lib = r04 ] && d=$PWD/ab/r04
  (cd $d && timeout 300 python tools/bench_conv.py vgg1_1 disc_first unet_first rec_first unet_head fwd fwdb 2>&1 | grep -v amdgpu.ids)
done > $O/bench.log 2>&1; cat $O/bench.log
