#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_c5; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench_shapes.py -x -q -m gpu -k "vgg5 or vgg4" > $O/tests.log 2>&1; tail -3 $O/tests.log
for lib in new r04; do
  echo "== $lib"
  d=$PWD; [ $lib = r04 ] && d=$PWD/ab/r04
  (cd $d && timeout 300 python tools/bench_conv.py vgg5_2b16 vgg5_2 vgg4_2 fwdb dgradg 2>&1 | grep -v amdgpu.ids)
done > $O/bench.log 2>&1; cat $O/bench.log
