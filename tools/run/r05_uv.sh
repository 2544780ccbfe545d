#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_uv; mkdir -p $O
export TMPDIR=/tmp
cp gpurun_out_r05_bench_n1.json profiles/r05_bench_n1.json 2>/dev/null
timeout 900 python -m pytest tests/test_bench_shapes.py -x -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log
for lib in libcsmri_hip.so libcsmri_hip_ufirst.so; do
  echo "== $lib"
  CSMRI_HIP_LIB=$PWD/csmri-refinement_amd/csmri_hip/$lib timeout 600 python tools/bench_conv.py vgg2_2b16 vgg3_1b16 vgg3_2b16 vgg4_1b16 vgg4_2b16 vgg2_2 vgg3_2 vgg4_2 fwdb dgradg 2>&1 | grep -v amdgpu.ids
done > $O/bench.log 2>&1; cat $O/bench.log
