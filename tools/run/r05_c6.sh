#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_c6; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
for lib in libcsmri_hip.so libcsmri_hip_bn256.so; do
  echo "== $lib"
  CSMRI_HIP_LIB=$PWD/csmri-refinement_amd/csmri_hip/$lib timeout 300 python tools/bench_conv.py vgg4_2 vgg4_2b16 vgg3_2 fwdb dgradg 2>&1 | grep -v amdgpu.ids
done; done > $O/bench.log 2>&1; cat $O/bench.log
