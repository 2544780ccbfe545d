#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_uf; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
for lib in libcsmri_hip.so libcsmri_hip_ufirst.so; do
  echo "== $lib"
  CSMRI_HIP_LIB=$PWD/csmri-refinement_amd/csmri_hip/$lib timeout 300 python tools/bench_conv.py vgg2_1b16 vgg2_1 fwd fwdb dgrad 2>&1 | grep -v amdgpu.ids
done; done > $O/uf.log 2>&1
cat $O/uf.log
