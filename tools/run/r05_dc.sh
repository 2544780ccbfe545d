#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_dc; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
for lib in libcsmri_hip.so libcsmri_hip_dc8.so; do
  echo "== $lib"
  CSMRI_HIP_LIB=$PWD/csmri-refinement_amd/csmri_hip/$lib timeout 300 python tools/bench_dc.py 2>&1 | grep -v amdgpu.ids | grep float32
done; done > $O/bench_dc_ab.log 2>&1
cat $O/bench_dc_ab.log
