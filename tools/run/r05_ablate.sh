#!/bin/bash
# uconv ablations (stamps builds): 1 no epilogue, 2 no DMA, 4 no LDS reads in the loop
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_ablate; mkdir -p $O
export TMPDIR=/tmp
{
for lib in libcsmri_hip_stamps.so libcsmri_hip_abl1.so libcsmri_hip_abl2.so libcsmri_hip_abl4.so libcsmri_hip_abl6.so libcsmri_hip_abl7.so; do
  echo "=== $lib"
  for a in "64 64 4 128 8" "32 32 4 256 8"; do
    UCONV_STAMP_LIB=$lib timeout 120 python tools/stamp_uconv.py $a 2>&1 | grep -v amdgpu.ids | grep -E "uconv_kernel|compute"
  done
done
} > $O/ablate.log 2>&1
cat $O/ablate.log
