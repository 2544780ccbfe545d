#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_prio; mkdir -p $O
export TMPDIR=/tmp
{
for lib in libcsmri_hip_stamps.so libcsmri_hip_abl0p0.so libcsmri_hip_abl0p2.so; do
  echo "=== $lib"
  for a in "32 32 4 256 8" "64 32 4 256 8" "64 64 3 256 16 zero"; do
    UCONV_STAMP_LIB=$lib timeout 120 python tools/stamp_uconv.py $a 2>&1 | grep -v amdgpu.ids | grep -E "uconv_kernel|total|barrier|issuing|waiting"
  done
done
} > $O/prio.log 2>&1
cat $O/prio.log
