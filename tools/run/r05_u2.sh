#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_u2; mkdir -p $O
export TMPDIR=/tmp
{
for a in "32 32 4 256 8" "64 64 4 128 8" "64 32 4 256 8" "32 64 4 256 8" "64 128 4 128 8" "64 64 3 256 16 zero" "64 64 4 128 8 reflection stats" "32 32 4 256 8 reflection stats"; do
  timeout 120 python tools/stamp_uconv.py $a
done
} > $O/stamps.log 2>&1
grep -v amdgpu.ids $O/stamps.log
