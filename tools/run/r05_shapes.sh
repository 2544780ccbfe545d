#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_shapes; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/profile_shapes.py > $O/shapes.log 2>&1
grep -v amdgpu.ids $O/shapes.log | cut -c1-175 | head -90
