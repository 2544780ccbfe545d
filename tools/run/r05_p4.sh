#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_p4; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do timeout 300 python tools/bench_conv.py vgg3_2 vgg3_2b16 vgg2_2 dgrad dgradg 2>&1 | grep -v amdgpu.ids; done > $O/dg.log 2>&1; cat $O/dg.log
