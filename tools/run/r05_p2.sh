#!/bin/bash
# pconv2 / uconv / tconv epilogue cost: forward with and without bias + ReLU, data gradient with and without the gate
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_p2; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/bench_conv.py vgg1_2 vgg2_1b16 vgg2_2b16 vgg3_1b16 vgg3_2b16 vgg4_1b16 vgg4_2b16 fwd fwdb > $O/fwd.log 2>&1; grep -v amdgpu.ids $O/fwd.log
timeout 900 python tools/bench_conv.py vgg1_2 vgg2_1 vgg2_2 vgg3_2 vgg4_2 dgrad dgradg > $O/dgrad.log 2>&1; grep -v amdgpu.ids $O/dgrad.log
