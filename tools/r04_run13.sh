#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_bench_shapes.py -m gpu -q -x -k "vgg" 2>&1 | tail -2
for a in "128 128 128 16" "256 256 64 16" "512 512 32 16"; do python tools/stamp_pconv2.py $a 2>&1 | tail -13 | head -7; done
echo NEW; python tools/bench_conv.py vgg2_1b16 vgg2_2b16 vgg3_1b16 vgg3_2b16 vgg4_1b16 vgg4_2b16 2>&1 | tail -8
echo PREV; (cd ab/prev && python tools/bench_conv.py vgg2_1b16 vgg2_2b16 vgg3_1b16 vgg3_2b16 vgg4_1b16 vgg4_2b16 2>&1 | tail -8)
