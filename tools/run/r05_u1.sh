#!/bin/bash
# uconv first light: parity at the bench shapes, micro-benchmarks, bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_u1; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench_shapes.py -x -q -m gpu > $O/shapes.log 2>&1; tail -15 $O/shapes.log
timeout 600 python tools/bench_conv.py unet32 unet64 unet_cat u32x64 vgg1_2 fwd dgrad > $O/bench_conv.log 2>&1; cat $O/bench_conv.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_u1/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,v in sorted(d['conv_kernels'].items(),key=lambda kv:-kv[1]['ms_per_step'])[:24]:
    print(f"{k:50s} n={v['launches_per_step']:3d} GF={v['gflop_per_step']:8.2f} ms={v['ms_per_step']:.4f} TF={v['tflops']:.1f}")
PY
