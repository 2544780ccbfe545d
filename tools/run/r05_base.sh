#!/bin/bash
# round-5 baseline on one box: tconv / wpatch stamps + micro-benchmarks + the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_base; mkdir -p $O
export TMPDIR=/tmp
{
for a in "32 32 4 256 8" "64 64 4 128 8" "64 32 4 256 8" "32 64 4 256 8" "64 128 4 128 8" "64 64 3 256 16 zero"; do
  timeout 120 python tools/stamp_tconv.py $a
done
} > $O/stamps.log 2>&1
timeout 600 python tools/bench_conv.py unet32 unet64 unet_cat u32x64 vgg1_2 fwd dgrad wgrad > $O/bench_conv.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -3 $O/stamps.log; cat $O/bench_conv.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_base/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
PY
