"""Per-shape table of the gconv launches in one GAN step (HIP events, eager).
usage: python tools/profile_shapes.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
import bench
from csmri_hip import ops
from data.synthetic import synth_batch

runner, conf = bench.build_runner('c3', 'bf16', 8)
dev = torch.device('cuda', 0)
hb = [synth_batch(8, 256, 256, acc=4, seed=i) for i in range(2)]
runner.train_epoch(bench.PinnedHostLoader(hb, 3, dev, resident=True), 0)
ops.PROFILE_SHAPES = True
ops.PROFILE = []
steps = 3
runner.train_epoch(bench.PinnedHostLoader(hb, steps, dev, resident=True), 1)
torch.cuda.synchronize()
recs, ops.PROFILE = ops.PROFILE, None
agg = {}
for label, flops, e0, e1 in recs:
  a = agg.setdefault(label, [0, 0.0, 0.0])
  a[0] += 1; a[1] += flops; a[2] += e0.elapsed_time(e1) * 1e-3
tot = 0.0
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2]):
  tot += v[2] / steps
  print('%-95s n=%2d %8.1f us/launch %7.1f TF/s %7.3f ms/step' % (
      k, v[0] // steps, v[2] / v[0] * 1e6, v[1] / v[2] / 1e12 if v[2] else 0, v[2] / steps * 1e3))
print('total ms/step', tot * 1e3)
