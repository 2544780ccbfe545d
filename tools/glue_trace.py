"""Attribute the torch-native (at::native) kernels of one eager GAN step to Python call sites.
  python tools/glue_trace.py [c3|c2]   (GPU box)"""
import collections
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
import bench
from data.synthetic import synth_batch
from torch.profiler import profile, ProfilerActivity

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c3'
B = bench.DEFAULT_BATCH[cfg]
runner, conf = bench.build_runner(cfg, 'bf16', B)
dev = torch.device('cuda', 0)
batches = [{k: v.to(dev) for k, v in synth_batch(B, 256, 256, acc=4, seed=i).items()} for i in range(2)]


class Loader(list):
  batch_size = B


runner.train_epoch(Loader(batches * 2), 0)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
  runner.train_epoch(Loader(batches[:1]), 1)
  torch.cuda.synchronize()
agg = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
  if ev.name.startswith('aten::') and ev.device_time_total > 0 and not ev.cpu_children:
    st = [f for f in (ev.stack or []) if 'site-packages' not in f and 'dist-packages' not in f and
          'glue_trace' not in f and '<built-in' not in f]
    key = (ev.name, ' <- '.join(s.split('/')[-1] for s in st[:3]) + ' shape=' + str(ev.input_shapes[:2] if ev.input_shapes else ''))
    agg[key] += 1
    dur[key] += ev.device_time_total
tot = 0
for k, v in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
  print('%3d x %7.1f us  %-22s | %s' % (v, dur[k], k[0], k[1]))
  tot += v
print('total aten launches with device time:', tot)
