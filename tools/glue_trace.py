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
rows = [e for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=8)
        if (e.key.startswith('aten::') or 'Memcpy' in e.key or 'Memset' in e.key) and e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
tot_n = tot_t = 0
for e in rows:
  st = [f for f in (e.stack or []) if 'site-packages' not in f and 'dist-packages' not in f and 'glue_trace' not in f
        and '<built-in' not in f]
  print('%3d x %8.1f us  %-28s %-40s | %s' % (e.count, e.self_device_time_total, e.key, str(e.input_shapes)[:40],
                                            ' <- '.join(x.split('/')[-1] for x in st[:4])))
  tot_n += e.count
  tot_t += e.self_device_time_total
print('total aten launches with device time: %d, %.1f us' % (tot_n, tot_t))
