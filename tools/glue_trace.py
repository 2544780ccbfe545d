"""Attribute the torch-native glue ops of one eager GAN step to Python call sites."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
import bench
from data.synthetic import synth_batch
from torch.profiler import profile, ProfilerActivity

runner, conf = bench.build_runner('bf16', 8)
dev = torch.device('cuda', 0)
batches = [{k: v.to(dev) for k, v in synth_batch(8, 256, 256, acc=4, seed=i).items()} for i in range(2)]
runner.train_epoch(bench.DeviceLoader(batches, 3), 0)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
  runner.train_epoch(bench.DeviceLoader(batches, 1), 1)
  torch.cuda.synchronize()
agg = collections.Counter()
names = ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::add', 'aten::add_', 'aten::mul', 'aten::bernoulli_',
         'aten::index_select', 'aten::cat', 'aten::div', 'aten::sum', 'aten::mean', 'aten::mul_', 'aten::sub',
         'aten::index_copy_', 'aten::neg', 'aten::log', 'aten::reciprocal', 'aten::lt', 'aten::ge', 'aten::rsub')
for ev in prof.events():
  if ev.name in names:
    st = [f for f in (ev.stack or []) if 'site-packages' not in f and 'dist-packages' not in f and 'glue_trace' not in f and '<built-in' not in f]
    key = (ev.name, ' <- '.join(s.split('/')[-1] for s in st[:3]) + ' shape=' + str(ev.input_shapes[:2] if ev.input_shapes else ''))
    agg[key] += 1
for k, v in agg.most_common(70):
  print(v, k[0], '|', k[1])
