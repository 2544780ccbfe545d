"""Which Python lines of the training step still launch torch's own kernels (aten elementwise / fill / cat / foreach)?
Runs the bench's C3 step eagerly under torch.profiler (CPU activity, with_stack) and prints every aten op that is
not a view/metadata op with the innermost frames of this repository that issued it.
usage (GPU box): python tools/native_ops.py [c3|c2] > gpurun_out/native_ops.txt"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c3'
runner, conf = bench.build_runner(cfg, 'bf16', bench.DEFAULT_BATCH[cfg])
from data.synthetic import synth_batch  # noqa: E402
dev = torch.device('cuda', 0)
batch = {k: v.to(dev) for k, v in synth_batch(bench.DEFAULT_BATCH[cfg], 256, 256, acc=4, seed=1).items()}
runner._request_data = lambda loader, volatile=False: dict(batch)
if cfg == 'c3':
  runner.prefetch_pretrained = False
  step = lambda: runner._train_single_step(None)
else:
  step = lambda: runner._train_step(None)
for _ in range(4):
  step()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

VIEWS = ('view', 'reshape', 'slice', 'select', 'detach', 'alias', 'as_strided', 'expand', 'permute', 'transpose',
         'unsqueeze', 'squeeze', 't.', 'empty', 'size', 'stride', 'is_', '_unsafe_view', 'unbind', 'split', 'narrow',
         'record_stream', 'lift_fresh', '_local_scalar_dense', 'item', 'resize_', 'set_', 'unfold', 'new_empty')
seen = collections.OrderedDict()


class Log(TorchDispatchMode):
  def __torch_dispatch__(self, func, types, args=(), kwargs=None):
    name = str(func)
    short = name.replace('aten.', '')
    if not short.startswith(VIEWS):
      cuda = any(torch.is_tensor(a) and a.is_cuda for a in list(args) + list((kwargs or {}).values())) or \
          any(isinstance(a, (list, tuple)) and a and torch.is_tensor(a[0]) and a[0].is_cuda for a in args)
      dev_kw = (kwargs or {}).get('device')
      if cuda or (dev_kw is not None and 'cuda' in str(dev_kw)):
        frames = ['%s:%d %s' % (f.filename.split('repo/')[-1].split('_amd/')[-1], f.lineno, f.name)
                  for f in traceback.extract_stack() if 'csmri' in f.filename and 'native_ops' not in f.filename][-3:]
        key = (name, ' <- '.join(reversed(frames)))
        seen[key] = seen.get(key, 0) + 1
    return func(*args, **(kwargs or {}))


with Log():
  step()
  torch.cuda.synchronize()
for (name, frames), n in seen.items():
  print('%3d x %-28s %s' % (n, name, frames or '(autograd engine / no repo frame)'))
if cfg == 'c3':
  rest = [(n, tuple(p.shape)) for n, p in runner.gen.named_parameters() if p.requires_grad and not getattr(p, '_kernel_grad', False)]
  print('generator parameters whose gradients autograd (not a kernel) writes:', rest)
  rest = [(n, tuple(p.shape)) for n, p in runner.disc.named_parameters() if p.requires_grad and not getattr(p, '_kernel_grad', False)]
  print('discriminator parameters whose gradients autograd (not a kernel) writes:', rest)
