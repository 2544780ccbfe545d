"""Diagnostic: per-phase cycle sums of gconv_glds256's K loop (needs the -DCSMRI_DBG_STAMPS library)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
from csmri_hip import ops, lib
import ctypes as C

dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device='cuda')
orig = ops._gconv_run
def patched(d, want_stats, flops=0.0):
  d.splitk = 1
  d.slab = dbg.data_ptr()
  lib.call('csmri_gconv', C.byref(d), ops.stream())
  return None
ops._gconv_run = patched
cin, cout, k, h, w, b = 256, 256, 3, 64, 64, 16
wt = torch.nn.Parameter((torch.randn(cout, cin, k, k) / math.sqrt(cin * k * k)).cuda())
layer = ops.ConvLayer(wt, None, 1, (1, 1, 1, 1), 'zero', torch.bfloat16)
x = torch.randn(b, h, w, cin, device='cuda').bfloat16()
for _ in range(20):
  ops.conv_forward(layer, x, None, False)
torch.cuda.synchronize()
t = dbg.view(256, 8, 8).double()
names = ['barrier wait', 'frag reads kc0', 'DMA issue', 'mma kc0', 'frag reads kc1', 'mma kc1']
tot = t[:, :, :6].sum(2).mean()
print('cycles per wave per launch (mean over 256 blocks x 8 waves): %.0f  (36 steps -> %.0f per step)' % (tot, tot / 36))
for i, n in enumerate(names):
  print('  %-16s %7.0f per step  %5.1f %%   waves 0-3 %7.0f  waves 4-7 %7.0f' % (
      n, t[:, :, i].mean() / 36, 100 * t[:, :, i].mean() / tot, t[:, :4, i].mean() / 36, t[:, 4:, i].mean() / 36))
