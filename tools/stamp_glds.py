"""Diagnostic: per-phase cycle sums of gconv_glds' K loop (needs the -DCSMRI_DBG_STAMPS library).
usage: python tools/stamp_glds.py cin cout H B   (3x3 zero-pad conv, split-K off)"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
from csmri_hip import ops, lib
import ctypes as C

cin, cout, h, b = [int(a) for a in sys.argv[1:5]]
dbg = torch.zeros(1 << 20, dtype=torch.int64, device='cuda')
def patched(d, want_stats, flops=0.0):
  d.splitk = 1
  d.slab = dbg.data_ptr()
  name = C.create_string_buffer(96)
  lib.call('csmri_gconv_kernel_name', C.byref(d), name, 96)
  patched.name = name.value.decode()
  lib.call('csmri_gconv', C.byref(d), ops.stream())
  return None
ops._gconv_run = patched
wt = torch.nn.Parameter((torch.randn(cout, cin, 3, 3) / math.sqrt(cin * 9)).cuda())
layer = ops.ConvLayer(wt, None, 1, (1, 1, 1, 1), 'zero', torch.bfloat16)
x = torch.randn(b, h, h, cin, device='cuda').bfloat16()
for _ in range(20):
  ops.conv_forward(layer, x, None, False)
torch.cuda.synchronize()
bn = 128 if cout % 128 == 0 else 64
blocks = (b * h * h + 127) // 128 * (cout // bn)
steps = 9 * cin // 64
t = dbg[:blocks * 32].view(blocks, 4, 8).double()
tot = t[:, :, :6].sum(2).mean()
print(patched.name, 'blocks', blocks, 'steps', steps, 'cycles per wave per step: %.0f' % (tot / steps))
one = ['DMA issue', 'vmcnt(0) wait', 'barrier', 'reads+mma (both kc)', 'barrier 2', '-']
two = ['vmcnt(0) wait', 'barrier', 'reads kc0 + DMA a', 'mma kc0', 'reads kc1 + DMA b', 'mma kc1']
names = one if ', 1>' in patched.name else two
for i, n in enumerate(names):
  print('  %-22s %7.0f per step  %5.1f %%' % (n, t[:, :, i].mean() / steps, 100 * t[:, :, i].mean() / tot))
