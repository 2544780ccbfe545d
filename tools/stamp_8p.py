"""Diagnostic: where a wave of gconv8p spends its cycles, per phase and segment (needs the stamped
library: make -C csmri-refinement_amd/csrc stamps; loaded through CSMRI_HIP_LIB).
usage: python tools/stamp_8p.py [cin cout H B]   (3x3 zero-pad conv, split-K off)"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault('CSMRI_8P', '1')
os.environ.setdefault('CSMRI_HIP_LIB', os.path.join(ROOT, 'csmri-refinement_amd', 'csmri_hip', 'libcsmri_hip_stamps.so'))
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
from csmri_hip import ops, lib
import ctypes as C

cin, cout, h, b = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else (256, 256, 64, 16)
k = 3
tiles = (b * h * h + 255) // 256 * (cout // 256)
dbg = torch.zeros(tiles * 8 * 20, dtype=torch.int64, device='cuda')
def patched(d, want_stats, flops=0.0):
  d.splitk = 1
  d.slab = dbg.data_ptr()
  lib.call('csmri_gconv', C.byref(d), ops.stream())
  return None
ops._gconv_run = patched
wt = torch.nn.Parameter((torch.randn(cout, cin, k, k) / math.sqrt(cin * k * k)).cuda())
layer = ops.ConvLayer(wt, None, 1, (1, 1, 1, 1), 'zero', torch.bfloat16)
x = torch.randn(b, h, h, cin, device='cuda').bfloat16()
for _ in range(20):
  ops.conv_forward(layer, x, None, False)
torch.cuda.synchronize()
steps = k * k * cin // 64
t = dbg.view(tiles, 8, 4, 5).double()
tot = t.sum((2, 3)).mean()
print('%d tiles, %d K tiles: cycles per wave per launch %.0f -> %.0f per K tile (ideal 2048)' % (tiles, steps, tot, tot / steps))
seg = ['wait+reads+stage', 'barrier 1', 'lgkm wait', 'MFMAs', 'barrier 2']
for ph in range(4):
  for s in range(5):
    print('  P%d %-18s all %6.0f   waves 0-3 %6.0f   waves 4-7 %6.0f' % (
        ph + 1, seg[s], t[:, :, ph, s].mean() / steps, t[:, :4, ph, s].mean() / steps, t[:, 4:, ph, s].mean() / steps))
