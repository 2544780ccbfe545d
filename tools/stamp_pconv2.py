"""Diagnostic: per-phase cycle sums of pconv2's step loop, compute waves and loader waves separately (needs the
-DCSMRI_DBG_STAMPS library: make -C csmri-refinement_amd/csrc stamps).
CAVEAT: the stamp accumulators push the kernel over its SGPR budget; the compiler spills SGPRs to VGPR lanes
(v_readlane / v_writelane + s_nop, 4,200 instructions in the once-per-tile epilogue segment): the 'epilogue' row and part of
the 'barrier A' row behind it are artefacts of the stamped build (a build without the stores measured the same).  The read,
MFMA and vmcnt rows are the ones to read: 32 MFMAs issue in ~527 cycles (16.5 each) and the 16 fragment reads take ~335,
hidden under the partner's MFMA block; what is left per half step is barrier latency and skew (~200 cycles).
usage: python tools/stamp_pconv2.py cin cout H B   (3x3 zero-pad stride-1 conv; the VGG shapes: 128 128 128 16 / 256 256 64 16 / 512 512 32 16)"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['CSMRI_HIP_LIB'] = os.path.join(ROOT, 'csmri-refinement_amd', 'csmri_hip', 'libcsmri_hip_stamps.so')
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
from csmri_hip import ops, lib
import ctypes as C

cin, cout, h, b = [int(a) for a in sys.argv[1:5]]
dbg = torch.zeros(1 << 20, dtype=torch.int64, device='cuda')
info = {}
def patched(d, want_stats, flops=0.0):
  d.splitk = 1
  d.slab = dbg.data_ptr()
  name = C.create_string_buffer(96)
  lib.call('csmri_gconv_kernel_name', C.byref(d), name, 96)
  info['name'] = name.value.decode()
  lib.call('csmri_gconv', C.byref(d), ops.stream())
  return None
ops._gconv_run = patched
wt = torch.nn.Parameter((torch.randn(cout, cin, 3, 3) / math.sqrt(cin * 9)).cuda())
layer = ops.ConvLayer(wt, None, 1, (1, 1, 1, 1), 'zero', torch.bfloat16)
x = torch.randn(b, h, h, cin, device='cuda').bfloat16()
for _ in range(5):
  ops.conv_forward(layer, x, None, False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
dbg.zero_()
e0.record()
ops.conv_forward(layer, x, None, False)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
assert 'pconv2' in info['name'], info
ntile = b * ((h + 15) // 16) ** 2
nb = cout // 128
maxw = max(1, 256 // nb)
rounds = (ntile + maxw - 1) // maxw
workers = (ntile + rounds - 1) // rounds
blocks = workers * nb
steps = rounds * (cin // 64) * 9                    # per workgroup (the fullest ones)
t = dbg[:blocks * 96].view(blocks, 12, 8).double().cpu()
flops = 2.0 * b * h * h * cout * cin * 9
print('%s: %.1f us (stamped build), %.0f TFLOP/s; %d workgroups, %d steps each; 100 MHz ticks per step' % (info['name'], us, flops / us / 1e6, blocks, steps))
comp, load = t[:, :8], t[:, 8:]
cn = ['16 fragment reads + lgkmcnt(0)', 'barrier A', '32 MFMAs', 'epilogue', 'barrier B']
ln = ['LDS-DMA issue (weights + patch pieces)', 'barrier A', 'vmcnt wait (stage s+1 landed)', 'barrier B']
cal = comp[:, :, 5].mean() / steps
tot = comp[:, :, :6].sum(2).mean()
print('compute waves: %.1f ticks per step, of which 6 stamps x %.1f; net of stamps:' % (tot / steps, cal))
for i, n in enumerate(cn):
  print('  %-42s %7.2f   (half 0: %7.2f, half 1: %7.2f)' % (n, comp[:, :, i].mean() / steps - cal,
                                                           comp[:, :4, i].mean() / steps - cal, comp[:, 4:, i].mean() / steps - cal))
cal = load[:, :, 4].mean() / steps
tot = load[:, :, :5].sum(2).mean()
print('loader waves: %.1f ticks per step, of which 5 stamps x %.1f; net of stamps:' % (tot / steps, cal))
for i, n in enumerate(ln):
  print('  %-42s %7.2f' % (n, load[:, :, i].mean() / steps - cal))
