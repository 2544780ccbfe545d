"""Diagnostic: per-phase cycle sums of gpipe's loader and compute waves (needs the -DCSMRI_DBG_STAMPS library:
CSMRI_HIP_LIB=csmri-refinement_amd/csmri_hip/libcsmri_hip_stamps.so).
usage: python tools/stamp_gpipe.py <bench_conv case> [fwd|dgrad]"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
from csmri_hip import ops, lib
ops.GCONV_FLAGS = 2
import ctypes as C
import bench_conv

name = sys.argv[1]
mode = sys.argv[2] if len(sys.argv) > 2 else 'fwd'
cin, cout, k, s, border, up, h, w, b = bench_conv.CASES[name]
dbg = torch.zeros(1 << 20, dtype=torch.int64, device='cuda')
info = {}
def patched(d, want_stats, flops=0.0):
  d.flags = 2
  d.splitk = 1                                  # (stamps are dumped by unsplit launches: the slab pointer carries the buffer)
  d.slab = dbg.data_ptr()
  nm = C.create_string_buffer(96)
  lib.call('csmri_gconv_kernel_name', C.byref(d), nm, 96)
  info['name'] = nm.value.decode()
  info['steps'] = d.TH * d.TW * d.Cin // 64
  lib.call('csmri_gconv', C.byref(d), ops.stream())
  return None
ops._gconv_run = patched
wt = torch.nn.Parameter((torch.randn(cout, cin, k, k) / math.sqrt(cin * k * k)).cuda())
layer = ops.ConvLayer(wt, None, s, bench_conv.pads_for(k, s), border, torch.bfloat16, upsample=up)
x = torch.randn(b, h, w, ops.pad8(cin), device='cuda').bfloat16()
ho, wo = layer.out_hw(h, w)
gy = torch.randn(b, ho, wo, ops.pad8(cout), device='cuda').bfloat16()
fn = (lambda: ops.conv_forward(layer, x, None, False)) if mode == 'fwd' else (lambda: ops.conv_dgrad(layer, gy, (h, w)))
for _ in range(3):
  fn()
torch.cuda.synchronize()
dbg.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record()
torch.cuda.synchronize()
t = dbg[:256 * 12 * 8].view(256, 12, 8).double()
busy = t[:, 0, :].sum(1) > 0
nb = int(busy.sum())
t = t[busy]
print(info['name'], name, mode, 'workgroups with work', nb, 'K steps per item', info['steps'], 'launch %.1f us' % (e0.elapsed_time(e1) * 1e3))
ld = t[:, 8:12, :5]
cp = t[:, 0:8, :5]
tot_l, tot_c = ld.sum(2).mean(), cp.sum(2).mean()
print(' loader waves: %.0f cycles in the step loop per wave' % tot_l)
for i, n in enumerate(['issue part 1 of stage s+2', 'barrier A', 'vmcnt wait for stage s+1', 'issue part 2 (+ address rebuilds)', 'barrier B']):
  print('   %-40s %9.0f  %5.1f %%' % (n, ld[:, :, i].mean(), 100 * ld[:, :, i].mean() / tot_l))
print(' compute waves: %.0f cycles in the step loop per wave' % tot_c)
for i, n in enumerate(['fragment reads + wait', 'barrier A', 'MFMA block', 'epilogue (item ends)', 'barrier B']):
  print('   %-40s %9.0f  %5.1f %%' % (n, cp[:, :, i].mean(), 100 * cp[:, :, i].mean() / tot_c))
