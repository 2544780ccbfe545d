"""Phase stamps of uconv (stamps build): where the loader and the compute waves of a workgroup spend their cycles.
usage: python tools/stamp_uconv.py cin cout k H B [reflection|zero] [stats]"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['CSMRI_HIP_LIB'] = os.path.join(ROOT, 'csmri-refinement_amd', 'csmri_hip', os.environ.get('UCONV_STAMP_LIB', 'libcsmri_hip_stamps.so'))
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
from csmri_hip import ops, lib
import ctypes as C
cin, cout, k, h, b = [int(a) for a in sys.argv[1:6]]
border = sys.argv[6] if len(sys.argv) > 6 else 'reflection'
want_stats = len(sys.argv) > 7 and sys.argv[7] == 'stats'
dbg = torch.zeros(1 << 20, dtype=torch.int64, device='cuda')
orig = ops._gconv_run
def patched(d, ws, flops=0.0):
  d.splitk = 1
  d.slab = dbg.data_ptr()
  stats = None
  if ws:
    rows = lib.raw('csmri_gconv_stats_rows')(C.byref(d))
    stats = torch.empty(rows, 2, d.Cout, dtype=torch.float32, device='cuda')
    d.stats_partial = stats.data_ptr()
  name = C.create_string_buffer(96)
  lib.call('csmri_gconv_kernel_name', C.byref(d), name, 96)
  patched.name = name.value.decode()
  lib.call('csmri_gconv', C.byref(d), ops.stream())
  return stats
ops._gconv_run = patched
t = max(k - 1, 0)
pads = (t // 2, t - t // 2, t // 2, t - t // 2)
wt = torch.nn.Parameter((torch.randn(cout, cin, k, k) / math.sqrt(cin * k * k)).cuda())
layer = ops.ConvLayer(wt, None, 1, pads, border, torch.bfloat16)
x = torch.randn(b, h, h, cin, device='cuda').bfloat16()
# UCONV_STAMP_WARM launches before the stamped one (default 5; a few thousand = the clock the chip holds under the sustained load)
for _ in range(int(os.environ.get('UCONV_STAMP_WARM', '5'))):
  ops.conv_forward(layer, x, None, False, want_stats=want_stats)
torch.cuda.synchronize()
dbg.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.conv_forward(layer, x, None, False, want_stats=want_stats); e1.record(); torch.cuda.synchronize()
v = dbg.view(-1, 8, 8).double().cpu()
used = (v.sum((1, 2)) > 0)
v = v[used]
print(patched.name, 'workgroups', v.shape[0], 'launch %.1f us' % (e0.elapsed_time(e1) * 1e3))
comp, load = v[:, :4, :6], v[:, 4:, :6]
clk = (v[:, :4, 6] / v[:, :4, 7].clamp(min=1)).median() * 100.0           # MHz: shader cycles per 100 MHz tick
life = v[:, :4, 7].median() / 100.0
print('  in-kernel clock %.0f MHz (s_memtime / s_memrealtime, median over compute waves); a compute wave lives %.1f us' % (float(clk), float(life)))
for i, n in enumerate(['wait for first patch + stage', 'multiplying (between barriers)', 'at the barrier', 'pass tail: MFMAs + epilogue']):
  print('  compute  %-34s %9.0f cycles (mean over waves; max %9.0f)' % (n, float(comp[:, :, i].mean()), float(comp[:, :, i].max())))
print('  compute  total %9.0f' % float(comp.sum(2).mean()))
for i, n in enumerate(['prologue issue', 'prologue landing', 'first barrier', 'waiting for DMA', 'waiting for compute waves', 'issuing']):
  print('  loader   %-34s %9.0f cycles (mean over waves; max %9.0f)' % (n, float(load[:, :, i].mean()), float(load[:, :, i].max())))
print('  loader   total %9.0f' % float(load.sum(2).mean()))
