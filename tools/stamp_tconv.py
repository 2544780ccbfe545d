"""Phase stamps of tconv (stamps build): staging / K loop / epilogue cycles per workgroup, and the spread of start times.
usage: python tools/stamp_tconv.py cin cout k H B [reflection|zero]"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['CSMRI_HIP_LIB'] = os.path.join(ROOT, 'csmri-refinement_amd', 'csmri_hip', 'libcsmri_hip_stamps.so')
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
from csmri_hip import ops, lib
import ctypes as C
cin, cout, k, h, b = [int(a) for a in sys.argv[1:6]]
border = sys.argv[6] if len(sys.argv) > 6 else 'reflection'
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
t = max(k - 1, 0)
pads = (t // 2, t - t // 2, t // 2, t - t // 2)
wt = torch.nn.Parameter((torch.randn(cout, cin, k, k) / math.sqrt(cin * k * k)).cuda())
layer = ops.ConvLayer(wt, None, 1, pads, border, torch.bfloat16)
x = torch.randn(b, h, h, cin, device='cuda').bfloat16()
for _ in range(5):
  ops.conv_forward(layer, x, None, False)
torch.cuda.synchronize()
dbg.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.conv_forward(layer, x, None, False); e1.record(); torch.cuda.synchronize()
blocks = b * ((h + 15) // 16) ** 2 * max(1, cout // 64)
v = dbg[:blocks * 16].view(blocks, 4, 4).double().cpu()
print(patched.name, 'blocks', blocks, 'launch %.1f us' % (e0.elapsed_time(e1) * 1e3))
for i, n in enumerate(['staging (patch + first weights)', 'K loop', 'epilogue']):
  print('  %-34s %8.0f cycles (mean over waves)' % (n, float(v[:, :, i].mean())))
start = v[:, 0, 3]
print('  workgroup start times: span %.0f cycles, total per-workgroup %.0f cycles' % (
    float(start.max() - start.min()), float(v[:, :, :3].sum(2).mean())))
