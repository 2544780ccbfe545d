"""Phase stamps of the fused conv-block kernel (stamps build: make -C csmri-refinement_amd/csrc stamps)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['CSMRI_HIP_LIB'] = os.path.join(ROOT, 'csmri-refinement_amd', 'csmri_hip', 'libcsmri_hip_stamps.so')
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
from csmri_hip import ops, lib
g = torch.Generator().manual_seed(0)
ws = [torch.randn(32, 2, 3, 3, generator=g) * 0.4, torch.randn(32, 32, 3, 3, generator=g) * 0.08,
      torch.randn(2, 32, 3, 3, generator=g) * 0.08]
b, size = int(os.environ.get('B', '8')), 256
layers = [ops.ConvLayer(torch.nn.Parameter(w.cuda()), torch.nn.Parameter(torch.zeros(w.shape[0]).cuda()), 1, (1, 1, 1, 1),
                        'zero', torch.bfloat16) for w in ws]
x = torch.randn(b, size, size, 8, generator=g).bfloat16().cuda()
d = lib.ConvBlockDesc()
d.dtype, d.num_convs, d.num_filters, d.kernel_size, d.num_inputs, d.num_outputs, d.border = 1, 3, 32, 3, 2, 2, 0
keep = []
for i, l in enumerate(layers):
  wp, kp, _, _ = l._pack(0); bp = l.bias_padded()
  d.w[i], d.Kp[i], d.bias[i] = wp.data_ptr(), kp, bp.data_ptr(); keep += [wp, bp]
d.x, d.x_pix_stride, d.B, d.H, d.W, d.slope = x.data_ptr(), 8, b, size, size, 0.01
y = torch.empty(b, size, size, 8, dtype=torch.float32, device='cuda')
d.out, d.out_dtype, d.out_pix_stride = y.data_ptr(), 0, 8
dbg = torch.zeros(512 * 4 * 8, dtype=torch.int64, device='cuda')
d.act[0] = dbg.data_ptr()
for _ in range(3):
  lib.call('csmri_convblock_fused_fwd', C.byref(d), ops.stream())
torch.cuda.synchronize()
dbg.zero_()
lib.call('csmri_convblock_fused_fwd', C.byref(d), ops.stream())
torch.cuda.synchronize()
v = dbg.view(512, 4, 8).double().cpu()
tiles = b * 256 / 512.0
names = ['patch load+store', 'barrier', 'layer 1', 'barrier', 'layer 2', 'barrier', 'layer 3 + store']
m = v.mean((0, 1)) / tiles
print('cycles (100 MHz s_memtime ticks x ?) per tile, mean over waves:')
for n, c in zip(names, m[:7].tolist()):
  print('  %-18s %9.1f' % (n, c))
print('  total %9.1f per tile; tiles per workgroup %.1f' % (float(m[:7].sum()), tiles))
