"""Phase stamps of csmri_convblock_fused_bwd at the C2 shape (stamps build: make -C csmri-refinement_amd/csrc stamps).
Per wave and tile: cycles between the stamps of csrc/convblock_bwd.hip (s_memtime ticks), mean / max over waves."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ['CSMRI_HIP_LIB'] = os.path.join(ROOT, 'csmri-refinement_amd', 'csmri_hip', 'libcsmri_hip_stamps.so')
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
from csmri_hip import ops, lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = 256
g = torch.Generator().manual_seed(1)
ws = [torch.randn(32, 2, 3, 3, generator=g) * 0.4, torch.randn(32, 32, 3, 3, generator=g) * 0.08,
      torch.randn(2, 32, 3, 3, generator=g) * 0.08]
layers = [ops.ConvLayer(torch.nn.Parameter(w.cuda()), torch.nn.Parameter(torch.zeros(w.shape[0]).cuda()), 1, (1, 1, 1, 1),
                        'zero', torch.bfloat16) for w in ws]
plan = [(l, 0.01 if i < 2 else 1.0) for i, l in enumerate(layers)]
x = torch.zeros(B, S, S, 8, dtype=torch.bfloat16, device='cuda')
x[..., :2] = torch.randn(B, S, S, 2, generator=g).to(torch.bfloat16).cuda()
gy = torch.randn(B, S, S, 2, generator=g).cuda()
saved = ops.convblock_fused_forward(x, plan, torch.float32, True, True)
d = lib.ConvBlockBwdDesc()
d.dtype = 1
d.num_convs, d.num_filters, d.kernel_size, d.num_inputs, d.num_outputs, d.border = 3, 32, 3, 2, 2, 0
d.x, d.x_pix_stride, d.B, d.H, d.W = x.data_ptr(), 8, B, S, S
a1, a2 = saved[1], saved[2]
d.act[0], d.act[1] = a1.data_ptr(), a2.data_ptr()
d.act_pix_stride[0], d.act_pix_stride[1] = 32, 32
d.gy, d.gy_dtype, d.gy_pix_stride = gy.data_ptr(), 0, 2
keep = []
for i, l in enumerate(layers):
  wp, kp, _, _ = l._pack(3)
  d.wd[i], d.Kp[i] = wp.data_ptr(), kp
  keep.append(wp)
d.slope = 0.01
z = lib.raw('csmri_convblock_fused_bwd_splits')(B, S, S)
d.splits = z
slabs = [torch.zeros(z * 32 * 288 + z * 32 + 1024, dtype=torch.float32, device='cuda') for _ in range(3)]
for i in range(3):
  d.slab[i] = slabs[i].data_ptr()
dbg = torch.zeros(z * 8 * 16, dtype=torch.int64, device='cuda')
dx = torch.empty(B, S, S, 8, dtype=torch.bfloat16, device='cuda')
for mode in ('with dX (timing only)', 'stamps (dX skipped)'):
  if mode.startswith('with'):
    d.dx, d.dx_pix_stride, d.want_db = dx.data_ptr(), 8, 1
  else:
    d.dx, d.dx_pix_stride, d.want_db = dbg.data_ptr(), 8, 2
  for _ in range(3):
    lib.call('csmri_convblock_fused_bwd', C.byref(d), ops.stream())
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  dbg.zero_()
  e0.record()
  lib.call('csmri_convblock_fused_bwd', C.byref(d), ops.stream())
  e1.record()
  torch.cuda.synchronize()
  print('%s: %.1f us per launch (B %d, %d workgroups)' % (mode, e0.elapsed_time(e1) * 1e3, B, z))
v = dbg.view(z, 8, 16).double().cpu()[..., :13]
tiles = B * 256.0 / z
names = ['store dY pixel + wait DMA (a2, dY of this tile)', 'barrier', 'issue DMA a1 / x', 'stage 1: dA2 (conv3 data gradient)',
         'weight gradient of layer 3', 'wait + barrier', 'next tile: load dY, issue DMA a2', 'stage 2: dA1 (conv2 data gradient)',
         'weight gradient of layer 2', 'barrier', 'stage 3: dX (skipped in the stamped run)', 'weight gradient of layer 1', 'barrier']
mean, mx = v.mean((0, 1)) / tiles, v.amax((0, 1)) / tiles
print('ticks per tile (mean over the %d waves | slowest wave):' % (z * 8))
for n, a, b in zip(names, mean.tolist(), mx.tolist()):
  print('  %-52s %9.1f | %9.1f' % (n, a, b))
print('  total %9.1f per tile; %.1f tiles per workgroup' % (float(mean.sum()), tiles))
for w in range(8):
  print('  wave %d: ' % w + ' '.join('%7.0f' % (c / tiles) for c in v[:, w].mean(0).tolist()))
