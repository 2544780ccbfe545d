"""Kernel-alone timing of csmri_convblock_fused_fwd / _bwd at the C2 shape (HIP events, back-to-back launches).
  python tools/bench_convblock_bwd.py [B] [size]      (CSMRI_HIP_LIB selects a variant build of the library)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
from csmri_hip import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
g = torch.Generator().manual_seed(1)
ws = [torch.randn(32, 2, 3, 3, generator=g) * 0.4, torch.randn(32, 32, 3, 3, generator=g) * 0.08,
      torch.randn(2, 32, 3, 3, generator=g) * 0.08]
bs = [torch.randn(32, generator=g) * 0.1, torch.randn(32, generator=g) * 0.1, torch.randn(2, generator=g) * 0.1]
params = [(torch.nn.Parameter(w.cuda()), torch.nn.Parameter(b.cuda())) for w, b in zip(ws, bs)]
plan = [(ops.ConvLayer(wp, bp, 1, (1, 1, 1, 1), 'zero', torch.bfloat16), 0.01 if i < 2 else 1.0)
        for i, (wp, bp) in enumerate(params)]
x = torch.zeros(B, S, S, 8, dtype=torch.bfloat16, device='cuda')
x[..., :2] = torch.randn(B, S, S, 2, generator=g).to(torch.bfloat16).cuda()
x.requires_grad_(True)
gy = torch.randn(B, S, S, 2, generator=g).cuda()
saved = ops.convblock_fused_forward(x.detach(), plan, torch.float32, True, True)
torch.cuda.synchronize()


def timeit(fn, n=20):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3


ops.WGRAD_FINISH_MULTI = '0'                       # finish right behind the kernel (both are timed)
t_f = timeit(lambda: ops.convblock_fused_forward(x.detach(), plan, torch.float32, True, True))
t_b = timeit(lambda: ops.convblock_fused_backward(plan, [x.detach(), saved[1], saved[2], None], gy, True, True))
t_b0 = timeit(lambda: ops.convblock_fused_backward(plan, [x.detach(), saved[1], saved[2], None], gy, False, True))
mf = 2.0 * B * S * S * 9 * (2 * 32 + 32 * 32 + 32 * 2)
print('%s B%d %dx%d  fwd %.1f us (%.0f TF/s)  bwd+finish %.1f us (%.0f TF/s on 2x fwd FLOPs)  bwd without dX %.1f us' % (
    os.path.basename(os.environ.get('CSMRI_HIP_LIB', 'libcsmri_hip.so')), B, S, S, t_f, mf / t_f / 1e6, t_b, 2 * mf / t_b / 1e6, t_b0))
