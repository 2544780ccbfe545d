"""Data-consistency layer (3 launches) and stand-alone FFT timing, HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
from csmri_hip import ops
g = torch.Generator().manual_seed(0)
for b, n in ((8, 256), (64, 256), (2, 512), (16, 512)):
  for dt in (torch.float32, torch.bfloat16):
    x = torch.randn(b, n, n, 2, generator=g).to(dt).cuda()
    k0 = torch.randn(b, n, n, 2, generator=g).cuda()
    m = (torch.rand(b, n, n, generator=g) < 0.25).to(torch.uint8).cuda()
    res = []
    for fn in (lambda: ops.dc_raw(x, k0, m, torch.bfloat16), lambda: ops.fft2(x, False, True)):
      for _ in range(5):
        fn()
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(50):
        fn()
      e1.record(); torch.cuda.synchronize()
      res.append(e0.elapsed_time(e1) / 50 * 1e3)
    es = 4 if dt == torch.float32 else 2
    alg = b * n * n * (2 * es * 2 + 8) + b * n * n          # x in, out, k0 (fp32), mask
    print('B%-3d %d^2 %-8s dc %7.1f us (%.2f TB/s algorithmic)   fft2 %7.1f us' % (
        b, n, str(dt).split('.')[-1], res[0], alg / res[0] / 1e6, res[1]))
