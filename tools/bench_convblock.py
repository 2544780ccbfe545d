"""Fused RecNet conv block (one launch) vs the three per-layer launches: HIP-event times at the workload's shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
from csmri_hip import ops

g = torch.Generator().manual_seed(0)
ws = [torch.randn(32, 2, 3, 3, generator=g) * 0.4, torch.randn(32, 32, 3, 3, generator=g) * 0.08,
      torch.randn(2, 32, 3, 3, generator=g) * 0.08]
for b, size in ((8, 256), (64, 256), (2, 512)):
  for train in (False, True):
    params = [(torch.nn.Parameter(w.clone().cuda(), requires_grad=train),
               torch.nn.Parameter(torch.zeros(w.shape[0]).cuda(), requires_grad=train)) for w in ws]
    plan = [(ops.ConvLayer(wp, bp, 1, (1, 1, 1, 1), 'zero', torch.bfloat16), 0.01 if i < 2 else 1.0)
            for i, (wp, bp) in enumerate(params)]
    x = torch.randn(b, size, size, 8, generator=g).bfloat16().cuda()
    x[..., 2:] = 0
    res = {}
    for fused in (True, False):
      ops.FUSED_CONVBLOCK = fused
      fn = lambda: ops.ConvActStack.apply(x.requires_grad_(train), plan, torch.float32, *[t for pr in params for t in pr])
      for _ in range(3):
        fn()
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(20):
        fn()
      e1.record(); torch.cuda.synchronize()
      res[fused] = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * b * size * size * 9 * (2 * 32 + 32 * 32 + 32 * 2)
    print('B%-3d %d^2 save=%d  fused %8.1f us (%6.1f TF)   per-layer %8.1f us (%6.1f TF)' % (
        b, size, train, res[True], fl / res[True] / 1e6, res[False], fl / res[False] / 1e6))
