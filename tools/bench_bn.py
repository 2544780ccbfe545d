"""Micro-benchmark of the BatchNorm kernels (use under rocprofv3 for per-kernel times)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
from csmri_hip import ops

SHAPES = [(8, 256, 256, 32), (8, 128, 128, 64), (8, 64, 64, 128), (16, 32, 32, 256), (16, 8, 8, 1024)]
for b, h, w, c in SHAPES:
  y = torch.randn(b, h, w, c, device='cuda').bfloat16()
  gz = torch.randn_like(y)
  bn = ops.BNState(torch.ones(c, device='cuda'), torch.zeros(c, device='cuda'),
                   torch.zeros(c, device='cuda'), torch.ones(c, device='cuda'))
  for it in range(12):
    z, mean, invstd, snap = ops._bn_forward(y, None, bn, c, 0.2, True, None, 1)
    rows = ops.lib.raw('csmri_bn_stats_rows')(b * h * w, c)
    partial = torch.empty(rows + 1, 2, c, dtype=torch.float32, device='cuda')
    ops.lib.call('csmri_bn_bwd_reduce', ops.dt_of(y), gz.data_ptr(), gz.stride(2), y.data_ptr(), y.stride(2), 0, 0,
                 b, h * w, c, mean.data_ptr(), invstd.data_ptr(), 0.2, 0, partial.data_ptr(), snap.data_ptr(), 1,
                 0, 0, ops.stream())
    gy = torch.empty_like(y)
    ops.lib.call('csmri_bn_bwd_apply', ops.dt_of(y), gz.data_ptr(), gz.stride(2), y.data_ptr(), y.stride(2), 0, 0,
                 gy.data_ptr(), gy.stride(2), b, h * w, c, c, mean.data_ptr(), invstd.data_ptr(),
                 bn.weight.data_ptr(), 0.2, 0, partial.data_ptr(), rows, 0, 0, 1, snap.data_ptr(), 1, 0, 0, ops.stream())
  torch.cuda.synchronize()
  print('done', b, h, w, c, 'MB per tensor', b * h * w * c * 2 / 1e6)
