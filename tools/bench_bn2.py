"""Per-kernel timing of the BatchNorm passes per layer shape (HIP events, 20 launches each)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
from csmri_hip import ops
lib = ops.lib

def t(fn, n=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3

SHAPES = [(8, 256, 256, 32), (8, 128, 128, 64), (8, 64, 64, 128), (16, 64, 64, 128), (16, 32, 32, 256), (16, 16, 16, 512), (16, 8, 8, 1024)]
for b, h, w, c in SHAPES:
  y = torch.randn(b, h, w, c, device='cuda').bfloat16()
  gz = torch.randn_like(y); z = torch.empty_like(y); gy = torch.empty_like(y)
  bn = ops.BNState(torch.ones(c, device='cuda'), torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda'), torch.ones(c, device='cuda'))
  n = b * h * w
  rows = lib.raw('csmri_bn_stats_rows')(n, c)
  stats = torch.empty(rows, 2, c, dtype=torch.float32, device='cuda')
  mean = torch.zeros(1, c, device='cuda'); invstd = torch.ones(1, c, device='cuda'); snap = torch.empty(2, c, device='cuda')
  partial = torch.empty(rows + 1, 2, c, dtype=torch.float32, device='cuda')
  dt = ops.dt_of(y)
  t_stats = t(lambda: lib.call('csmri_bn_stats', dt, y.data_ptr(), y.stride(2), n, c, stats.data_ptr(), 1, ops.stream()))
  t_fin = t(lambda: lib.call('csmri_bn_finalize', stats.data_ptr(), rows, c, c, n, 1e-5, 0.1, mean.data_ptr(), invstd.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), 1, ops.stream()))
  t_act = t(lambda: lib.call('csmri_bn_act', dt, y.data_ptr(), y.stride(2), z.data_ptr(), z.stride(2), b, h * w, c, c, mean.data_ptr(), invstd.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), 0.2, 0, snap.data_ptr(), 1, 0, 0, ops.stream()))
  t_red = t(lambda: lib.call('csmri_bn_bwd_reduce', dt, gz.data_ptr(), gz.stride(2), y.data_ptr(), y.stride(2), 0, 0, b, h * w, c, mean.data_ptr(), invstd.data_ptr(), 0.2, 0, partial.data_ptr(), snap.data_ptr(), 1, 0, 0, ops.stream()))
  t_app = t(lambda: lib.call('csmri_bn_bwd_apply', dt, gz.data_ptr(), gz.stride(2), y.data_ptr(), y.stride(2), 0, 0, gy.data_ptr(), gy.stride(2), b, h * w, c, c, mean.data_ptr(), invstd.data_ptr(), bn.weight.data_ptr(), 0.2, 0, partial.data_ptr(), rows, 0, 0, 1, snap.data_ptr(), 1, 0, 0, ops.stream()))
  mb = n * c * 2 / 1e6
  print('%2dx%3dx%3dx%4d %6.1f MB rows %4d | stats %6.1f us (%4.2f TB/s) finalize %5.1f | act %6.1f (%4.2f) | bwd_reduce %6.1f (%4.2f) | bwd_finalize+apply %6.1f (%4.2f)' %
        (b, h, w, c, mb, rows, t_stats, mb / t_stats, t_fin, t_act, 2 * mb / t_act, t_red, 2 * mb / t_red, t_app, 3 * mb / t_app))
