"""Warm vs cold (Infinity Cache flushed) time of single conv launches: how much of an in-step launch is first-touch HBM latency?"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
from csmri_hip import ops
from bench_conv import CASES, pads_for

flush = torch.empty(768 * 1024 * 1024 // 4, dtype=torch.float32, device='cuda')
def run(name, mode, cold, what='all'):
  cin, cout, k, s, border, up, h, w, b = CASES[name]
  wt = torch.nn.Parameter((torch.randn(cout, cin, k, k) / math.sqrt(cin * k * k)).cuda())
  layer = ops.ConvLayer(wt, None, s, pads_for(k, s), border, torch.bfloat16, upsample=up)
  x = torch.randn(b, h, w, ops.pad8(cin), device='cuda').bfloat16()
  y, _ = ops.conv_forward(layer, x, None, False)
  gy = torch.randn_like(y)
  fn = {'fwd': lambda: ops.conv_forward(layer, x, None, False), 'dgrad': lambda: ops.conv_dgrad(layer, gy, (h, w))}[mode]
  for _ in range(3):
    fn()
  ts = []
  for _ in range(12):
    if cold:
      flush.add_(1.0)                      # 1.5 GB of traffic: evicts the 256 MB Infinity Cache and every L2
      if what == 'weights_warm':           # re-touch only the weights (what a prefetch launch would do)
        for pk in layer._packs.values():
          pk[1].float().sum()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
  ts.sort()
  return ts[len(ts) // 2]
for name, mode in (('disc6', 'fwd'), ('disc5', 'fwd'), ('disc3', 'fwd'), ('vgg4_2b16', 'fwd'), ('vgg4_2', 'dgrad'), ('vgg3_2b16', 'fwd'), ('u128', 'fwd'), ('unet64', 'fwd')):
  print('%-10s %-5s warm %7.1f us   cold %7.1f us   cold but weights re-touched %7.1f us' % (
      name, mode, run(name, mode, False), run(name, mode, True), run(name, mode, True, 'weights_warm')))
