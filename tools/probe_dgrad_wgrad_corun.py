"""Probe for the 'one launch = [data-gradient tiles | weight-gradient tiles]' idea on the discriminator's deep layers:
the best case of such a fused launch is the two kernels co-resident on the chip, which two streams give without writing it.
Times, per layer pair of the D-phase backward (16 images): data gradient of layer L alone, weight gradient of layer L+1
alone, both back to back on one stream, both at once on two streams.
usage: python tools/probe_dgrad_wgrad_corun.py"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
from csmri_hip import ops

L = {2: (64, 128, 2, 128), 3: (128, 256, 2, 64), 4: (256, 512, 2, 32), 5: (512, 1024, 2, 16), 6: (1024, 1024, 1, 8)}
B = 16


def pads_for(k, s):
  total = int(math.ceil((k - 1.0) / s)); lo = total // 2; hi = lo if total % 2 == 0 else lo + 1
  return (lo, hi, lo, hi)


def make(l):
  cin, cout, s, h = L[l]
  wt = torch.nn.Parameter((torch.randn(cout, cin, 4, 4) / math.sqrt(cin * 16)).cuda())
  layer = ops.ConvLayer(wt, None, s, pads_for(4, s), 'reflection', torch.bfloat16)
  x = torch.randn(B, h, h, cin, device='cuda').bfloat16()
  y, _ = ops.conv_forward(layer, x, None, False)
  gy = torch.randn_like(y)
  return layer, x, gy, h


def timed(fn, iters=30):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters):
    fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e3


side = torch.cuda.Stream()
for l in (5, 4, 3, 2):
  la, xa, gya, ha = make(l)          # data gradient of layer l
  lb, xb, gyb, hb = make(l + 1)      # weight gradient of layer l + 1 (its dY is ready when layer l's data gradient starts)
  dgrad = lambda: ops.conv_dgrad(la, gya, (ha, ha))
  wgrad = lambda: ops.conv_wgrad(lb, xb, None, gyb, accumulate=False)

  def both_serial():
    dgrad(); wgrad()

  def both_corun():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      wgrad()
    dgrad()
    torch.cuda.current_stream().wait_stream(side)
  td, tw, ts, tc = timed(dgrad), timed(wgrad), timed(both_serial), timed(both_corun)
  print('dgrad L%d %6.1f us | wgrad L%d %6.1f us | back to back %6.1f us | two streams at once %6.1f us' % (l, td, l + 1, tw, ts, tc))
