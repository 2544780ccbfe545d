"""The patch convolutions (pconv / pconv2, as csmri_gconv dispatches them) against torch fp32 on bf16-rounded operands + timing; usage: check_pconv2.py [case ...] [fwd|dgrad]"""
import os, sys, math
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from csmri_hip import ops, lib
from bench_conv import CASES, pads_for

CASES.update({
    'vgg3_2b8': (256, 256, 3, 1, 'zero', False, 64, 64, 8),
    'vgg4_2b8': (512, 512, 3, 1, 'zero', False, 32, 32, 8),
    'vgg2_2b8': (128, 128, 3, 1, 'zero', False, 128, 128, 8),
    'vgg3_1b8d': (256, 128, 3, 1, 'zero', False, 64, 64, 8),
    'u128b8': (128, 128, 4, 1, 'reflection', False, 64, 64, 8),
    'vgg1_2b16': (64, 64, 3, 1, 'zero', False, 256, 256, 16),
    'vgg1_2b8': (64, 64, 3, 1, 'zero', False, 256, 256, 8),
    'odd64': (64, 64, 3, 1, 'zero', False, 150, 137, 3),
    'odd': (128, 128, 3, 1, 'zero', False, 50, 37, 3),
    'oddr': (192, 256, 4, 1, 'reflection', False, 35, 50, 2),
})


def run(name, mode, iters=30):
  cin, cout, k, s, border, up, h, w, b = CASES[name]
  torch.manual_seed(1)
  wt = torch.nn.Parameter((torch.randn(cout, cin, k, k) / math.sqrt(cin * k * k)).cuda())
  bias = torch.nn.Parameter(torch.randn(cout).cuda() * 0.1)
  pads = pads_for(k, s)
  layer = ops.ConvLayer(wt, bias, s, pads, border, torch.bfloat16, upsample=up)
  x = torch.randn(b, h, w, ops.pad8(cin), device='cuda').bfloat16()
  y, _ = ops.conv_forward(layer, x, None, False)
  wf = wt.detach().bfloat16().float()
  if mode == 'fwd':
    xin = x.float().permute(0, 3, 1, 2)[:, :cin]
    xp = F.pad(xin, pads, mode='reflect' if border == 'reflection' else 'constant')
    ref = F.conv2d(xp, wf, bias.detach(), stride=s).permute(0, 2, 3, 1)
    out = y.float()[..., :cout]
    fn = lambda: ops.conv_forward(layer, x, None, False)
  else:
    gy = torch.randn_like(y)
    out = ops.conv_dgrad(layer, gy, (h, w)).float()[..., :cin]
    xin = x.float().permute(0, 3, 1, 2)[:, :cin].clone().requires_grad_(True)
    xp = F.pad(xin, pads, mode='reflect' if border == 'reflection' else 'constant')
    yy = F.conv2d(xp, wf, None, stride=s)
    yy.backward(gy.float()[..., :cout].permute(0, 3, 1, 2))
    ref = xin.grad.permute(0, 2, 3, 1)
    fn = lambda: ops.conv_dgrad(layer, gy, (h, w))
  err = (out - ref).abs().max().item() / ref.abs().max().item()
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters):
    fn()
  e1.record(); torch.cuda.synchronize()
  us = e0.elapsed_time(e1) / iters * 1e3
  flops = 2.0 * b * y.shape[1] * y.shape[2] * cout * cin * k * k
  print('%-10s %-6s rel.err %.2e %8.1f us %7.1f TFLOP/s' % (name, mode, err, us, flops / us / 1e6), flush=True)


if __name__ == '__main__':
  names = [a for a in sys.argv[1:] if a in CASES]
  modes = [a for a in sys.argv[1:] if a in ('fwd', 'dgrad')] or ['fwd', 'dgrad']
  for n in names:
    for m in modes:
      run(n, m)
