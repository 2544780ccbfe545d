"""Micro-benchmark of single conv launches (HIP events) for kernel tuning.
usage: python tools/bench_conv.py [name ...]"""
import os, sys, math
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
from csmri_hip import ops

CASES = {
    # name: (cin, cout, k, stride, border, up, H, W, B)
    'rec_mid': (32, 32, 3, 1, 'zero', False, 256, 256, 8),
    'unet32': (32, 32, 4, 1, 'reflection', False, 256, 256, 8),
    'unet64': (64, 64, 4, 1, 'reflection', False, 128, 128, 8),
    'unet_cat': (64, 32, 4, 1, 'reflection', False, 256, 256, 8),
    'vgg1_2': (64, 64, 3, 1, 'zero', False, 256, 256, 8),
    'vgg2_1': (64, 128, 3, 1, 'zero', False, 128, 128, 8),
    'vgg2_2': (128, 128, 3, 1, 'zero', False, 128, 128, 8),
    'vgg3_2': (256, 256, 3, 1, 'zero', False, 64, 64, 8),
    'vgg4_2': (512, 512, 3, 1, 'zero', False, 32, 32, 8),
    'vgg5_2': (512, 512, 3, 1, 'zero', False, 16, 16, 8),
    'disc3': (128, 256, 4, 2, 'reflection', False, 64, 64, 8),
    'disc5': (512, 1024, 4, 2, 'reflection', False, 16, 16, 8),
    'disc6': (1024, 1024, 4, 1, 'reflection', False, 8, 8, 8),
    'u32x64': (32, 64, 4, 1, 'reflection', False, 256, 256, 8),
    'u32x64z': (32, 64, 4, 1, 'zero', False, 259, 259, 8),
    'vgg5_2b16': (512, 512, 3, 1, 'zero', False, 16, 16, 16),
    'vgg3_2c5': (256, 256, 3, 1, 'zero', False, 128, 128, 4), 'vgg4_2c5': (512, 512, 3, 1, 'zero', False, 64, 64, 4),
    'vgg4_2b16': (512, 512, 3, 1, 'zero', False, 32, 32, 16),
    'vgg3_2b16': (256, 256, 3, 1, 'zero', False, 64, 64, 16),
    'vgg4_1b16': (256, 512, 3, 1, 'zero', False, 32, 32, 16),
    'vgg3_1b16': (128, 256, 3, 1, 'zero', False, 64, 64, 16),
    'vgg2_2b16': (128, 128, 3, 1, 'zero', False, 128, 128, 16),
    'vgg2_1b16': (64, 128, 3, 1, 'zero', False, 128, 128, 16),
    'u128': (128, 128, 4, 1, 'reflection', False, 64, 64, 8),
    'disc2b8': (64, 128, 4, 2, 'reflection', False, 64, 64, 8),
    'rec_first': (2, 32, 3, 1, 'zero', False, 256, 256, 8),
    'unet_head': (32, 1, 1, 1, 'zero', False, 256, 256, 8),
    'disc_final': (1024, 1, 4, 1, 'zero', False, 8, 8, 16),
    'disc_first': (1, 64, 4, 2, 'reflection', False, 256, 256, 16),
    'rec_last': (32, 2, 3, 1, 'zero', False, 256, 256, 8),
    'vgg1_1': (3, 64, 3, 1, 'zero', False, 256, 256, 16),
    'unet_first': (2, 32, 4, 1, 'reflection', False, 256, 256, 8),
    'd1cat': (128, 64, 4, 1, 'reflection', False, 128, 128, 8),
    'd1up': (128, 64, 4, 1, 'reflection', True, 64, 64, 8),
    # discriminator layers 2-6 at the 16-image (D-phase backward), 8-image (generator-phase data gradient) and 24-image (forward) batches
    'dl2b16': (64, 128, 4, 2, 'reflection', False, 128, 128, 16), 'dl3b16': (128, 256, 4, 2, 'reflection', False, 64, 64, 16),
    'dl4b16': (256, 512, 4, 2, 'reflection', False, 32, 32, 16), 'dl5b16': (512, 1024, 4, 2, 'reflection', False, 16, 16, 16),
    'dl6b16': (1024, 1024, 4, 1, 'reflection', False, 8, 8, 16),
    'dl2b8': (64, 128, 4, 2, 'reflection', False, 128, 128, 8), 'dl3b8': (128, 256, 4, 2, 'reflection', False, 64, 64, 8),
    'dl4b8': (256, 512, 4, 2, 'reflection', False, 32, 32, 8), 'dl5b8': (512, 1024, 4, 2, 'reflection', False, 16, 16, 8),
    'dl6b8': (1024, 1024, 4, 1, 'reflection', False, 8, 8, 8),
    'dl2b24': (64, 128, 4, 2, 'reflection', False, 128, 128, 24), 'dl3b24': (128, 256, 4, 2, 'reflection', False, 64, 64, 24),
    'dl4b24': (256, 512, 4, 2, 'reflection', False, 32, 32, 24), 'dl5b24': (512, 1024, 4, 2, 'reflection', False, 16, 16, 24),
    'dl6b24': (1024, 1024, 4, 1, 'reflection', False, 8, 8, 24),
}


def pads_for(k, s):
  total = int(math.ceil((k - 1.0) / s)); lo = total // 2; hi = lo if total % 2 == 0 else lo + 1
  return (lo, hi, lo, hi)


def run(name, mode='fwd', iters=20):
  cin, cout, k, s, border, up, h, w, b = CASES[name]
  wt = torch.nn.Parameter((torch.randn(cout, cin, k, k) / math.sqrt(cin * k * k)).cuda())
  bias = torch.nn.Parameter(torch.randn(cout).cuda() * 0.1) if mode == 'fwdb' else None
  layer = ops.ConvLayer(wt, bias, s, pads_for(k, s), border, torch.bfloat16, upsample=up)
  x = torch.randn(b, h, w, ops.pad8(cin), device='cuda').bfloat16()
  y, _ = ops.conv_forward(layer, x, None, False)
  gy = torch.randn_like(y)
  if mode in ('fwdb8', 'fwdbq'):
    # frozen-stack forms (ops.Fp8Chain): fwdbq = bf16 operands, bias + ReLU, fp8 copy + maximum of the output from the epilogue;
    # fwdb8 = the same with fp8 operands (needs cin == cout: the layer is chained behind itself)
    fl = ops.ConvLayer(wt, torch.nn.Parameter(torch.randn(cout).cuda() * 0.1), s, pads_for(k, s), border, torch.bfloat16, frozen=True)
    chain = ops.Fp8Chain([('conv', fl, 0.0), ('conv', fl, 0.0)], x.device)
    xr = torch.relu(x)
    ops.frozen_conv_forward(fl, xr, 0.0, None, 0, 0, chain); chain.finish()
    y1, yq1 = ops.frozen_conv_forward(fl, xr, 0.0, None, 0, 0, chain)
    slot = None if os.environ.get('F8_NOOUT') else 0      # F8_NOOUT=1: without the fp8 copy / maximum of the output
    if mode == 'fwdbq':
      fn8 = lambda: ops.frozen_conv_forward(fl, xr, 0.0, None, 0, slot, chain)
    else:
      assert cin == cout
      fn8 = lambda: ops.frozen_conv_forward(fl, y1, 0.0, yq1, chain.dq_scale_ptr(0), slot, chain)
  fn = {'fwdb8': lambda: fn8(), 'fwdbq': lambda: fn8(),
        'fwd': lambda: ops.conv_forward(layer, x, None, False),
        'fwdb': lambda: ops.conv_forward(layer, x, None, True, 0.0),          # bias + ReLU (the VGG forward)
        'fwds': lambda: ops.conv_forward(layer, x, None, False, 1.0, True),   # BatchNorm partial sums (the U-Net forward)
        'dgrad': lambda: ops.conv_dgrad(layer, gy, (h, w)),
        'dgradg': lambda: ops.conv_dgrad(layer, gy, (h, w), g_src=x, g_slope=0.0),   # gated by the producer's ReLU (VGG backward)
        'wgrad': lambda: ops.conv_wgrad(layer, x, None, gy)}[mode]
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
  byts = (x.numel() + y.numel()) * 2
  print('%-10s %-6s %8.1f us  %7.1f TFLOP/s  %6.2f TB/s(in+out)' % (name, mode, us, flops / us / 1e6, byts / us / 1e6))


if __name__ == '__main__':
  ops.GCONV_FLAGS = int(os.environ.get('GCONV_FLAGS', '0'))
  names = [a for a in sys.argv[1:] if a in CASES] or list(CASES)
  modes = [a for a in sys.argv[1:] if a in ('fwd', 'fwdb', 'fwdb8', 'fwdbq', 'fwds', 'dgrad', 'dgradg', 'wgrad')] or ['fwd']
  for n in names:
    for m in modes:
      run(n, m)
