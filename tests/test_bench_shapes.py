"""GPU parity of the convolution library at the EXACT layer shapes of the benchmarked workload
(bench.py: configs/2-refinement.json at full width, 256x256, 8 slices per GPU, bf16) against the CPU
oracle (plain torch fp32 F.conv2d on the same bf16-rounded operands), one case per distinct layer:

  frozen RecNet (reference models/recnet.py:29-62), U-Net (models/unet.py:27-290 as configured in
  configs/2-refinement.json:31-47), CNNDiscriminator (models/discriminators.py:137-172; the D phase
  forward is one 24-image pass over [pool-fake; real; current-fake], the D-phase backward a 16-image and the
  generator-phase data gradient an 8-image pass), VGG19
  (models/vgg.py:8-80; [pred; target] = 16 images forward, 8 images data gradient).

Whatever kernel instance the dispatcher picks for a shape is the one checked here (its name is logged
and collected in `SEEN`); test_dispatch_variants_are_all_exercised then asserts that every instance
family the library can dispatch to at these shapes has been compared with the oracle.

Tolerances (bf16 compute, fp32 accumulation): forward / data gradient are rounded to bf16 on output:
relative L2 <= 3e-3 (the rounding alone is 1.1e-3 RMS); weight / bias gradients are fp32 outputs of
exact bf16 products: relative L2 <= 2e-5 (summation order only)."""
import math
import zlib

import pytest
import torch
import torch.nn.functional as F

import csmri_oracle as O

pytestmark = pytest.mark.gpu

# name, cin, cout, k, stride, border, upsample, H, W, B, c0 (channels of the first concat source), checks
S = 256
CASES = [
    # frozen RecNet / C2 training (B = 8)
    ('recnet_first', 2, 32, 3, 1, 'zero', False, S, S, 8, None, 'fw'),
    ('recnet_mid', 32, 32, 3, 1, 'zero', False, S, S, 8, None, 'fdw'),
    ('recnet_last', 32, 2, 3, 1, 'zero', False, S, S, 8, None, 'fdw'),
    # U-Net (B = 8)
    ('unet_e0a', 2, 32, 4, 1, 'reflection', False, S, S, 8, None, 'fw'),
    ('unet_e0b', 32, 32, 4, 1, 'reflection', False, S, S, 8, None, 'fdw'),
    ('unet_e1a', 32, 64, 4, 1, 'reflection', False, S // 2, S // 2, 8, None, 'fdw'),
    ('unet_e1b', 64, 64, 4, 1, 'reflection', False, S // 2, S // 2, 8, None, 'fdw'),
    ('unet_e2a', 64, 128, 4, 1, 'reflection', False, S // 4, S // 4, 8, None, 'fdw'),
    ('unet_e2b', 128, 128, 4, 1, 'reflection', False, S // 4, S // 4, 8, None, 'fdw'),
    ('unet_d1up', 128, 64, 4, 1, 'reflection', True, S // 4, S // 4, 8, None, 'fdw'),
    ('unet_d1cat', 128, 64, 4, 1, 'reflection', False, S // 2, S // 2, 8, 64, 'fdw'),
    ('unet_d1b', 64, 64, 4, 1, 'reflection', False, S // 2, S // 2, 8, None, 'fdw'),
    ('unet_d0up', 64, 32, 4, 1, 'reflection', True, S // 2, S // 2, 8, None, 'fdw'),
    ('unet_d0cat', 64, 32, 4, 1, 'reflection', False, S, S, 8, 32, 'fdw'),
    ('unet_head', 32, 1, 1, 1, 'zero', False, S, S, 8, None, 'fdw'),
    # discriminator: grouped D-phase pass (16) with weight gradients, generator-phase pass (8) data gradient only
    ('disc1_b16', 1, 64, 4, 2, 'reflection', False, S, S, 16, None, 'fw'),
    ('disc2_b16', 64, 128, 4, 2, 'reflection', False, S // 2, S // 2, 16, None, 'fdw'),
    ('disc3_b16', 128, 256, 4, 2, 'reflection', False, S // 4, S // 4, 16, None, 'fdw'),
    ('disc4_b16', 256, 512, 4, 2, 'reflection', False, S // 8, S // 8, 16, None, 'fdw'),
    ('disc5_b16', 512, 1024, 4, 2, 'reflection', False, S // 16, S // 16, 16, None, 'fdw'),
    ('disc6_b16', 1024, 1024, 4, 1, 'reflection', False, S // 32, S // 32, 16, None, 'fdw'),
    ('disc_final_b16', 1024, 1, 4, 1, 'none', False, S // 32, S // 32, 16, None, 'fdw'),
    ('disc1_b8', 1, 64, 4, 2, 'reflection', False, S, S, 8, None, 'fd'),
    ('disc2_b8', 64, 128, 4, 2, 'reflection', False, S // 2, S // 2, 8, None, 'fd'),
    ('disc3_b8', 128, 256, 4, 2, 'reflection', False, S // 4, S // 4, 8, None, 'fd'),
    ('disc4_b8', 256, 512, 4, 2, 'reflection', False, S // 8, S // 8, 8, None, 'fd'),
    ('disc5_b8', 512, 1024, 4, 2, 'reflection', False, S // 16, S // 16, 8, None, 'fd'),
    ('disc6_b8', 1024, 1024, 4, 1, 'reflection', False, S // 32, S // 32, 8, None, 'fd'),
    ('disc_final_b8', 1024, 1, 4, 1, 'none', False, S // 32, S // 32, 8, None, 'fd'),
    # discriminator, round 4: ONE forward over [pool-fake; real; current-fake] (24 images); the D-phase backward runs
    # on its first 16 images (cases *_b16 above: 'dw'), the generator-phase data gradient on the last 8 (*_b8: 'd')
    ('disc1_b24', 1, 64, 4, 2, 'reflection', False, S, S, 24, None, 'f'),
    ('disc2_b24', 64, 128, 4, 2, 'reflection', False, S // 2, S // 2, 24, None, 'f'),
    ('disc3_b24', 128, 256, 4, 2, 'reflection', False, S // 4, S // 4, 24, None, 'f'),
    ('disc4_b24', 256, 512, 4, 2, 'reflection', False, S // 8, S // 8, 24, None, 'f'),
    ('disc5_b24', 512, 1024, 4, 2, 'reflection', False, S // 16, S // 16, 24, None, 'f'),
    ('disc6_b24', 1024, 1024, 4, 1, 'reflection', False, S // 32, S // 32, 24, None, 'f'),
    ('disc_final_b24', 1024, 1, 4, 1, 'none', False, S // 32, S // 32, 24, None, 'f'),
    # VGG19: [pred; target] forward (16), data gradient on the prediction half (8)
    ('vgg1_1', 3, 64, 3, 1, 'zero', False, S, S, 16, None, 'f'),
    ('vgg1_2', 64, 64, 3, 1, 'zero', False, S, S, 16, None, 'f'),
    ('vgg2_1', 64, 128, 3, 1, 'zero', False, S // 2, S // 2, 16, None, 'f'),
    ('vgg2_2', 128, 128, 3, 1, 'zero', False, S // 2, S // 2, 16, None, 'f'),
    ('vgg3_1', 128, 256, 3, 1, 'zero', False, S // 4, S // 4, 16, None, 'f'),
    ('vgg3_2', 256, 256, 3, 1, 'zero', False, S // 4, S // 4, 16, None, 'f'),
    ('vgg4_1', 256, 512, 3, 1, 'zero', False, S // 8, S // 8, 16, None, 'f'),
    ('vgg4_2', 512, 512, 3, 1, 'zero', False, S // 8, S // 8, 16, None, 'f'),
    ('vgg5_1', 512, 512, 3, 1, 'zero', False, S // 16, S // 16, 16, None, 'f'),
    ('vgg1_1_b8', 3, 64, 3, 1, 'zero', False, S, S, 8, None, 'd'),
    ('vgg1_2_b8', 64, 64, 3, 1, 'zero', False, S, S, 8, None, 'd'),
    ('vgg2_1_b8', 64, 128, 3, 1, 'zero', False, S // 2, S // 2, 8, None, 'd'),
    ('vgg2_2_b8', 128, 128, 3, 1, 'zero', False, S // 2, S // 2, 8, None, 'd'),
    ('vgg3_1_b8', 128, 256, 3, 1, 'zero', False, S // 4, S // 4, 8, None, 'd'),
    ('vgg3_2_b8', 256, 256, 3, 1, 'zero', False, S // 4, S // 4, 8, None, 'd'),
    ('vgg4_1_b8', 256, 512, 3, 1, 'zero', False, S // 8, S // 8, 8, None, 'd'),
    ('vgg4_2_b8', 512, 512, 3, 1, 'zero', False, S // 8, S // 8, 8, None, 'd'),
    ('vgg5_1_b8', 512, 512, 3, 1, 'zero', False, S // 16, S // 16, 8, None, 'd'),
]

SEEN = {}      # case name -> list of (kind, kernel instance, splitk)


@pytest.fixture(scope='module')
def hip():
  import csmri_hip
  assert torch.cuda.is_available()
  return csmri_hip


def to_dev_nhwc(x, cp=None):
  b, c, h, w = x.shape
  cp = cp or (c + 7) // 8 * 8
  t = torch.zeros(b, h, w, cp, dtype=torch.bfloat16)
  t[..., :c] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
  return t.cuda()


def from_dev_nhwc(t, c):
  return t.float().cpu()[..., :c].permute(0, 3, 1, 2).contiguous()


def rel_l2(a, b):
  return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_bench_layer_vs_oracle_bf16(hip, case):
  ops = hip.ops
  name, cin, cout, k, stride, border, up, h, w, b, c0, checks = case
  g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 100000)
  wt = (torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).bfloat16().float()
  bias = torch.randn(cout, generator=g) * 0.1
  x = torch.randn(b, cin, h, w, generator=g).bfloat16().float()
  pads, mode = ((0, 0, 0, 0), 'zero') if border == 'none' else (O.same_padding(k, stride), border)
  wd = torch.nn.Parameter(wt.clone().cuda())
  bd = torch.nn.Parameter(bias.clone().cuda())
  layer = ops.ConvLayer(wd, bd, stride, pads, mode, torch.bfloat16, upsample=up)
  if c0 is None:
    x0, x1 = to_dev_nhwc(x), None
  else:
    x0, x1 = to_dev_nhwc(x[:, :c0]), to_dev_nhwc(x[:, c0:])
  log = ops.LAUNCH_LOG = []
  try:
    # oracle (CPU, fp32 on the bf16-rounded operands)
    xr = x.clone().requires_grad_('d' in checks)
    wr = wt.clone().requires_grad_('w' in checks)
    br = bias.clone().requires_grad_('w' in checks)
    xin = F.interpolate(xr, scale_factor=2, mode='nearest') if up else xr
    yr = F.conv2d(O.pad2d(xin, pads, mode), wr, br, stride=stride)
    ho, wo = yr.shape[2], yr.shape[3]
    # the forward as the benchmarked step runs this layer family: U-Net / discriminator layers under BatchNorm -- no bias,
    # BatchNorm partial sums out of the epilogue (reference models/unet.py:48, models/discriminators.py:140-143); VGG --
    # bias + ReLU (models/vgg.py:35); the rest -- bias, no activation
    bn = (name.startswith('unet') and name != 'unet_head') or (name.startswith('disc') and name[:5] not in ('disc1', 'disc_'))
    vgg = name.startswith('vgg')
    if 'f' in checks:
      if bn:
        y, st = ops.conv_forward(layer, x0, x1, False, 1.0, True, None)
        ref = (yr - br.view(1, -1, 1, 1)).detach()
      elif vgg:
        y, st = ops.conv_forward(layer, x0, x1, True, 0.0, False, None)
        ref = torch.relu(yr.detach())
      else:
        y, st = ops.conv_forward(layer, x0, x1, True, 1.0, False, None)
        ref = yr.detach()
      torch.cuda.synchronize()
      err = rel_l2(from_dev_nhwc(y, cout), ref)
      print('%-16s fwd   rel_l2 %.3e  %s' % (name, err, log[-1][1:]))
      assert err < 3e-3, (name, 'fwd', err)
      if y.shape[3] > cout:
        assert float(y[..., cout:].float().abs().max()) == 0.0
      if bn:
        # partial rows [2][Cout_pad][rows] of the fp32 accumulators: per-channel sum and sum of squares
        part = st.reshape(2, layer.cout_p, -1).sum(-1).cpu()
        s_ref, q_ref = ref.sum((0, 2, 3)), (ref * ref).sum((0, 2, 3))
        e1 = float((part[0, :cout] - s_ref).abs().max() / (ref.abs().sum((0, 2, 3)).max() + 1e-30))
        e2 = rel_l2(part[1, :cout], q_ref)
        print('%-16s stats sum %.2e  sumsq %.2e' % (name, e1, e2))
        assert e1 < 1e-4 and e2 < 1e-4, (name, 'stats', e1, e2)
    if 'd' in checks or 'w' in checks:
      gy = torch.randn(b, cout, ho, wo, generator=g).bfloat16().float()
      yr.backward(gy)
      gyd = to_dev_nhwc(gy, layer.cout_p)
      if 'd' in checks:
        n0 = len(log)
        # VGG conv*_2 (.._3, .._4): gated by the producer's ReLU, whose output is this layer's input (models/vgg.py:35
        # backward); conv*_1 follow a max-pool (the pool's backward applies the ReLU derivative) or the image: plain
        if vgg and not name.split('_')[1] == '1':
          gx = ops.conv_dgrad(layer, gyd, (h, w), g_src=x0, g_slope=0.0)
          gref = xr.grad * (x > 0).float()
        else:
          gx = ops.conv_dgrad(layer, gyd, (h, w))
          gref = xr.grad
        torch.cuda.synchronize()
        err = rel_l2(from_dev_nhwc(gx, cin), gref)
        print('%-16s dgrad rel_l2 %.3e  %s' % (name, err, [e[1:] for e in log[n0:]]))
        assert err < 3e-3, (name, 'dgrad', err)
      if 'w' in checks:
        n0 = len(log)
        ops.conv_wgrad(layer, x0, x1, gyd, accumulate=False)
        ops.join_wgrad_stream()
        torch.cuda.synchronize()
        ew, eb = rel_l2(wd.grad.cpu(), wr.grad), rel_l2(bd.grad.cpu(), br.grad)
        print('%-16s wgrad rel_l2 %.3e  bias %.3e  %s' % (name, ew, eb, [e[1:] for e in log[n0:]]))
        assert ew < 2e-5, (name, 'wgrad', ew)
        assert eb < 2e-5, (name, 'bgrad', eb)
        # the accumulating form (a second backward pass into the same .grad): twice the gradient
        ops.conv_wgrad(layer, x0, x1, gyd, accumulate=True)
        ops.join_wgrad_stream()
        torch.cuda.synchronize()
        ew2, eb2 = rel_l2(wd.grad.cpu(), 2.0 * wr.grad), rel_l2(bd.grad.cpu(), 2.0 * br.grad)
        assert ew2 < 2e-5 and eb2 < 2e-5, (name, 'wgrad accumulate', ew2, eb2)
  finally:
    ops.LAUNCH_LOG = None
  SEEN[name] = list(log)


def test_dispatch_variants_are_all_exercised():
  """Every kernel instance the BENCHMARKED step launches (the `conv_kernels` table of the committed bench line,
  profiles/r06_bench_n1.json: keys are the instance names rocprofv3 prints) has been reached by a bench shape above,
  i.e. has been compared with the oracle; a dispatch change that strands an instance, or a bench line taken with a
  build whose instances these shapes no longer reach, fails here."""
  import json
  import os
  from conftest import ROOT
  if len(SEEN) < len(CASES):
    pytest.skip('runs after the full parametrized set')
  names = sorted(set(e[1] for v in SEEN.values() for e in v))
  print('\n'.join(names))
  fam = set(n.split('<')[0].replace('void ', '') for n in names)
  splitk = any(e[2] > 1 and e[0] == 'gconv' for v in SEEN.values() for e in v)
  assert splitk, 'no split-K convolution among the bench shapes'
  need = {'tconv_kernel', 'uconv_kernel', 'gconv_kernel', 'gconv_glds_kernel', 'gpipe_kernel', 'pconv2_kernel',
          'thin_out1_tile_kernel', 'wpatch_kernel', 'wrow_kernel', 'wgrad_glds_row_kernel', 'wthin_out_kernel'}
  assert need <= fam, (need - fam, fam)
  path = os.path.join(ROOT, 'profiles', 'r06_bench_n1.json')
  assert os.path.exists(path), 'commit the bench line of this build as profiles/r06_bench_n1.json'
  table = json.load(open(path))['conv_kernels']
  # second stages / fused blocks that are not csmri_gconv / csmri_wgrad main kernels (covered by tests/test_hip_ops.py)
  other = {'gconv_reduce_kernel', 'convblock_fwd_kernel'}
  missing = sorted(k for k in table if k not in other and k not in names)
  assert not missing, 'bench launches kernel instances no bench-shape case reaches: %s' % missing
