"""GPU parity tests of the C-ABI kernels against the CPU oracle (plain torch fp32).

Tolerances: fp32 path rtol 1e-4 / atol 1e-5 (scaled by the reduction magnitude);
bf16 path: relative L2 error <= 1e-2 (inputs and weights are rounded to bf16
before the oracle runs so only accumulation order / output rounding differ).
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import csmri_oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def hip():
  import csmri_hip
  assert torch.cuda.is_available()
  return csmri_hip


def to_dev_nhwc(x, dtype, cp=None):
  """CPU NCHW fp32 -> GPU NHWC padded (built with torch only, independent of our converters)."""
  b, c, h, w = x.shape
  cp = cp or (c + 7) // 8 * 8
  t = torch.zeros(b, h, w, cp, dtype=torch.float32)
  t[..., :c] = x.permute(0, 2, 3, 1)
  return t.to(dtype).cuda()


def from_dev_nhwc(t, c):
  return t.float().cpu()[..., :c].permute(0, 3, 1, 2).contiguous()


def rel_l2(a, b):
  return float((a - b).norm() / (b.norm() + 1e-30))


def check(name, got, ref, dtype):
  err = rel_l2(got, ref)
  mx = float((got - ref).abs().max())
  print('%-40s rel_l2 %.3e  max_abs %.3e  ref_max %.3e' % (name, err, mx, float(ref.abs().max())))
  if dtype == torch.float32:
    assert err < 2e-5, name
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-5 * max(1.0, float(ref.abs().max()))), name
  else:
    assert err < 1e-2, name


CONV_CASES = [
    # name, cin, cout, k, stride, border, upsample, H, W, B
    ('recnet_first', 2, 32, 3, 1, 'zero', False, 32, 32, 2),
    ('recnet_mid', 32, 32, 3, 1, 'zero', False, 32, 48, 2),
    ('recnet_last', 32, 2, 3, 1, 'zero', False, 32, 32, 2),
    ('unet_k4', 32, 64, 4, 1, 'reflection', False, 32, 32, 2),
    ('unet_up', 64, 32, 4, 1, 'reflection', True, 16, 16, 2),
    ('unet_head', 32, 1, 1, 1, 'zero', False, 32, 32, 2),
    ('disc_s2', 16, 128, 4, 2, 'reflection', False, 32, 32, 2),
    ('disc_first', 1, 64, 4, 2, 'reflection', False, 64, 64, 2),
    ('disc_deep', 256, 256, 4, 1, 'reflection', False, 8, 8, 2),
    ('disc_final', 64, 1, 4, 1, 'none', False, 8, 8, 2),
    ('vgg', 64, 128, 3, 1, 'zero', False, 24, 40, 1),
    ('odd_m', 8, 24, 3, 1, 'zero', False, 13, 7, 3),
    ('vgg3_tile256', 256, 256, 3, 1, 'zero', False, 128, 121, 4),
    ('vgg2_patch', 128, 128, 3, 1, 'zero', False, 128, 120, 8),      # 512 tiles of 16x16: pconv2, fwd and dgrad (tile edge at x=120)
    ('vgg2_1_patch2', 64, 128, 3, 1, 'zero', False, 128, 120, 4),     # pconv2 with ONE 64-channel chunk per tile (256 tiles, edge at x=120)
    ('p2_c192', 192, 256, 3, 1, 'zero', False, 64, 60, 8),             # pconv2: three chunks per tile, two channel blocks, partial tiles
    ('vgg4_patch2', 512, 512, 3, 1, 'zero', False, 40, 36, 12),       # 108 tiles x 4 channel blocks: pconv2 (loader waves), fwd and dgrad; partial edge tiles
    ('unet_patch_k4', 128, 128, 4, 1, 'reflection', False, 96, 96, 16),  # a 4x4 reflection-padded 128-channel layer at 1152 row tiles (gconv_glds, one-buffer variant)
    # the discriminator's layer forms on the persistent pipelined gather kernel (gpipe.hip), reduced batches:
    ('gp_l2', 64, 128, 4, 2, 'reflection', False, 128, 128, 4),    # forward 64 tiles; data gradient 4 classes x 67 ragged tiles of 64 channels (two rounds of work items)
    ('gp_l3', 128, 256, 4, 2, 'reflection', False, 32, 32, 6),     # 192-row tiles
    ('gp_l5', 512, 1024, 4, 2, 'reflection', False, 16, 16, 3),    # small M: split-K slabs, forward and per-class data gradient
    ('gp_l6', 1024, 1024, 4, 1, 'reflection', False, 8, 8, 3),     # stride 1 on an 8 x 8 map: 16 taps of 1024 channels, padded data gradient with halo window
]


def make_layer(hip, case, dtype, seed=0):
  name, cin, cout, k, stride, border, up, h, w, b = case
  g = torch.Generator().manual_seed(seed)
  wt = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
  bias = torch.randn(cout, generator=g) * 0.1
  x = torch.randn(b, cin, h, w, generator=g)
  if border == 'none':
    pads, mode = (0, 0, 0, 0), 'zero'
  else:
    pads, mode = O.same_padding(k, stride), border
  if dtype == torch.bfloat16:
    wt, x = wt.bfloat16().float(), x.bfloat16().float()
  wd = torch.nn.Parameter(wt.clone().cuda())
  bd = torch.nn.Parameter(bias.clone().cuda())
  layer = hip.ops.ConvLayer(wd, bd, stride, pads, mode, dtype, upsample=up)
  return layer, wt, bias, x, pads, mode


def ref_conv(x, wt, bias, stride, pads, mode, up, slope):
  if up:
    x = F.interpolate(x, scale_factor=2, mode='nearest')
  y = F.conv2d(O.pad2d(x, pads, mode), wt, bias, stride=stride)
  return F.leaky_relu(y, slope) if slope != 1.0 else y


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_bwd(hip, case, dtype):
  ops = hip.ops
  name, cin, cout, k, stride, border, up, h, w, b = case
  if (name in ('vgg3_tile256', 'vgg2_patch', 'vgg2_1_patch2', 'p2_c192', 'vgg4_patch2', 'unet_patch_k4') or name.startswith('gp_')) and dtype == torch.float32:
    pytest.skip('the 256-row / patch kernels are bf16-only; at 16M outputs the fp32 comparison trips on '
                'LeakyReLU-derivative sign flips of pre-activations at the rounding floor')
  layer, wt, bias, x, pads, mode = make_layer(hip, case, dtype)
  slope = 0.2
  xd = to_dev_nhwc(x, dtype).requires_grad_(True)
  y = ops.ConvAct.apply(xd, None, layer.weight, layer.bias, layer, slope, None)
  xr = x.clone().requires_grad_(True)
  wr = wt.clone().requires_grad_(True)
  br = bias.clone().requires_grad_(True)
  yr = ref_conv(xr, wr, br, stride, pads, mode, up, slope)
  check(name + ' fwd', from_dev_nhwc(y, cout), yr.detach(), dtype)
  # pad channels of the output must be exactly zero
  assert float(y[..., cout:].float().abs().max()) == 0.0 if y.shape[3] > cout else True
  g = torch.randn(yr.shape, generator=torch.Generator().manual_seed(5))
  if dtype == torch.bfloat16:
    g = g.bfloat16().float()
  yr.backward(g)
  y.backward(to_dev_nhwc(g, dtype, y.shape[3]))
  torch.cuda.synchronize()
  check(name + ' dgrad', from_dev_nhwc(xd.grad, cin), xr.grad, dtype)
  check(name + ' wgrad', layer.weight.grad.cpu(), wr.grad, dtype)
  check(name + ' bgrad', layer.bias.grad.cpu(), br.grad, dtype)


@pytest.mark.parametrize('g_slope', [0.0, 0.2], ids=['relu', 'leaky'])
@pytest.mark.parametrize('case', [('refl_k3_c32', 32, 32, 3, 40, 36, 2), ('refl_k4_c128', 128, 128, 4, 16, 16, 4),
                                  ('refl_k4_c64_big', 64, 64, 4, 64, 64, 2)], ids=lambda c: c[0])
def test_reflection_dgrad_gated_window_vs_oracle(hip, case, g_slope):
  """Data gradient of a reflection-padded stride-1 layer WITH the producer's activation derivative fused
  (conv_dgrad(g_src=, g_slope=): models/recnet.py:40-48 with padding='reflection', models/unet.py:48): the kernel
  writes the un-padded centre gated and the halo positions UNGATED (include/csmri_hip.h, out_halo); csmri_fold_halo
  mirrors the halo back and gates it where it lands.  A halo lane that was gated twice loses the mirrored
  contribution (ReLU) or scales it by the slope (leaky): round-5 advisor finding.  Checked against CPU autograd and
  against the FOLD_WINDOW = False path (full padded gradient + fold kernel)."""
  ops = hip.ops
  name, cin, cout, k, h, w, b = case
  dtype = torch.bfloat16
  layer, wt, bias, x, pads, mode = make_layer(hip, (name, cin, cout, k, 1, 'reflection', False, h, w, b), dtype)
  gen = torch.Generator().manual_seed(11)
  gy = torch.randn(b, cout, h, w, generator=gen).bfloat16().float()
  gs = torch.randn(b, cin, h, w, generator=gen).bfloat16().float()
  xr = x.clone().requires_grad_(True)
  F.conv2d(O.pad2d(xr, pads, mode), wt, None).backward(gy)
  ref = xr.grad * torch.where(gs > 0, torch.ones_like(gs), torch.full_like(gs, g_slope))
  gyd, gsd = to_dev_nhwc(gy, dtype), to_dev_nhwc(gs, dtype)
  log = ops.LAUNCH_LOG = []
  try:
    got = ops.conv_dgrad(layer, gyd, (h, w), g_src=gsd, g_slope=g_slope)
  finally:
    ops.LAUNCH_LOG = None
  old = ops.FOLD_WINDOW
  ops.FOLD_WINDOW = False
  try:
    plain = ops.conv_dgrad(layer, gyd, (h, w), g_src=gsd, g_slope=g_slope)
  finally:
    ops.FOLD_WINDOW = old
  torch.cuda.synchronize()
  print(name, [e[1] for e in log])
  check(name + ' gated window dgrad', from_dev_nhwc(got, cin), ref, dtype)
  check(name + ' gated full-pad dgrad', from_dev_nhwc(plain, cin), ref, dtype)
  # the border rows / columns are where the halo lands: they alone must agree as well
  gb, rb = from_dev_nhwc(got, cin), ref
  for sl in ((slice(None), slice(None), slice(0, 2)), (slice(None), slice(None), slice(h - 3, h)),
             (slice(None), slice(None), slice(None), slice(0, 2)), (slice(None), slice(None), slice(None), slice(w - 3, w))):
    assert rel_l2(gb[sl], rb[sl]) < 1e-2, (name, sl)


GPIPE_CASES = [
    # name, cin, cout, stride, H, W, B[, kernel size, border]
    ('l2', 64, 128, 2, 128, 128, 4), ('l3', 128, 256, 2, 64, 64, 3), ('l4', 256, 512, 2, 32, 32, 5),
    ('l5', 512, 1024, 2, 16, 16, 5), ('l6', 1024, 1024, 1, 8, 8, 5), ('l4_ragged', 256, 256, 2, 22, 26, 3),
    # what else the dispatcher may hand it: 3 x 3 zero-padded stride 1 (tap index by multiplication, out-of-image taps
    # through the zero page, a direct data gradient without window), maps that are no powers of two (magic-number divisions)
    ('k3_zero', 128, 64, 1, 24, 20, 3, 3, 'zero'), ('k3_zero_deep', 256, 128, 1, 12, 12, 4, 3, 'zero'),
]


@pytest.mark.parametrize('case', GPIPE_CASES, ids=[c[0] for c in GPIPE_CASES])
def test_gpipe_discriminator_layer_forms(hip, case):
  """The persistent pipelined gather kernel (gpipe.hip) on every form the discriminator's layers 2-6 launch it in
  (reference models/discriminators.py:137-172: [ReflectionPad -> Conv4x4 stride 2] x 5, [ReflectionPad(1,2,1,2) -> Conv4x4
  stride 1]), against CPU autograd on the same bf16-rounded operands:
    forward with the BatchNorm partial sums from the epilogue (the sums against the CPU output's, 1e-3 relative),
    forward without (split-K slabs where the plan splits), the data gradient of a stride-2 layer as four output-parity
    classes, of the stride-1 layer on its padded extent, both through the halo window + fold, ungated and gated by the
    producer's LeakyReLU(0.2) (the gradient that reaches layer 1 is gated: discriminators.py:144).
  The launch log must name gpipe_kernel for every one of them.  Tolerance: relative L2 <= 1e-2 (bf16 output rounding is
  ~2e-3; the reductions are fp32)."""
  ops = hip.ops
  name, cin, cout, stride, h, w, b = case[:7]
  k, border = (case[7], case[8]) if len(case) > 7 else (4, 'reflection')
  dtype = torch.bfloat16
  layer, wt, bias, x, pads, mode = make_layer(hip, (name, cin, cout, k, stride, border, False, h, w, b), dtype)
  xd = to_dev_nhwc(x, dtype)
  log = ops.LAUNCH_LOG = []
  old_flags, ops.GCONV_FLAGS = ops.GCONV_FLAGS, 2          # CSMRI_GCONV_USE_GPIPE: wherever it is eligible, not only from 32 K steps on
  try:
    ref = F.conv2d(O.pad2d(x, pads, mode), wt, None, stride=stride)
    m = ref.shape[0] * ref.shape[2] * ref.shape[3]
    y, _ = ops.conv_forward(layer, xd, None, False)
    check(name + ' fwd', from_dev_nhwc(y, cout), ref, dtype)
    if m % 64 == 0:
      ys, stats = ops.conv_forward(layer, xd, None, False, 1.0, True)
      torch.cuda.synchronize()
      check(name + ' fwd (stats form)', from_dev_nhwc(ys, cout), ref, dtype)
      st = stats.cpu().reshape(-1)
      rows = st.numel() // (2 * y.shape[3])
      st = st.view(2, y.shape[3], rows).double().sum(2)           # [2][Cout][rows]
      s1, s2 = ref.double().sum((0, 2, 3)), (ref.double() ** 2).sum((0, 2, 3))
      assert float((st[0, :cout] - s1).abs().max()) < 1e-3 * float(s1.abs().max() + 1.0), name
      assert float((st[1, :cout] - s2).abs().max()) < 1e-3 * float(s2.abs().max()), name
    gen = torch.Generator().manual_seed(3)
    gy = torch.randn(ref.shape, generator=gen).bfloat16().float()
    gs = torch.randn(x.shape, generator=gen).bfloat16().float()
    xr = x.clone().requires_grad_(True)
    F.conv2d(O.pad2d(xr, pads, mode), wt, None, stride=stride).backward(gy)
    gyd, gsd = to_dev_nhwc(gy, dtype), to_dev_nhwc(gs, dtype)
    dx = ops.conv_dgrad(layer, gyd, (h, w))
    check(name + ' dgrad', from_dev_nhwc(dx, cin), xr.grad, dtype)
    dxg = ops.conv_dgrad(layer, gyd, (h, w), g_src=gsd, g_slope=0.2)
    gated = xr.grad * torch.where(gs > 0, torch.ones_like(gs), torch.full_like(gs, 0.2))
    check(name + ' dgrad gated', from_dev_nhwc(dxg, cin), gated, dtype)
  finally:
    ops.LAUNCH_LOG = None
    ops.GCONV_FLAGS = old_flags
  names = [e[1] for e in log if e[0] == 'gconv']
  print(name, sorted(set((e[1], e[2]) for e in log if e[0] == 'gconv')))
  if k == 4:
    assert names and all(n.startswith('gpipe_kernel') for n in names), names
  else:       # (a patch kernel ahead of it in the dispatch may take the 3 x 3 forms: at least one launch must be gpipe's)
    assert any(n.startswith('gpipe_kernel') for n in names), names


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
def test_conv_two_source_concat(hip, dtype):
  ops = hip.ops
  g = torch.Generator().manual_seed(3)
  b, h, w = 2, 16, 16
  xa, xb = torch.randn(b, 16, h, w, generator=g), torch.randn(b, 24, h, w, generator=g)
  wt = torch.randn(40, 40, 4, 4, generator=g) / math.sqrt(40 * 16)
  if dtype == torch.bfloat16:
    xa, xb, wt = xa.bfloat16().float(), xb.bfloat16().float(), wt.bfloat16().float()
  pads = O.same_padding(4, 1)
  layer = ops.ConvLayer(torch.nn.Parameter(wt.cuda()), None, 1, pads, 'reflection', dtype)
  # sources are channel slices of wider buffers (pixel stride > channels)
  wide_a = torch.zeros(b, h, w, 32, dtype=dtype, device='cuda')
  wide_a[..., 8:24] = to_dev_nhwc(xa, dtype)
  a = wide_a[..., 8:24].requires_grad_(True)
  bb = to_dev_nhwc(xb, dtype).requires_grad_(True)
  y = ops.ConvAct.apply(a, bb, layer.weight, None, layer, 1.0, None)
  xr = torch.cat((xa, xb), 1).requires_grad_(True)
  wr = wt.clone().requires_grad_(True)
  yr = F.conv2d(O.pad2d(xr, pads, 'reflection'), wr)
  check('concat fwd', from_dev_nhwc(y, 40), yr.detach(), dtype)
  gg = torch.randn(yr.shape, generator=g)
  if dtype == torch.bfloat16:
    gg = gg.bfloat16().float()
  yr.backward(gg)
  y.backward(to_dev_nhwc(gg, dtype))
  check('concat dgrad a', from_dev_nhwc(a.grad, 16), xr.grad[:, :16], dtype)
  check('concat dgrad b', from_dev_nhwc(bb.grad, 24), xr.grad[:, 16:], dtype)
  check('concat wgrad', layer.weight.grad.cpu(), wr.grad, dtype)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(2, 16, 32, 64, 64, 1), (2, 64, 128, 8, 8, 1), (2, 8, 16, 64, 64, 2),
                                   (2, 8, 8, 32, 32, 1), (8, 32, 64, 32, 32, 2), (4, 64, 512, 16, 16, 2),
                                   (2, 32, 1024, 8, 8, 1)],
                         ids=['stats_fused', 'stats_standalone', 'c16_s2', 'c8', 'c64_s2', 'small_c512_s2',
                              'small_c1024'])
def test_conv_bn_act(hip, dtype, shape):
  ops = hip.ops
  b, cin, cout, h, w, stride = shape
  g = torch.Generator().manual_seed(11)
  x = torch.randn(b, cin, h, w, generator=g)
  wt = torch.randn(cout, cin, 4, 4, generator=g) / math.sqrt(cin * 16)
  gamma = 1 + 0.1 * torch.randn(cout, generator=g)
  beta = 0.1 * torch.randn(cout, generator=g)
  keep = torch.bernoulli(torch.full((b, cout), 0.5), generator=g) * 2.0
  if dtype == torch.bfloat16:
    x, wt = x.bfloat16().float(), wt.bfloat16().float()
  pads = O.same_padding(4, stride)
  layer = ops.ConvLayer(torch.nn.Parameter(wt.cuda()), None, stride, pads, 'reflection', dtype)
  bn = ops.BNState(torch.nn.Parameter(gamma.cuda()), torch.nn.Parameter(beta.cuda()),
                   torch.zeros(cout).cuda(), torch.ones(cout).cuda())
  xd = to_dev_nhwc(x, dtype).requires_grad_(True)
  z = ops.ConvBnAct.apply(xd, None, layer.weight, bn.weight, bn.bias, layer, bn, 0.2, True, keep.cuda())
  xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
  gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
  rm, rv = torch.zeros(cout), torch.ones(cout)
  yr = F.conv2d(O.pad2d(xr, pads, 'reflection'), wr, stride=stride)
  zr = F.leaky_relu(F.batch_norm(yr, rm, rv, gr, br, True, 0.1, 1e-5), 0.2) * keep[:, :, None, None]
  tol_dtype = dtype
  check('bn fwd', from_dev_nhwc(z, cout), zr.detach(), tol_dtype)
  check('bn running_mean', bn.running_mean.cpu(), rm, torch.float32 if dtype == torch.float32 else dtype)
  check('bn running_var', bn.running_var.cpu(), rv, torch.float32 if dtype == torch.float32 else dtype)
  gz = torch.randn(zr.shape, generator=g)
  if dtype == torch.bfloat16:
    gz = gz.bfloat16().float()
  zr.backward(gz)
  z.backward(to_dev_nhwc(gz, dtype))
  # bf16 stores y,z rounded, so BN backward sees slightly different operands
  lo = torch.float32 if dtype == torch.float32 else torch.bfloat16
  check('bn dgrad', from_dev_nhwc(xd.grad, cin), xr.grad, lo)
  check('bn wgrad', layer.weight.grad.cpu(), wr.grad, lo)
  check('bn dgamma', bn.weight.grad.cpu(), gr.grad, lo)
  check('bn dbeta', bn.bias.grad.cpu(), br.grad, lo)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(8, 64, 1024, 16, 16), (8, 512, 512, 4, 4), (8, 32, 16, 16, 16), (2, 128, 256, 8, 8),
                                   (3, 32, 64, 24, 24)],
                         ids=['8x8x1024', '2x2x512', '8x8x16', '4x4x256', '12x12x64_b3'])
def test_small_map_batchnorm_one_launch(hip, dtype, shape):
  """csmri_bn_small_fwd / csmri_bn_small_bwd (one launch each for the inner U-Net layers, reference models/unet.py
  :60-108) against the three-launch sequences they replace: output, saved statistics, running statistics, input /
  weight / affine gradients.  fp32: agreement to summation-order rounding; bf16: within two roundings of the stored
  tensors."""
  ops = hip.ops
  b, cin, cout, h, w = shape
  g = torch.Generator().manual_seed(21)
  x = torch.randn(b, cin, h, w, generator=g)
  wt = torch.randn(cout, cin, 4, 4, generator=g) / math.sqrt(cin * 16)
  gamma, beta = 1 + 0.1 * torch.randn(cout, generator=g), 0.1 * torch.randn(cout, generator=g)
  keep = (torch.bernoulli(torch.full((b, ops.pad8(cout)), 0.5), generator=g) * 2.0).cuda()
  gz = torch.randn(b, cout, h // 2, w // 2, generator=g)
  assert ops.lib.raw('csmri_bn_small_ok')(b * (h // 2) * (w // 2), ops.pad8(cout), 1)
  res = {}
  for small in (True, False):
    ops.BN_SMALL = small
    log = ops.LAUNCH_LOG = []
    try:
      layer = ops.ConvLayer(torch.nn.Parameter(wt.clone().cuda()), None, 2, O.same_padding(4, 2), 'reflection', dtype)
      bn = ops.BNState(torch.nn.Parameter(gamma.clone().cuda()), torch.nn.Parameter(beta.clone().cuda()),
                       torch.zeros(cout).cuda(), torch.ones(cout).cuda())
      xd = to_dev_nhwc(x, dtype).requires_grad_(True)
      z = ops.ConvBnAct.apply(xd, None, layer.weight, bn.weight, bn.bias, layer, bn, 0.2, True, keep)
      z.backward(to_dev_nhwc(gz, dtype))
      ops.join_wgrad_stream()
      torch.cuda.synchronize()
    finally:
      ops.BN_SMALL, ops.LAUNCH_LOG = True, None
    res[small] = dict(z=from_dev_nhwc(z.detach(), cout), rm=bn.running_mean.cpu(), rv=bn.running_var.cpu(),
                      dx=from_dev_nhwc(xd.grad, cin), dw=layer.weight.grad.cpu(), dg=bn.weight.grad.cpu(),
                      db=bn.bias.grad.cpu())
  tol = 2e-5 if dtype == torch.float32 else 2e-2
  for k in res[True]:
    a, r = res[True][k], res[False][k]
    err = (a - r).abs().max().item() / max(r.abs().max().item(), 1e-6)
    print(k, 'one launch vs three: rel err', err)
    assert err <= tol, (k, err)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
def test_replay_nodes_sum_two_gradients_in_kernel(hip, dtype):
  """The ``tap`` form of ConvBnActReplay / ConvActReplay returns two aliases of the layer output (next layer,
  feature-matching loss: reference models/discriminators.py:118-126 hands the same tensor to both) and sums their
  gradients inside csmri_bn_bwd_* / csmri_act_bwd (``dz2``).  Against the untapped node fed autograd's own sum:
  identical in fp32, within one bf16 rounding of the sum in bf16 (the fused form adds in fp32, autograd's add kernel
  rounds the sum to bf16 first)."""
  ops = hip.ops
  g = torch.Generator().manual_seed(5)
  b, cin, cout, h, w = 4, 16, 32, 24, 24
  x = torch.randn(b, cin, h, w, generator=g)
  wt = torch.randn(cout, cin, 4, 4, generator=g) / math.sqrt(cin * 16)
  pads = O.same_padding(4, 2)
  res = {}
  for kind in ('bn', 'act'):
    for tap in (False, True):
      layer = ops.ConvLayer(torch.nn.Parameter(wt.clone().cuda()), torch.nn.Parameter(torch.zeros(cout).cuda())
                            if kind == 'act' else None, 2, pads, 'reflection', dtype)
      xd = to_dev_nhwc(x, dtype).requires_grad_(True)
      if kind == 'bn':
        bn = ops.BNState(torch.nn.Parameter(torch.ones(cout).cuda()), torch.nn.Parameter(torch.zeros(cout).cuda()),
                         torch.zeros(cout).cuda(), torch.ones(cout).cuda())
        with torch.no_grad():
          rec = ops.ConvBnAct.run_forward(xd.detach(), None, layer, bn, 0.2, True, None, 1)
        out = ops.ConvBnActReplay.apply(xd, None, layer.weight, bn.weight, bn.bias, layer, bn, 0.2, None, rec, b, 0, 1,
                                        True, tap)
      else:
        with torch.no_grad():
          y, _ = ops.conv_forward(layer, xd.detach(), None, True, 0.2, False, None)
        out = ops.ConvActReplay.apply(xd, None, layer.weight, layer.bias, layer, 0.2, [y], True, tap)
      za, zb = out if tap else (out, out)
      assert za.data_ptr() == zb.data_ptr()
      ga = torch.randn(za.shape, generator=torch.Generator().manual_seed(8)).to(dtype).cuda()
      gb = torch.randn(za.shape, generator=torch.Generator().manual_seed(9)).to(dtype).cuda()
      ((za.float() * ga.float()).sum() + (zb.float() * gb.float()).sum()).backward()
      ops.join_wgrad_stream()
      torch.cuda.synchronize()
      res[kind, tap] = (xd.grad.float().cpu(), layer.weight.grad.float().cpu())
    for i, name in enumerate(('dx', 'dw')):
      a, r = res[kind, True][i], res[kind, False][i]
      err = (a - r).abs().max().item() / r.abs().max().item()
      print(kind, name, 'tap vs autograd sum: rel err', err)
      assert err <= (1e-6 if dtype == torch.float32 else 1.2e-2), (kind, name, err)
  # a tapped node whose second alias has no consumer
  layer = ops.ConvLayer(torch.nn.Parameter(wt.clone().cuda()), torch.nn.Parameter(torch.zeros(cout).cuda()), 2, pads,
                        'reflection', dtype)
  xd = to_dev_nhwc(x, dtype).requires_grad_(True)
  with torch.no_grad():
    y, _ = ops.conv_forward(layer, xd.detach(), None, True, 0.2, False, None)
  za, zb = ops.ConvActReplay.apply(xd, None, layer.weight, layer.bias, layer, 0.2, [y], True, True)
  gb = torch.randn(za.shape, generator=torch.Generator().manual_seed(9)).to(dtype).cuda()
  (zb.float() * gb.float()).sum().backward()
  xd2 = to_dev_nhwc(x, dtype).requires_grad_(True)
  z1 = ops.ConvActReplay.apply(xd2, None, layer.weight, layer.bias, layer, 0.2, [y], False)
  (z1.float() * gb.float()).sum().backward()
  torch.cuda.synchronize()
  assert torch.equal(xd.grad, xd2.grad)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
def test_maxpool(hip, dtype):
  ops = hip.ops
  x = torch.randn(2, 16, 12, 20, generator=torch.Generator().manual_seed(2))
  if dtype == torch.bfloat16:
    x = x.bfloat16().float()
  xd = to_dev_nhwc(x, dtype).requires_grad_(True)
  y = ops.MaxPool2.apply(xd)
  xr = x.clone().requires_grad_(True)
  yr = F.max_pool2d(xr, 2, 2)
  assert torch.equal(from_dev_nhwc(y, 16), yr.detach())
  g = torch.randn(yr.shape, generator=torch.Generator().manual_seed(4))
  if dtype == torch.bfloat16:
    g = g.bfloat16().float()
  yr.backward(g)
  y.backward(to_dev_nhwc(g, dtype))
  assert torch.equal(from_dev_nhwc(xd.grad, 16), xr.grad)


def test_wgrad_finish_multi_equals_per_layer_reductions(hip, monkeypatch):
  """csmri_wgrad_finish_multi (one launch for the slab reductions of every layer of a backward pass) against the
  per-layer reductions inside csmri_wgrad: bit-identical weight and bias gradients -- patch kernels (bias partials in
  the slab), row kernels (transposing reduction, bias by column sums), accumulation onto existing gradients, and a
  layer used twice in one backward pass (the queue is flushed in between)."""
  ops = hip.ops
  specs = [(2, 32, 3, 128, 128), (32, 32, 3, 128, 128), (32, 2, 3, 128, 128), (64, 128, 4, 32, 32), (128, 256, 4, 16, 16)]

  def run(mode):
    monkeypatch.setattr(ops, 'WGRAD_FINISH_MULTI', mode)
    gen = torch.Generator().manual_seed(33)
    grads, log = [], []
    layers, xs = [], []
    for cin, cout, k, h, w in specs:
      wt = torch.nn.Parameter((torch.randn(cout, cin, k, k, generator=gen) * 0.05).cuda())
      bi = torch.nn.Parameter((torch.randn(cout, generator=gen) * 0.1).cuda())
      wt.grad = torch.full_like(wt, 0.25)                      # accumulate onto something
      bi.grad = torch.full_like(bi, -0.5)
      pads = (1, 1, 1, 1) if k == 3 else (1, 2, 1, 2)
      layers.append(ops.ConvLayer(wt, bi, 1, pads, 'zero', torch.bfloat16))
      xs.append(to_dev_nhwc(torch.randn(8, cin, h, w, generator=gen), torch.bfloat16).requires_grad_(True))
    ops.LAUNCH_LOG = log
    try:
      outs = [ops.ConvAct.apply(x, None, l.weight, l.bias, l, 0.2, None) for l, x in zip(layers, xs)]
      outs.append(ops.ConvAct.apply(xs[1].detach() * 0.5, None, layers[1].weight, layers[1].bias, layers[1], 0.2, None))
      total = sum((o.float() * o.float()).mean() for o in outs)
      total.backward()
      ops.join_wgrad_stream()
      torch.cuda.synchronize()
    finally:
      ops.LAUNCH_LOG = None
    for l in layers:
      grads += [l.weight.grad.clone(), l.bias.grad.clone()]
    return grads, [e[1] for e in log if e[0] == 'wgrad']
  g0, k0 = run('0')
  g1, k1 = run('1')
  assert k0 == k1 and any('wpatch' in k for k in k0) and any('wgrad_glds' in k for k in k0), k0
  for a, b in zip(g0, g1):
    assert torch.equal(a, b)
  g2, _ = run('auto')                               # 'auto' without a side stream = '1'
  for a, b in zip(g0, g2):
    assert torch.equal(a, b)


def test_flat_adam_lazy_zero_only_skips_kernel_written_gradients(hip):
  """FlatAdam.lazy_zero: zero_grad() only MARKS the gradients the library's kernels write (they overwrite on their
  first launch); a parameter whose gradient torch autograd accumulates (the refinement wrapper's scale) is still
  zeroed, and a marked gradient nobody wrote is zeroed in apply() before the update."""
  from training.optimizers import FlatAdam      # (the package directory is on sys.path: tests/conftest.py)
  w = torch.nn.Parameter(torch.randn(8, 8, 3, 3).cuda())
  s = torch.nn.Parameter(torch.ones(1).cuda())
  opt = FlatAdam([w, s], lr=1e-3)
  opt.lazy_zero = True
  w.grad.fill_(3.0); s.grad.fill_(5.0)
  opt.zero_grad()                              # nothing known about the writers yet: everything zeroed
  assert float(w.grad.abs().max()) == 0.0 and float(s.grad.abs().max()) == 0.0
  w._kernel_grad = True                        # (what ops.conv_wgrad sets on the weights it writes)
  w.grad.fill_(3.0); s.grad.fill_(5.0)
  opt.zero_grad()
  assert getattr(w, '_grad_fresh', False) and float(w.grad.min()) == 3.0          # marked, not touched
  assert float(s.grad.abs().max()) == 0.0 and not getattr(s, '_grad_fresh', False)
  before = w.detach().clone()
  opt.apply()                                  # no kernel wrote w.grad: it counts as zero, w must not move
  torch.cuda.synchronize()
  assert float(w.grad.abs().max()) == 0.0 and torch.equal(w.detach(), before)


def test_pack_group_repack_equals_single_layer_pack(hip):
  """csmri_pack_weight_multi (one launch for every layer of a network after an optimizer step; vectorised path for
  full 64-channel 4x4 tiles, generic path otherwise) against csmri_pack_weight layer by layer: bit-identical packed
  buffers for the forward, data-gradient and sub-pixel modes, bf16 and fp32, aligned and unaligned weight storage."""
  ops = hip.ops
  gen = torch.Generator().manual_seed(21)
  shapes = [(128, 64, 4, 4), (64, 128, 4, 4), (96, 72, 4, 4), (40, 24, 3, 3), (1, 512, 4, 4), (256, 256, 4, 4)]
  flat = torch.randn(sum(a * b * c * d for a, b, c, d in shapes) + 8, generator=gen).cuda()
  for dtype in (torch.bfloat16, torch.float32):
    for shift in (0, 1):             # shift 1: weights start 4 bytes off a 16-byte boundary (flat optimizer buffers)
      layers, off = [], shift
      for sh in shapes:
        n = sh[0] * sh[1] * sh[2] * sh[3]
        layers.append(ops.ConvLayer(flat[off:off + n].view(sh), None, 1, (1, 2, 1, 2), 'zero', dtype))
        off += n
      group = ops.PackGroup(layers)
      modes = (0, 3, 2)
      for l in layers:
        for m in modes:
          if m == 2 and l.kh % 2:
            continue
          l._pack(m)
      first = {(i, m): l._packs[m][1].clone() for i, l in enumerate(layers) for m in l._packs}
      flat.mul_(-0.5).add_(0.25)       # "optimizer step"
      group.bump()
      for m in modes:
        assert group.repack(m)
      for i, l in enumerate(layers):
        ref = ops.ConvLayer(l.weight, None, 1, (1, 2, 1, 2), 'zero', dtype)
        for m in l._packs:
          got = l._packs[m][1]
          want = ref._pack(m)[0]
          assert torch.equal(got, want), (dtype, shift, i, m)
          assert not torch.equal(got, first[(i, m)]), (dtype, shift, i, m)
      flat.sub_(0.25).div_(-0.5)


def test_image_pool_exchange_matches_sequential_reference(hip):
  """csmri_image_pool_exchange (one launch) against the oracle's sequential pool (reference utils/image_pool.py:29-60)
  on the same decisions: filling phase, draws, a slot drawn twice in one batch (the second draw must see the image
  the first one stored), plus the device pool content afterwards.  Bit-exact (copies)."""
  import random
  import csmri_oracle as O
  from utils.image_pool import ImagePool
  rng = random.Random(7)
  a, b = ImagePool(6), O.ImagePool(6)
  for step in range(12):
    x = torch.randn(4, 8, 8, 8, generator=torch.Generator().manual_seed(step)).bfloat16()
    dec = a.decide(4)
    if step >= 2:                      # force pool draws, some hitting the same slot within the batch
      dec = [(rng.random() < 0.7, rng.randrange(3)) for _ in range(4)]
    ra = a.query(x.cuda(), dec)
    rb = b.query(x, list(dec))
    assert torch.equal(ra.cpu(), rb), step
  for slot in range(6):
    assert torch.equal(a.buffer[slot].cpu(), b.images[slot].reshape(a.buffer[slot].shape)), slot


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
def test_maxpool_skip_and_act_fused_backward(hip, dtype):
  """csmri_maxpool2_bwd_act: un-pooling + skip-connection gradient (U-Net encoder, MaxPool2Skip) and un-pooling x
  activation derivative (VGG backward) against autograd on max_pool2d / leaky_relu; exact in fp32, one bf16
  rounding of the sum in bf16."""
  ops = hip.ops
  gen = torch.Generator().manual_seed(12)
  x = torch.randn(2, 16, 12, 20, generator=gen)
  g = torch.randn(2, 16, 6, 10, generator=gen)
  gs = torch.randn(2, 16, 12, 20, generator=gen)
  if dtype == torch.bfloat16:
    x, g, gs = x.bfloat16().float(), g.bfloat16().float(), gs.bfloat16().float()
  # pool + skip
  xd = to_dev_nhwc(x, dtype).requires_grad_(True)
  y, skip = ops.MaxPool2Skip.apply(xd)
  xr = x.clone().requires_grad_(True)
  yr = F.max_pool2d(xr, 2, 2)
  assert torch.equal(from_dev_nhwc(y, 16), yr.detach()) and torch.equal(from_dev_nhwc(skip, 16), x)
  torch.autograd.backward([yr, xr], [g, gs])
  torch.autograd.backward([y, skip], [to_dev_nhwc(g, dtype), to_dev_nhwc(gs, dtype)])
  got, ref = from_dev_nhwc(xd.grad, 16), xr.grad
  if dtype == torch.float32:
    assert torch.equal(got, ref)
  else:
    assert torch.equal(got, ref.bfloat16().float())
  # only one of the two outputs used
  for which in (0, 1):
    xd = to_dev_nhwc(x, dtype).requires_grad_(True)
    outs = ops.MaxPool2Skip.apply(xd)
    outs[which].backward(to_dev_nhwc((g, gs)[which], dtype))
    xr = x.clone().requires_grad_(True)
    (F.max_pool2d(xr, 2, 2) if which == 0 else xr * 1).backward((g, gs)[which])
    assert torch.equal(from_dev_nhwc(xd.grad, 16), xr.grad)
  # pool gradient times the derivative of the activation that fed the pool
  for slope in (0.0, 0.2):
    z = F.leaky_relu(x, slope) if slope else F.relu(x)
    if dtype == torch.bfloat16:
      z = z.bfloat16().float()
    zd = to_dev_nhwc(z, dtype)
    pooled, arg = ops.maxpool2_fwd(zd)
    got = ops.maxpool2_bwd(to_dev_nhwc(g, dtype), arg, tuple(zd.shape), g_src=zd, g_slope=slope)
    two = ops.act_bwd(ops.maxpool2_bwd(to_dev_nhwc(g, dtype), arg, tuple(zd.shape)), zd, slope)
    assert torch.equal(got, two)
    # the same with the POOLED tensor as the gate (csmri_maxpool2_bwd_pooled_gate: what the frozen VGG stack's backward uses)
    assert torch.equal(ops.maxpool2_bwd(to_dev_nhwc(g, dtype), arg, tuple(zd.shape), g_pooled=pooled, g_slope=slope), got)
    zr = z.clone().requires_grad_(True)
    F.max_pool2d(zr, 2, 2).backward(g)
    ref = zr.grad * torch.where(z > 0, torch.ones_like(z), torch.full_like(z, slope))
    ref = ref if dtype == torch.float32 else ref.bfloat16().float()
    assert torch.equal(from_dev_nhwc(got, 16), ref)


@pytest.mark.parametrize('shape', [(2, 64, 64), (1, 256, 256), (2, 128, 64), (1, 512, 512), (3, 32, 32)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_dc(hip, shape):
  ops = hip.ops
  b, h, w = shape
  g = torch.Generator().manual_seed(h * 7 + w)
  x = torch.randn(b, 2, h, w, generator=g)
  m2 = (torch.rand(b, 1, h, w, generator=g) < 0.3).float().expand(b, 2, h, w).contiguous()
  k0 = torch.randn(b, 2, h, w, generator=g) * m2
  gy = torch.randn(b, 2, h, w, generator=g)
  ref = O.dc_layer(x.double(), k0.double(), m2.double()).float()
  refg = O.dc_adjoint(gy.double(), m2.double()).float()
  xd = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
  k0d = k0.permute(0, 2, 3, 1).contiguous().cuda()
  mu8 = ops.mask_to_u8(m2.cuda())
  assert torch.equal(mu8.cpu(), m2[:, 0].to(torch.uint8))      # integer mask: bit exact
  out, pad = ops.DataConsistency.apply(xd, k0d, mu8, torch.bfloat16)
  got = out.detach().cpu().permute(0, 3, 1, 2)
  err = rel_l2(got, ref)
  print('dc %s fwd rel_l2 %.3e max_abs %.3e' % (shape, err, float((got - ref).abs().max())))
  assert err < 5e-6
  assert torch.allclose(got, ref, rtol=1e-4, atol=2e-5)
  assert torch.equal(pad[..., :2].float().cpu(), out.detach().bfloat16().float().cpu())
  assert float(pad[..., 2:].float().abs().max()) == 0.0
  out.backward(gy.permute(0, 2, 3, 1).contiguous().cuda())
  gg = xd.grad.cpu().permute(0, 3, 1, 2)
  assert rel_l2(gg, refg) < 5e-6
  assert torch.allclose(gg, refg, rtol=1e-4, atol=2e-5)
  # linearity + fixed point: DC(x) with k0 = mask*FFT(x) returns x
  kx = torch.fft.fft2(torch.complex(x[:, 0], x[:, 1]), norm='ortho')
  kx = torch.stack((kx.real, kx.imag), 1) * m2
  out2, _ = ops.dc_raw(xd.detach(), kx.permute(0, 2, 3, 1).contiguous().cuda(), mu8)
  assert torch.allclose(out2.cpu(), xd.detach().cpu(), atol=2e-5)


@pytest.mark.parametrize('shape', [(2, 64, 64), (1, 256, 256), (2, 128, 32), (1, 512, 512)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_dc_and_fft2_bf16_storage(hip, shape):
  """The "bf16 cFFT" of BASELINE config 5 (csmri_dc_bf16 / csmri_fft2_bf16): images and the intermediates between
  the three passes are stored as bf16, arithmetic is fp32.  Checked against (a) a numpy restatement that rounds at
  exactly those storage points (float64 arithmetic in between): relative L2 <= 1e-3 (isolated one-ulp flips where
  the fp32 value sits on a bf16 rounding boundary); (b) the exact transform (myfft.py:131-163 in float64):
  relative L2 <= 5e-3, max error <= 1e-2 of the output maximum -- the cost of the storage format."""
  ops = hip.ops
  b, h, w = shape
  rng = np.random.RandomState(h * 3 + w)

  def bf(a):      # round a complex float64 array to bf16 storage
    t = torch.from_numpy(np.stack((a.real, a.imag), -1)).float().bfloat16().double().numpy()
    return t[..., 0] + 1j * t[..., 1]
  x = bf(rng.randn(b, h, w) + 1j * rng.randn(b, h, w))
  m = rng.rand(b, h, w) < 0.3
  k0 = ((rng.randn(b, h, w) + 1j * rng.randn(b, h, w)) * m).astype(np.complex64).astype(np.complex128)
  sc = 1.0 / np.sqrt(h * w)
  exact = np.fft.ifft2(np.where(m, 0, np.fft.fft2(x, norm='ortho')) + k0, norm='ortho')
  t = bf(np.fft.fft(x, axis=2))                                   # pass 1: rows, unscaled
  t = np.where(m, 0, np.fft.fft(t, axis=1) * sc) + k0              # pass 2: columns, scale, merge ...
  t = bf(np.fft.ifft(t, axis=1) * h)                               # ... inverse columns (unscaled)
  emu = bf(np.fft.ifft(t, axis=2) * w * sc)                        # pass 3: inverse rows, scale
  xd = torch.from_numpy(np.stack((x.real, x.imag), -1)).bfloat16().cuda()
  k0d = torch.from_numpy(np.stack((k0.real, k0.imag), -1)).float().cuda()
  mu8 = torch.from_numpy(m.astype(np.uint8)).cuda()
  out, pad = ops.dc_raw(xd, k0d, mu8, torch.bfloat16)
  assert out.dtype == torch.bfloat16
  got = out.float().cpu().numpy().astype(np.float64)
  got = got[..., 0] + 1j * got[..., 1]
  e_emu = np.linalg.norm(got - emu) / np.linalg.norm(emu)
  e_ex = np.linalg.norm(got - exact) / np.linalg.norm(exact)
  e_max = np.abs(got - exact).max() / np.abs(exact).max()
  print('dc bf16 %s vs rounding restatement %.2e | vs exact rel_l2 %.2e max %.2e' % (shape, e_emu, e_ex, e_max))
  assert e_emu < 1e-3 and e_ex < 5e-3 and e_max < 1e-2
  assert torch.equal(pad[..., :2], out) and float(pad[..., 2:].float().abs().max()) == 0.0
  # channel-padded bf16 conv output as the input (pixel stride 8)
  x8 = torch.zeros(b, h, w, 8, dtype=torch.bfloat16, device='cuda')
  x8[..., :2] = xd
  x8[..., 2:] = 7.0
  out8, _ = ops.dc_raw(x8, k0d, mu8, None)
  assert torch.equal(out8, out)
  # stand-alone transform, both directions
  for inverse in (False, True):
    ref = (np.fft.ifft2 if inverse else np.fft.fft2)(x, norm='ortho')
    g2 = ops.fft2(xd, inverse, True).float().cpu().numpy().astype(np.float64)
    err = np.linalg.norm((g2[..., 0] + 1j * g2[..., 1]) - ref) / np.linalg.norm(ref)
    assert err < 4e-3, (inverse, err)


@pytest.mark.parametrize('shape', [(2, 64, 64), (1, 256, 256), (3, 40, 56), (2, 17, 33), (1, 512, 512)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_convblock_fused_equals_per_layer_path(hip, shape):
  """csmri_convblock_fused_fwd (one launch for RecNet's [pad, conv, LeakyReLU] x 2 + [pad, conv] block, reference
  models/recnet.py:29-62) against the per-layer csmri_gconv path it replaces: output, both saved activations and --
  through the unchanged backward kernels -- every gradient must be BIT-IDENTICAL (same K order per output
  element; halo pixels are recomputed with the same arithmetic as their owner tile); and against the CPU oracle on
  bf16-rounded operands within the bf16 tolerance of the per-layer tests (relative L2 <= 1e-2)."""
  ops = hip.ops
  b, h, w = shape
  g = torch.Generator().manual_seed(h * 5 + w)
  ws = [torch.randn(32, 2, 3, 3, generator=g) * 0.4, torch.randn(32, 32, 3, 3, generator=g) * 0.08,
        torch.randn(2, 32, 3, 3, generator=g) * 0.08]
  bs = [torch.randn(32, generator=g) * 0.1, torch.randn(32, generator=g) * 0.1, torch.randn(2, generator=g) * 0.1]
  x = torch.randn(b, 2, h, w, generator=g)
  gy = torch.randn(b, 2, h, w, generator=g)

  ops.FUSED_CONVBLOCK_BWD = False          # this test pins the forward; the fused backward has its own below

  def run(fused, train):
    ops.FUSED_CONVBLOCK = fused
    try:
      params = [(torch.nn.Parameter(wt.clone().cuda(), requires_grad=train),
                 torch.nn.Parameter(bi.clone().cuda(), requires_grad=train)) for wt, bi in zip(ws, bs)]
      plan = [(ops.ConvLayer(wp, bp, 1, (1, 1, 1, 1), 'zero', torch.bfloat16), 0.01 if i < 2 else 1.0)
              for i, (wp, bp) in enumerate(params)]
      xd = to_dev_nhwc(x, torch.bfloat16).requires_grad_(train)
      log = ops.LAUNCH_LOG = []
      y = ops.ConvActStack.apply(xd, plan, torch.float32, *[t for pr in params for t in pr])
      names = [e[1] for e in log]
      ops.LAUNCH_LOG = None
      grads = None
      if train:
        y.backward(to_dev_nhwc(gy, torch.float32))
        ops.join_wgrad_stream()
        grads = [xd.grad.clone()] + [t.grad.clone() for pr in params for t in pr]
      torch.cuda.synchronize()
      return y.detach().clone(), grads, names
    finally:
      ops.FUSED_CONVBLOCK = True
      ops.LAUNCH_LOG = None
  for train in (False, True):
    yf, gf, nf = run(True, train)
    yu, gu, nu = run(False, train)
    assert nf == ['convblock_fwd_kernel<%s>' % ('true' if train else 'false')], nf
    assert len(nu) == 3 and not any('convblock' in n for n in nu), nu
    assert torch.equal(yf, yu), float((yf - yu).abs().max())
    assert float(yf[..., 2:].abs().max()) == 0.0
    if train:
      for a, c in zip(gf, gu):
        assert torch.equal(a, c)
  # oracle: the reference block on bf16-rounded operands (fp32 accumulate), intermediates rounded to bf16
  r = x.bfloat16().float()
  for i in range(3):
    r = F.conv2d(F.pad(r, (1, 1, 1, 1)), ws[i].bfloat16().float(), bs[i])
    if i < 2:
      r = F.leaky_relu(r, 0.01).bfloat16().float()
  err = rel_l2(from_dev_nhwc(yf, 2), r)
  print('convblock fused %s vs oracle rel_l2 %.3e' % (shape, err))
  assert err < 1e-2
  # dense complex output ([B,H,W,2] fp32, what RecNet hands to the DC layer): same values as channels 0,1 of the
  # padded output and the same gradients from a 2-channel output gradient, fused and per layer
  for fused in (True, False):
    ops.FUSED_CONVBLOCK = fused
    try:
      params = [(torch.nn.Parameter(wt.clone().cuda()), torch.nn.Parameter(bi.clone().cuda())) for wt, bi in zip(ws, bs)]
      plan = [(ops.ConvLayer(wp, bp, 1, (1, 1, 1, 1), 'zero', torch.bfloat16), 0.01 if i < 2 else 1.0)
              for i, (wp, bp) in enumerate(params)]
      xd = to_dev_nhwc(x, torch.bfloat16).requires_grad_(True)
      yc = ops.ConvActStack.apply(xd, plan, ('complex', torch.float32), *[t for pr in params for t in pr])
      assert tuple(yc.shape) == (b, h, w, 2) and yc.dtype == torch.float32 and yc.is_contiguous()
      assert torch.equal(yc.detach(), yf[..., :2])
      yc.backward(to_dev_nhwc(gy, torch.float32)[..., :2].contiguous())
      ops.join_wgrad_stream()
      got = [xd.grad] + [t.grad for pr in params for t in pr]
      for a, c in zip(got, gf):
        assert torch.equal(a, c)
    finally:
      ops.FUSED_CONVBLOCK = True
  ops.FUSED_CONVBLOCK_BWD = True


@pytest.mark.parametrize('need_dx', [True, False], ids=['dx', 'nodx'])
@pytest.mark.parametrize('complex_out', [False, True], ids=['padded', 'complex'])
@pytest.mark.parametrize('shape', [(2, 64, 64), (1, 256, 256), (3, 40, 56), (2, 17, 33), (8, 128, 128)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_convblock_fused_backward(hip, shape, complex_out, need_dx):
  """csmri_convblock_fused_bwd (the backward of RecNet's conv block, reference models/recnet.py:29-62 under
  loss.backward() of training/runner.py:163, as ONE launch + the slab reduction) against (a) the six-kernel per-layer
  backward it replaces and (b) torch autograd on the CPU with the same bf16-rounded operands and saved activations.
  The two device paths round the intermediate gradients dA2 / dA1 to bf16 from fp32 sums taken in different orders,
  so they agree to bf16 rounding (dX relative L2 <= 4e-3, weight / bias gradients <= 3e-3), and the fused path's
  distance from the oracle must not exceed the per-layer path's by more than 30 % (+ 1e-4)."""
  ops = hip.ops
  b, h, w = shape
  g = torch.Generator().manual_seed(h * 7 + w + b)
  ws = [torch.randn(32, 2, 3, 3, generator=g) * 0.4, torch.randn(32, 32, 3, 3, generator=g) * 0.08,
        torch.randn(2, 32, 3, 3, generator=g) * 0.08]
  bs = [torch.randn(32, generator=g) * 0.1, torch.randn(32, generator=g) * 0.1, torch.randn(2, generator=g) * 0.1]
  x = torch.randn(b, 2, h, w, generator=g)
  gy = torch.randn(b, 2, h, w, generator=g)

  def run(fused_bwd):
    ops.FUSED_CONVBLOCK_BWD = fused_bwd
    try:
      params = [(torch.nn.Parameter(wt.clone().cuda()), torch.nn.Parameter(bi.clone().cuda())) for wt, bi in zip(ws, bs)]
      plan = [(ops.ConvLayer(wp, bp, 1, (1, 1, 1, 1), 'zero', torch.bfloat16), 0.01 if i < 2 else 1.0)
              for i, (wp, bp) in enumerate(params)]
      xd = to_dev_nhwc(x, torch.bfloat16).requires_grad_(need_dx)
      y = ops.ConvActStack.apply(xd, plan, ('complex', torch.float32) if complex_out else torch.float32,
                                 *[t for pr in params for t in pr])
      gd = to_dev_nhwc(gy, torch.float32)
      log = ops.LAUNCH_LOG = []
      y.backward(gd[..., :2].contiguous() if complex_out else gd)
      ops.join_wgrad_stream()
      torch.cuda.synchronize()
      names = [e[1] for e in log]
      ops.LAUNCH_LOG = None
      return ([from_dev_nhwc(xd.grad, 2)] if need_dx else []) + [t.grad.cpu().clone() for pr in params for t in pr], names
    finally:
      ops.FUSED_CONVBLOCK_BWD = True
      ops.LAUNCH_LOG = None
  gf, nf = run(True)
  gu, nu = run(False)
  assert nf == ['convblock_bwd_kernel'], nf
  assert 'convblock_bwd_kernel' not in nu and len(nu) >= 5, nu
  # oracle: autograd through the reference block with the forward's bf16 roundings (operands, saved activations)
  xr = x.bfloat16().float().requires_grad_(True)
  wr = [wt.bfloat16().float().requires_grad_(True) for wt in ws]
  br = [bi.clone().requires_grad_(True) for bi in bs]
  r = xr
  for i in range(3):
    r = F.conv2d(F.pad(r, (1, 1, 1, 1)), wr[i], br[i])
    if i < 2:
      r = F.leaky_relu(r, 0.01)
      r = r + (r.detach().bfloat16().float() - r.detach())      # value rounded to bf16, gradient straight through
  r.backward(gy.bfloat16().float())
  want = ([xr.grad] if need_dx else []) + [t.grad for pr in zip(wr, br) for t in pr]
  labels = (['dx'] if need_dx else []) + ['dw1', 'db1', 'dw2', 'db2', 'dw3', 'db3']
  for lab, a, c, o in zip(labels, gf, gu, want):
    e_fu, e_fo, e_uo = rel_l2(a, c), rel_l2(a, o), rel_l2(c, o)
    print('convblock bwd %s %-4s fused vs per-layer %.2e | vs oracle: fused %.2e per-layer %.2e' % (shape, lab, e_fu, e_fo, e_uo))
    assert e_fu < (4e-3 if lab == 'dx' else 3e-3), (lab, e_fu)
    assert e_fo <= 1.3 * e_uo + 1e-4, (lab, e_fo, e_uo)
  if need_dx:
    xg = None
  # pad channels of dX stay exact zeros
  if need_dx:
    ops.FUSED_CONVBLOCK_BWD = True
    params = [(torch.nn.Parameter(wt.clone().cuda()), torch.nn.Parameter(bi.clone().cuda())) for wt, bi in zip(ws, bs)]
    plan = [(ops.ConvLayer(wp, bp, 1, (1, 1, 1, 1), 'zero', torch.bfloat16), 0.01 if i < 2 else 1.0)
            for i, (wp, bp) in enumerate(params)]
    xd = to_dev_nhwc(x, torch.bfloat16).requires_grad_(True)
    y = ops.ConvActStack.apply(xd, plan, torch.float32, *[t for pr in params for t in pr])
    y.backward(to_dev_nhwc(gy, torch.float32))
    ops.join_wgrad_stream()
    assert float(xd.grad[..., 2:].float().abs().max()) == 0.0


def test_convblock_fused_forward_backward_at_bench_shape(hip):
  """The C2 benchmark's own shape -- 64 slices of 256 x 256 (bench.py --config c2), where
  csmri_convblock_fused_bwd_splits(b, h, w) and the persistent-worker tiling differ from the small cases above --
  with ABSOLUTE bounds against torch autograd on the CPU (reference models/recnet.py:29-62 under loss.backward() of
  training/runner.py:163) on the same bf16-rounded operands and saved activations:
    forward  relative L2 <= 1e-2 (bf16 intermediates), dense complex fp32 output;
    backward dX relative L2 <= 4e-3 (dA2, dA1 and dX are each rounded to bf16 once), dW <= 3e-3 (fp32 sums of
    ~4 M products of bf16-rounded factors), db <= 6e-3 (plain sums of the rounded gradient images: cancellation)."""
  ops = hip.ops
  b, h, w = 64, 256, 256
  g = torch.Generator().manual_seed(64256)
  ws = [torch.randn(32, 2, 3, 3, generator=g) * 0.4, torch.randn(32, 32, 3, 3, generator=g) * 0.08,
        torch.randn(2, 32, 3, 3, generator=g) * 0.08]
  bs = [torch.randn(32, generator=g) * 0.1, torch.randn(32, generator=g) * 0.1, torch.randn(2, generator=g) * 0.1]
  x = torch.randn(b, 2, h, w, generator=g)
  gy = torch.randn(b, 2, h, w, generator=g)
  params = [(torch.nn.Parameter(wt.clone().cuda()), torch.nn.Parameter(bi.clone().cuda())) for wt, bi in zip(ws, bs)]
  plan = [(ops.ConvLayer(wp, bp, 1, (1, 1, 1, 1), 'zero', torch.bfloat16), 0.01 if i < 2 else 1.0)
          for i, (wp, bp) in enumerate(params)]
  xd = to_dev_nhwc(x, torch.bfloat16).requires_grad_(True)
  log = ops.LAUNCH_LOG = []
  try:
    y = ops.ConvActStack.apply(xd, plan, ('complex', torch.float32), *[t for pr in params for t in pr])
    y.backward(to_dev_nhwc(gy, torch.float32)[..., :2].contiguous())
    ops.join_wgrad_stream()
    torch.cuda.synchronize()
  finally:
    ops.LAUNCH_LOG = None
  names = [e[1] for e in log]
  assert names == ['convblock_fwd_kernel<true>', 'convblock_bwd_kernel'], names
  assert log[1][2] == hip.lib.raw('csmri_convblock_fused_bwd_splits')(b, h, w)
  got_y = y.detach().permute(0, 3, 1, 2).float().cpu()
  got = [from_dev_nhwc(xd.grad, 2)] + [t.grad.cpu().clone() for pr in params for t in pr]
  assert float(xd.grad[..., 2:].float().abs().max()) == 0.0
  del y, xd
  torch.cuda.empty_cache()
  # oracle
  xr = x.bfloat16().float().requires_grad_(True)
  wr = [wt.bfloat16().float().requires_grad_(True) for wt in ws]
  br = [bi.clone().requires_grad_(True) for bi in bs]
  r = xr
  for i in range(3):
    r = F.conv2d(F.pad(r, (1, 1, 1, 1)), wr[i], br[i])
    if i < 2:
      r = F.leaky_relu(r, 0.01)
      r = r + (r.detach().bfloat16().float() - r.detach())      # value rounded to bf16, gradient straight through
  e_y = rel_l2(got_y, r.detach())
  print('convblock bench shape fwd vs oracle rel_l2 %.3e' % e_y)
  assert e_y < 1e-2
  r.backward(gy.bfloat16().float())
  want = [xr.grad] + [t.grad for pr in zip(wr, br) for t in pr]
  errs = {}
  for lab, a, o in zip(['dx', 'dw1', 'db1', 'dw2', 'db2', 'dw3', 'db3'], got, want):
    errs[lab] = rel_l2(a, o)
    print('convblock bench shape bwd %-4s vs CPU autograd rel_l2 %.3e' % (lab, errs[lab]))
  for lab, e in errs.items():
    # bias gradients are plain sums of the bf16-rounded gradient images over 4.2 M pixels (the oracle sums the
    # unrounded fp32 ones): cancellation leaves them the loosest of the seven (measured 3.4e-3 on db1)
    assert e < {'dx': 4e-3, 'db1': 6e-3, 'db2': 6e-3, 'db3': 6e-3}.get(lab, 3e-3), (lab, e)


def test_split_bf16_image_format(hip):
  """CSMRI_BF16_SPLIT (include/csmri_hip.h): a 2-channel image in a channel-padded bf16 pixel as hi = bf16(v) in channels
  0,1 and lo = bf16(v - hi) in channels 2,3.  Producers: csmri_nchw_to_nhwc and the padded copy of csmri_dc; consumer
  of a split GRADIENT: csmri_dc_in_bf16 (the DC adjoint), which reads channels (0,1) + (2,3).  hi + lo reproduces the
  fp32 value to 2^-16 relative (bf16 alone: 2^-9); a plain padded bf16 tensor (zeros in channels 2,3) reads unchanged."""
  ops = hip.ops
  g = torch.Generator().manual_seed(77)
  b, h, w = 2, 64, 64
  x = torch.randn(b, 2, h, w, generator=g)
  t = ops.nchw_to_nhwc(x.cuda(), torch.bfloat16, 8, split=True)
  assert t.shape == (b, h, w, 8) and t.dtype == torch.bfloat16
  tf = t.float().cpu()
  want = x.permute(0, 2, 3, 1)
  assert torch.equal(tf[..., :2], want.bfloat16().float())                   # hi = the plain bf16 rounding
  assert torch.equal(tf[..., 2:4], (want - tf[..., :2]).bfloat16().float())  # lo = bf16 of the remainder
  assert float(tf[..., 4:].abs().max()) == 0.0
  rec = tf[..., :2] + tf[..., 2:4]
  e_split, e_plain = float((rec - want).abs().max() / want.abs().max()), float((tf[..., :2] - want).abs().max() / want.abs().max())
  print('split bf16: max rel err hi+lo %.2e, hi alone %.2e' % (e_split, e_plain))
  assert e_split < 2.0 ** -15 and e_plain > 2.0 ** -10
  # DC: the padded copy in split form against the fp32 result of the same call
  k0 = torch.randn(b, h, w, 2, generator=g).cuda()
  mask = (torch.rand(b, h, w, generator=g) < 0.3).to(torch.uint8).cuda()
  xc = ops.nchw_to_nhwc(x.cuda(), torch.float32, 2)
  out, pad = ops.dc_raw(xc, k0, mask, (torch.bfloat16, 'split'))
  out2, pad2 = ops.dc_raw(xc, k0, mask, torch.bfloat16)
  assert torch.equal(out, out2)
  pf, of = pad.float(), out
  assert torch.equal(pf[..., :2], pad2.float()[..., :2]) and float(pad2.float()[..., 2:].abs().max()) == 0.0
  assert torch.equal(pf[..., 2:4], (of - pf[..., :2]).bfloat16().float()) and float(pf[..., 4:].abs().max()) == 0.0
  assert float((pf[..., :2] + pf[..., 2:4] - of).abs().max() / of.abs().max()) < 2.0 ** -15
  # the adjoint on a split gradient == the fp32 adjoint of (hi + lo); on a plain padded gradient == of its channels 0,1
  gsp = pad                                                   # any split tensor serves as a gradient
  a_split, _ = ops.dc_raw(gsp, None, mask, None, out_fp32=True, x_split=True)
  a_ref, _ = ops.dc_raw((pf[..., :2] + pf[..., 2:4]).contiguous(), None, mask, None)
  a_plain, _ = ops.dc_raw(pad2, None, mask, None, out_fp32=True)
  a_ref_plain, _ = ops.dc_raw(pad2.float()[..., :2].contiguous(), None, mask, None)
  assert rel_l2(a_split.cpu(), a_ref.cpu()) < 1e-6 and rel_l2(a_plain.cpu(), a_ref_plain.cpu()) < 1e-6
  # the format is DECLARED (ABI 101), not inferred from the stride: a PLAIN padded gradient whose channels 2..7 are not
  # zero (a user-supplied gradient, one summed with another consumer of the padded tensor) is read on channels 0,1 alone
  noisy = torch.randn(b, h, w, 8, generator=g).bfloat16().cuda()
  a_noisy, _ = ops.dc_raw(noisy, None, mask, None, out_fp32=True)
  a_ref_noisy, _ = ops.dc_raw(noisy.float()[..., :2].contiguous(), None, mask, None)
  assert rel_l2(a_noisy.cpu(), a_ref_noisy.cpu()) < 1e-6


def test_convblock_fused_split_images(hip):
  """RecNet's fused conv block (reference models/recnet.py:29-62) fed a CSMRI_BF16_SPLIT input, 2 x 128 x 128: against
  torch autograd on the CPU with the UNROUNDED fp32 input (weights and saved activations bf16-rounded as on the device),
    forward: closer to that oracle than the same block on the plain bf16 input;
    dW1 (x^T dA1): likewise -- the hi and the lo products are summed into one gradient;
    dX: returned split, hi + lo closer to the oracle than the plain bf16 dX (whose output rounding alone is 1.7e-3);
    the other gradients (they read a1, a2 and dY, which move with layer 1's input) stay within 1e-1 of the plain mode's."""
  ops = hip.ops
  b, h, w = 2, 128, 128
  g = torch.Generator().manual_seed(4128)
  ws = [torch.randn(32, 2, 3, 3, generator=g) * 0.4, torch.randn(32, 32, 3, 3, generator=g) * 0.08,
        torch.randn(2, 32, 3, 3, generator=g) * 0.08]
  bs = [torch.randn(32, generator=g) * 0.1, torch.randn(32, generator=g) * 0.1, torch.randn(2, generator=g) * 0.1]
  x = torch.randn(b, 2, h, w, generator=g)
  gy = torch.randn(b, 2, h, w, generator=g)

  def run(split):
    params = [(torch.nn.Parameter(wt.clone().cuda()), torch.nn.Parameter(bi.clone().cuda())) for wt, bi in zip(ws, bs)]
    plan = [(ops.ConvLayer(wp, bp, 1, (1, 1, 1, 1), 'zero', torch.bfloat16), 0.01 if i < 2 else 1.0)
            for i, (wp, bp) in enumerate(params)]
    xd = ops.nchw_to_nhwc(x.cuda(), torch.bfloat16, 8, split=split).requires_grad_(True)
    out = ('complex', torch.float32, 'split') if split else ('complex', torch.float32)
    log = ops.LAUNCH_LOG = []
    try:
      y = ops.ConvActStack.apply(xd, plan, out, *[t for pr in params for t in pr])
      y.backward(to_dev_nhwc(gy, torch.float32)[..., :2].contiguous())
      ops.join_wgrad_stream()
      torch.cuda.synchronize()
    finally:
      ops.LAUNCH_LOG = None
    assert [e[1] for e in log] == ['convblock_fwd_kernel<true>', 'convblock_bwd_kernel'], log
    dx = xd.grad.float().cpu()
    dx = (dx[..., :2] + dx[..., 2:4]) if split else dx[..., :2]
    if not split:
      assert float(xd.grad[..., 2:].float().abs().max()) == 0.0
    return (y.detach().permute(0, 3, 1, 2).float().cpu(), dx.permute(0, 3, 1, 2).contiguous(),
            [t.grad.cpu().clone() for pr in params for t in pr])
  ys, dxs, gs = run(True)
  yp, dxp, gp = run(False)
  xr = x.clone().requires_grad_(True)                       # the UNROUNDED input
  wr = [wt.bfloat16().float().requires_grad_(True) for wt in ws]
  br = [bi.clone().requires_grad_(True) for bi in bs]
  r = xr
  for i in range(3):
    r = F.conv2d(F.pad(r, (1, 1, 1, 1)), wr[i], br[i])
    if i < 2:
      r = F.leaky_relu(r, 0.01)
      r = r + (r.detach().bfloat16().float() - r.detach())
  r.backward(gy.bfloat16().float())
  e = {'fwd': (rel_l2(ys, r.detach()), rel_l2(yp, r.detach())), 'dx': (rel_l2(dxs, xr.grad), rel_l2(dxp, xr.grad)),
       'dw1': (rel_l2(gs[0], wr[0].grad), rel_l2(gp[0], wr[0].grad))}
  for k, (a, c) in e.items():
    print('convblock split images %-4s vs fp32-input oracle: split %.3e  plain bf16 %.3e' % (k, a, c))
  # measured: forward 3.7e-4 (plain 4.5e-3); dX 4.2e-3 and dW1 4.2e-3 (plain 5.2e-2: rounding the input to 8 bits flips
  # LeakyReLU signs of layer 1, which the gradients see at full size)
  assert e['fwd'][0] < 0.3 * e['fwd'][1] and e['fwd'][0] < 1.5e-3
  assert e['dx'][0] < 0.3 * e['dx'][1] and e['dx'][0] < 1e-2
  assert e['dw1'][0] < 0.3 * e['dw1'][1] and e['dw1'][0] < 1e-2
  # the other gradients read a1, a2, dY only: the forward differs through x, so they are close, not identical
  for i, lab in ((1, 'db1'), (2, 'dw2'), (3, 'db2'), (4, 'dw3'), (5, 'db3')):
    assert rel_l2(gs[i], gp[i]) < 1e-1, (lab, rel_l2(gs[i], gp[i]))


def test_nchw_gradient_to_nhwc_with_addend(hip):
  """csmri_nchw_to_nhwc_add: pad(NCHW fp32) + NHWC addend, rounded once to the output dtype -- the gradient of a tensor
  handed out as an NCHW API tensor AND as a device-layout feature map (the discriminator's logits, reference
  models/discriminators.py:236-247).  Bit-exact against torch on the same operands."""
  ops = hip.ops
  lib = ops.lib
  g = torch.Generator().manual_seed(3)
  for b, c, h, w, cp in ((8, 1, 5, 5, 8), (3, 2, 7, 9, 8), (2, 5, 4, 4, 16)):
    src = torch.randn(b, c, h, w, generator=g).cuda()
    for odt in (torch.float32, torch.bfloat16):
      for adt in (None, torch.float32, torch.bfloat16):
        add = None if adt is None else torch.randn(b, h, w, cp, generator=g).to(adt).cuda()
        out = torch.full((b, h, w, cp), 7.0, dtype=odt).cuda()
        lib.call('csmri_nchw_to_nhwc_add', src.data_ptr(), b, c, h, w, out.data_ptr(), ops.dt_of(out), cp, cp,
                 ops.ptr(add), ops.dt_of(add) if add is not None else 0, cp if add is not None else 0, ops.stream())
        ref = torch.zeros(b, h, w, cp, device='cuda')
        ref[..., :c] = src.permute(0, 2, 3, 1)
        if add is not None:
          ref = ref + add.float()
        assert torch.equal(out, ref.to(odt)), (b, c, odt, adt)
  assert lib.raw('csmri_nchw_to_nhwc_add')(src.data_ptr(), 1, 5, 2, 2, out.data_ptr(), 0, 4, 4, 0, 0, 0, 0) < 0   # Cpad < C


def test_layout_roundtrip(hip):
  ops = hip.ops
  x = torch.randn(2, 3, 8, 12)
  for dt in (torch.float32, torch.bfloat16):
    t = ops.nchw_to_nhwc(x.cuda(), dt)
    assert t.shape == (2, 8, 12, 8)
    ref = x.to(dt).float()
    assert torch.equal(from_dev_nhwc(t, 3), ref)
    assert float(t[..., 3:].float().abs().max()) == 0
    back = ops.nhwc_to_nchw(t, 3).cpu()
    assert torch.equal(back, ref)
  # the 1- and 2-channel fast paths (four pixels per thread) incl. the dense interleaved complex form, and a shape
  # whose H*W is not a multiple of 4 (generic kernel): bit-exact
  for c, h, w in ((2, 16, 24), (1, 16, 24), (2, 5, 7), (2, 256, 256)):
    x = torch.randn(3, c, h, w)
    for dt in (torch.float32, torch.bfloat16):
      t = ops.nchw_to_nhwc(x.cuda(), dt)
      assert t.shape == (3, h, w, 8) and torch.equal(from_dev_nhwc(t, c), x.to(dt).float())
      assert float(t[..., c:].float().abs().max()) == 0
    if c == 2:
      t2 = ops.nchw_to_nhwc(x.cuda(), torch.float32, 2)
      assert t2.shape == (3, h, w, 2) and torch.equal(t2.cpu(), x.permute(0, 2, 3, 1).contiguous())


def test_small_ops(hip):
  ops = hip.ops
  g = torch.Generator().manual_seed(9)
  b, h, w = 2, 16, 24
  pre = torch.randn(b, 2, h, w, generator=g)
  u = torch.randn(b, 1, h, w, generator=g)
  scale = torch.tensor([0.37])
  # refinement combine vs oracle scale/unscale
  pre_d = pre.permute(0, 2, 3, 1).contiguous().cuda()
  ud = to_dev_nhwc(u, torch.float32).requires_grad_(True)
  sd = scale.clone().cuda().requires_grad_(True)
  pred, scaled, pred_b, u_b = ops.RefineCombine.apply(pre_d, ud, sd)
  assert pred_b.data_ptr() == pred.data_ptr() and u_b.data_ptr() == ud.data_ptr()
  ur, sr = u.clone().requires_grad_(True), scale.clone().requires_grad_(True)
  rs, mn, mx = O.scale_minmax(pre[:, 0:1].contiguous())
  outr = O.unscale_minmax(rs + sr * ur, mn, mx)
  predr = torch.cat((outr, pre[:, 1:2]), 1)
  assert torch.allclose(pred.detach().cpu().permute(0, 3, 1, 2), predr.detach(), atol=1e-6, rtol=1e-5)
  gp = torch.randn(predr.shape, generator=g)
  predr.backward(gp)
  pred.backward(gp.permute(0, 2, 3, 1).contiguous().cuda())
  assert torch.allclose(from_dev_nhwc(ud.grad, 1), ur.grad, atol=1e-6, rtol=1e-5)
  assert torch.allclose(sd.grad.cpu(), sr.grad, atol=1e-4, rtol=1e-4)
  # ... with every fan-in the node sums in its own kernel: two consumers of pred, a second consumer of u (the feature
  # penalty), the scale parameter's gradient written by the launch (on top of what .grad holds)
  sp = torch.nn.Parameter(scale.clone().cuda())
  sp.grad = torch.full_like(sp, 0.25)
  ud2 = to_dev_nhwc(u, torch.float32).requires_grad_(True)
  pred, scaled, pred_b, u_b = ops.RefineCombine.apply(pre_d, ud2, sp)
  gp2, gu2 = torch.randn(predr.shape, generator=g), torch.randn(u.shape, generator=g)
  ur, sr = u.clone().requires_grad_(True), scale.clone().requires_grad_(True)
  outr = O.unscale_minmax(rs + sr * ur, mn, mx)
  predr = torch.cat((outr, pre[:, 1:2]), 1)
  ((predr * (gp + gp2)).sum() + (ur * gu2).sum()).backward()
  ((pred * gp.permute(0, 2, 3, 1).cuda()).sum() + (pred_b * gp2.permute(0, 2, 3, 1).cuda()).sum()
   + (u_b * to_dev_nhwc(gu2, torch.float32)).sum()).backward()
  assert torch.allclose(from_dev_nhwc(ud2.grad, 1), ur.grad, atol=1e-5, rtol=1e-5)
  assert float(ud2.grad[..., 1:].abs().max()) == 0
  assert torch.allclose(sp.grad.cpu() - 0.25, sr.grad, atol=1e-4, rtol=1e-4)
  sp._grad_fresh = True            # FlatAdam's lazy zero: the first write overwrites
  pred, scaled, pred_b, u_b = ops.RefineCombine.apply(pre_d, ud2, sp)
  (pred_b * (gp + gp2).permute(0, 2, 3, 1).cuda()).sum().backward()
  assert torch.allclose(sp.grad.cpu(), sr.grad, atol=1e-4, rtol=1e-4) and not sp._grad_fresh
  # only the alias of u consumed
  ud3 = to_dev_nhwc(u, torch.float32).requires_grad_(True)
  u_b = ops.RefineCombine.apply(pre_d, ud3, sp)[3]
  (u_b * 2.0).sum().backward()
  assert float((ud3.grad - 2.0).abs().max()) == 0
  # complex abs (+ VGG normalisation) fwd/bwd
  xc = torch.randn(b, 2, h, w, generator=g)
  xd = xc.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
  a3 = ops.ComplexAbs.apply(xd, torch.float32, 3)
  xr = xc.clone().requires_grad_(True)
  ar = O.complex_abs(xr)
  mean = torch.tensor(O.VGG_MEAN).view(1, 3, 1, 1)
  std = torch.tensor(O.VGG_STD).view(1, 3, 1, 1)
  a3r = (torch.cat((ar, ar, ar), 1) - mean) / std
  assert torch.allclose(from_dev_nhwc(a3, 3), a3r.detach(), atol=1e-5, rtol=1e-5)
  g3 = torch.randn(a3r.shape, generator=g)
  a3r.backward(g3)
  a3.backward(to_dev_nhwc(g3, torch.float32))
  assert torch.allclose(xd.grad.cpu().permute(0, 3, 1, 2), xr.grad, atol=1e-5, rtol=1e-4)
  # losses
  for kind, fn in ((0, F.l1_loss), (1, F.mse_loss)):
    a = torch.randn(2, 5, 6, 7, generator=g)
    bt = torch.randn(2, 5, 6, 7, generator=g)
    ad = to_dev_nhwc(a, torch.float32).requires_grad_(True)
    bd = to_dev_nhwc(bt, torch.float32)
    l = ops.MeanLoss.apply(ad, bd, kind, 5)
    ar_ = a.clone().requires_grad_(True)
    lr = fn(ar_, bt)
    assert abs(l.item() - lr.item()) < 1e-6
    (l * 3.0).backward()
    (lr * 3.0).backward()
    assert torch.allclose(from_dev_nhwc(ad.grad, 5), ar_.grad, atol=1e-7, rtol=1e-5)
  # BCE on logits
  lg = torch.randn(2, 1, 5, 5, generator=g) * 3
  for t in (0.0, 0.9, 1.0):
    ld = lg.clone().cuda().requires_grad_(True)
    l = ops.BCELogits.apply(ld, t)
    lr_ = lg.clone().requires_grad_(True)
    p = torch.sigmoid(lr_)
    ref = F.binary_cross_entropy(p, torch.full_like(p, t))
    assert abs(l.item() - ref.item()) < 1e-6
    l.backward()
    ref.backward()
    assert torch.allclose(ld.grad.cpu(), lr_.grad, atol=1e-7, rtol=1e-4)
  # PSNR
  pr = torch.rand(3, 2, 16, 16, generator=g) * 1.2
  tg = torch.rand(3, 2, 16, 16, generator=g)
  mse = ops.psnr_mse(pr.permute(0, 2, 3, 1).contiguous().cuda(), tg.permute(0, 2, 3, 1).contiguous().cuda())
  val = float(np.mean(10 * np.log10(1.0 / mse.cpu().double().numpy())))
  assert abs(val - O.psnr_batch(pr, tg)) < 1e-4
  # Adam vs torch.optim.Adam, 3 steps
  p0 = torch.randn(1001, generator=g)
  pt = p0.clone().requires_grad_(True)
  opt = torch.optim.Adam([pt], 2e-4, betas=(0.5, 0.999))
  pd, m, v = p0.clone().cuda(), torch.zeros(1001).cuda(), torch.zeros(1001).cuda()
  for step in range(1, 4):
    gr = torch.randn(1001, generator=g)
    pt.grad = gr.clone()
    opt.step()
    ops.adam_step(pd, gr.cuda(), m, v, 2e-4, 0.5, 0.999, 1e-8, step)
  assert torch.allclose(pd.cpu(), pt.detach(), atol=1e-7, rtol=1e-6)


@pytest.mark.gpu
def test_ssim_vs_reference_golden_and_oracle(hip):
  """csmri_ssim against the reference's own values (F9) and the oracle on a
  256x256 batch; fp32, tolerance 2e-5 absolute on values in [0, 1]."""
  from csmri_hip import ops
  f = np.load(os.path.join(GOLDEN, 'F9_ssim.npz'))
  pred, target = torch.from_numpy(f['pred']), torch.from_numpy(f['target'])
  got = ops.ssim(pred.permute(0, 2, 3, 1).contiguous().cuda(), target.permute(0, 2, 3, 1).contiguous().cuda())
  assert np.allclose(got.cpu().numpy(), f['ssim_per_image'], rtol=0, atol=2e-5), (got, f['ssim_per_image'])
  g = torch.Generator().manual_seed(3)
  t2 = torch.rand(2, 2, 256, 200, generator=g)
  p2 = t2 + 0.1 * torch.randn(2, 2, 256, 200, generator=g)
  want = O.ssim_images(p2, t2)
  got2 = ops.ssim(p2.permute(0, 2, 3, 1).contiguous().cuda(), t2.permute(0, 2, 3, 1).contiguous().cuda())
  assert np.allclose(got2.cpu().numpy(), want, rtol=0, atol=2e-5), (got2, want)


@pytest.mark.parametrize('size', [64, 256])
def test_device_forward_model_matches_host_synthesis(hip, size):
  """csmri_undersample (fp32 FFTs on the GPU) against the host forward model in
  complex128 (the oracle's synth_batch, reference compressed_sensing.py:460-512): kspace and inp
  within 2e-6 of max|.| (fp32 FFT rounding), sampled set bit-exact (zeros stay exact zeros)."""
  from data.synthetic import synth_batch_device
  want = O.synth_batch(3, size, size, acc=4, seed=31)
  got = synth_batch_device(3, size, size, acc=4, seed=31)
  for k in ('inp', 'kspace', 'target'):
    g = got[k].permute(0, 3, 1, 2).cpu()
    scale = float(want[k].abs().max())
    err = float((g - want[k]).abs().max())
    assert err < 2e-6 * max(scale, 1.0), (k, err, scale)
  assert torch.equal(got['mask'].cpu(), want['mask'])
  ks = got['kspace'].cpu()
  m = want['mask'][:, 0] != 0
  assert float(ks[~m].abs().max()) == 0.0


def test_radial_512_forward_model_and_dc(hip):
  """BASELINE config 5 data format (512x512, radial golden-angle spokes, mask NOT constant along
  W): the per-element uint8 mask path of the forward-model and data-consistency kernels against
  the host complex128 arithmetic; mask conversion bit-exact."""
  ops = hip.ops
  n, nx, spokes = 2, 512, 70
  m = O.radial_mask((n, nx, nx), spokes, rand=True, golden_angle=True, centred=False,
                    rng=np.random.RandomState(4321))
  img = np.stack([O.phantom(nx, nx, 50 + i) for i in range(n)])
  k_u = m * np.fft.fft2(img.astype(np.complex128), norm='ortho')
  x_u = np.fft.ifft2(k_u, norm='ortho')
  mt = torch.from_numpy(np.stack((m, m), 1).astype(np.float32))
  mu8 = ops.mask_to_u8(mt.cuda())
  assert torch.equal(mu8.cpu(), torch.from_numpy(m.astype(np.uint8)))
  tgt = torch.from_numpy(np.stack((img, np.zeros_like(img)), -1).astype(np.float32)).cuda()
  ks, inp = ops.undersample(tgt, mu8)
  want_k = torch.from_numpy(np.stack((k_u.real, k_u.imag), -1).astype(np.float32))
  want_x = torch.from_numpy(np.stack((x_u.real, x_u.imag), -1).astype(np.float32))
  assert float((ks.cpu() - want_k).abs().max()) < 2e-6 * float(want_k.abs().max())
  assert float((inp.cpu() - want_x).abs().max()) < 2e-6 * max(1.0, float(want_x.abs().max()))
  # data consistency with that mask on a perturbed image: (1-m) FFT(x) + k0
  g = torch.Generator().manual_seed(5)
  x = want_x + 0.05 * torch.randn(want_x.shape, generator=g)
  ref = O.dc_layer(x.permute(0, 3, 1, 2).double(), want_k.permute(0, 3, 1, 2).double(), mt.double()).float()
  out, _ = ops.dc_raw(x.cuda(), ks, mu8)
  got = out.cpu().permute(0, 3, 1, 2)
  assert rel_l2(got, ref) < 5e-6 and torch.allclose(got, ref, rtol=1e-4, atol=2e-5)


def test_fp32_kernels_exact_next_to_mfma_kernels_in_one_graph(hip):
  """Two branches of one hipGraph: a chain of data-consistency layers (fp32 VALU + LDS) next to
  a chain of MFMA convolutions.  The DC results must be bit-identical to the serial run.  This
  failed on MI355X while the fp32 kernels were built with packed-fp32 VALU instructions
  (csrc/Makefile): it guards the build flag and every multi-stream graph."""
  ops = hip.ops
  b, size, depth = 8, 256, 4
  g = torch.Generator().manual_seed(3)
  x = torch.randn(b, size, size, 2, generator=g).cuda()
  k0 = torch.randn(b, size, size, 2, generator=g).cuda()
  m8 = (torch.rand(b, size, size, generator=g) < 0.25).to(torch.uint8).cuda()
  wt = torch.randn(64, 64, 3, 3, generator=g) * 0.05
  layer = ops.ConvLayer(torch.nn.Parameter(wt.cuda()), None, 1, (1, 1, 1, 1), 'zero', torch.bfloat16)
  act = torch.randn(b, size, size, 64, generator=g).bfloat16().cuda()

  def dc_chain():
    t, outs = x, []
    for _ in range(depth):
      t, _ = ops.dc_raw(t, k0, m8, None)
      outs.append(t)
    return outs

  def conv_chain():
    t = act
    for _ in range(6):
      t, _ = ops.conv_forward(layer, t, use_bias=False, act_slope=0.2)
    return t

  with torch.no_grad():
    ref = [t.clone() for t in dc_chain()]
    ref_conv_out = conv_chain().clone()
  s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

  def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.no_grad():
      with torch.cuda.stream(s2):
        c = conv_chain()
      with torch.cuda.stream(s1):
        d = dc_chain()
    cur.wait_stream(s1)
    cur.wait_stream(s2)
    return d, c

  for _ in range(2):
    both()
  torch.cuda.synchronize()
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph, capture_error_mode='thread_local'):
    d, c = both()
  for _ in range(10):
    graph.replay()
    torch.cuda.synchronize()
    for got, want in zip(d, ref):
      assert torch.equal(got, want)
    assert torch.equal(c, ref_conv_out)


def test_torch_elementwise_kernels_exact_next_to_mfma_kernels_in_one_graph(hip):
  """Same hazard as the test above, for the kernels this library does not build: the wheel's own
  elementwise / reduction / cat kernels (the criteria algebra, the image pool and the optimizer glue
  use them inside the captured step) run on a second branch of one hipGraph next to a chain of MFMA
  convolutions and must stay bit-identical to their serial results."""
  ops = hip.ops
  g = torch.Generator().manual_seed(1)
  b, size = 8, 256
  wt = torch.randn(64, 64, 3, 3, generator=g) * 0.05
  layer = ops.ConvLayer(torch.nn.Parameter(wt.cuda()), None, 1, (1, 1, 1, 1), 'zero', torch.bfloat16)
  act = torch.randn(b, size, size, 64, generator=g).bfloat16().cuda()
  xf = [torch.randn(b * size * size * 8, generator=g).cuda() for _ in range(3)]
  xb = [t.bfloat16() for t in xf]

  def torch_chain():
    outs = []
    t = xf[0]
    for i in range(6):
      t = t + xf[1 + i % 2]
      outs.append(t)
    u = xb[0]
    for i in range(6):
      u = u + xb[1 + i % 2]
      outs.append(u)
    v = xf[0]
    for i in range(4):
      v = v * 1.0001 + xf[1]
      outs.append(v)
    outs.append(torch.stack([t.sum() for t in xf]))
    outs.append(torch.cat([xb[0][:4096], xb[1][:4096]]) * 0.5)
    return outs

  def conv_chain():
    t = act
    for _ in range(10):
      t, _ = ops.conv_forward(layer, t, use_bias=False, act_slope=0.2)
    return t

  with torch.no_grad():
    ref = [t.clone() for t in torch_chain()]
    ref_conv_out = conv_chain().clone()
  s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

  def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.no_grad():
      with torch.cuda.stream(s2):
        c = conv_chain()
      with torch.cuda.stream(s1):
        a = torch_chain()
    cur.wait_stream(s1)
    cur.wait_stream(s2)
    return a, c

  for _ in range(2):
    both()
  torch.cuda.synchronize()
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph, capture_error_mode='thread_local'):
    a, c = both()
  for _ in range(10):
    graph.replay()
    torch.cuda.synchronize()
    for i, (got, want) in enumerate(zip(a, ref)):
      assert torch.equal(got, want), i
    assert torch.equal(c, ref_conv_out)


@pytest.mark.parametrize('shape', [(2, 64, 64), (1, 256, 256), (2, 128, 32), (1, 512, 512), (3, 32, 64)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_fft2_standalone_matches_numpy(hip, shape):
  """csmri_fft2 (the reference's Fft2d / Ifft2d, myfft.py:78-128) against numpy.fft -- the same
  known-answer relation the reference's own myfft.py __main__ block checks (:186-189,203-207) -- in both
  normalisations, plus the autograd adjoint (myfft.py:92-102) and the round trip."""
  ops = hip.ops
  b, h, w = shape
  rng = np.random.RandomState(h + w)
  x = (rng.randn(b, h, w) + 1j * rng.randn(b, h, w)).astype(np.complex64)
  xd = torch.from_numpy(np.stack((x.real, x.imag), -1)).cuda()
  for inverse in (False, True):
    for ortho in (True, False):
      fn = np.fft.ifft2 if inverse else np.fft.fft2
      ref = fn(x.astype(np.complex128), norm='ortho' if ortho else None)
      got = ops.fft2(xd, inverse, ortho).cpu().numpy()
      got = got[..., 0] + 1j * got[..., 1]
      err = np.abs(got - ref).max() / np.abs(ref).max()
      print('fft2 %s inverse=%d ortho=%d rel max err %.2e' % (shape, inverse, ortho, err))
      assert err < 2e-6, (inverse, ortho, err)
  back = ops.fft2(ops.fft2(xd, False, True), True, True)
  assert float((back - xd).abs().max()) < 5e-6 * float(xd.abs().max()) * 4
  # adjoint: <F x, g> = <x, F^H g>
  xg = xd.clone().requires_grad_(True)
  g = torch.randn(xd.shape, generator=torch.Generator().manual_seed(1)).cuda()
  (ops.Fft2d.apply(xg) * g).sum().backward()
  gref = np.fft.ifft2((g[..., 0].cpu().numpy() + 1j * g[..., 1].cpu().numpy()).astype(np.complex128), norm='ortho')
  gg = xg.grad.cpu().numpy()
  assert np.abs((gg[..., 0] + 1j * gg[..., 1]) - gref).max() < 2e-6 * np.abs(gref).max() + 1e-7


@pytest.mark.parametrize('n,p', [(4, 0.5), (40960, 0.5), (4099, 0.5), (1001, 0.2), (1 << 18, 0.75)])
def test_dropout2d_mask_matches_the_philox_oracle_bit_for_bit(hip, n, p):
  """csmri_dropout2d_mask (nn.Dropout2d draws of reference models/discriminators.py:150-152) against the oracle's
  numpy Philox4x32-10 (pinned to the generator's published known-answer vectors on the CPU side): every mask value
  equal, the call counter advances by one per launch, untouched memory behind the mask stays untouched."""
  seed = 0x0123456789abcdef ^ n
  st = torch.tensor([seed, 5, 0], dtype=torch.int64, device='cuda')      # {seed, call, finished workgroups}
  out = torch.full((n + 5,), -3.0, device='cuda')
  for call in (5, 6):
    hip.lib.call('csmri_dropout2d_mask', out.data_ptr(), n, p, st.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert st.tolist() == [seed, call + 1, 0]
    want = O.dropout2d_mask(seed, call, n, p)
    assert torch.equal(out[:n].cpu(), want), (n, p, call)
    assert bool((out[n:] == -3.0).all())
  assert hip.lib.raw('csmri_dropout2d_mask')(out.data_ptr(), n, 1.0, st.data_ptr(), None) == -1     # p < 1

