"""The fp8 convolution variant (BASELINE.json config 5 "fp8 MFMA convs"; VERDICT row ns-1).

The reference has no fp8 path (its convolutions are fp32 nn.Conv2d: models/unet.py:40-52,
models/discriminators.py:137-172), so the oracle for this variant is the reference convolution applied to
operands rounded the way csrc/fp8.hip rounds them -- oracle/csmri_lowprec.py restates that rounding
(e4m3fn, ties to even, one power-of-two scale per tensor) and is itself pinned here to torch's
float8_e4m3fn conversion on the CPU.

Stated tolerances:
  * quantised bytes and scales: bit-exact (byte work);
  * exact-integer operands: the fp8 convolution equals the integer convolution exactly (pins the
    operand lane map of v_mfma_scale_f32_16x16x128_f8f6f4 and the K order of the LDS tiles);
  * random operands, fp32 output: relative L2 <= 5e-5 against the oracle convolution on the same
    fp8-rounded operands (measured 1.2e-5..1.6e-5: the block-scaled MFMA sums its 128 products in a wider
    fixed-point tree and rounds once per instruction, not as 128 IEEE fp32 additions); bf16 output: <= 3e-3;
  * against the un-quantised fp32 convolution the variant's own error is the format's: relative L2
    2.5e-2..4.5e-2 for Gaussian operands (asserted < 6e-2) -- that is what "fp8" costs, not a kernel property."""
import math
import zlib

import pytest
import torch
import torch.nn.functional as F

import csmri_oracle as O
import csmri_lowprec as L


def rel_l2(a, b):
  return float((a - b).norm() / (b.norm() + 1e-30))


# ---------------------------------------------------------------------------------------------------
# CPU: the restated rounding against torch's own e4m3fn conversion
# ---------------------------------------------------------------------------------------------------
def test_e4m3_restatement_matches_torch_float8():
  g = torch.Generator().manual_seed(0)
  x = torch.cat([torch.randn(100000, generator=g) * s for s in (1e-3, 0.1, 1.0, 30.0, 200.0)])
  edge = torch.tensor([0.0, -0.0, 448.0, -448.0, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -9, 2.5 * 2.0 ** -9, 2.0 ** -6,
                       0.0175, 17.0, 18.0, 19.0, 463.9, 1000.0])
  x = torch.cat([x, edge])
  ref = x.clamp(-448, 448).to(torch.float8_e4m3fn)
  got = L.e4m3_round(x)
  assert torch.equal(got, ref.to(torch.float32))
  assert torch.equal(L.e4m3_bits(got), ref.view(torch.uint8))


def test_fp8_scale_rule():
  # scale = 2^(7 - floor(log2 amax)): scaled maximum in [128, 256)
  for amax in (1.0, 0.99999, 3.7, 1e-3, 447.0, 1e4, 2.0 ** -20):
    s = L.fp8_scale(amax)
    assert 128.0 <= amax * s < 256.0 and math.log2(s) == int(math.log2(s))
  assert L.fp8_scale(0.0) == 1.0


def test_fp8_emulated_conv_backward_is_the_16bit_path():
  """emulate(fp8=True): forward on fp8 operands, gradients as the bf16 emulation computes them."""
  g = torch.Generator().manual_seed(1)
  x = torch.randn(2, 128, 8, 8, generator=g).bfloat16().float().requires_grad_(True)
  w = (torch.randn(64, 128, 3, 3, generator=g) * 0.05).requires_grad_(True)
  gy = torch.randn(2, 64, 8, 8, generator=g)
  with L.emulate('bf16', fp8=True):
    y8 = O.F.conv2d(x, w, None, padding=1)
    y8.backward(gy)
  gx8, gw8 = x.grad.clone(), w.grad.clone()
  x.grad = w.grad = None
  with L.emulate('bf16'):
    y16 = O.F.conv2d(x, w, None, padding=1)
    y16.backward(gy)
  assert 1e-2 < rel_l2(y8, y16) < 6e-2            # the forward differs by the fp8 rounding ...
  assert rel_l2(gx8, x.grad) < 1e-6 and rel_l2(gw8, w.grad) < 5e-3   # ... the backward does not (bf16 rounding of dW only)


# ---------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def hip():
  import csmri_hip
  assert torch.cuda.is_available()
  return csmri_hip


def to_dev_nhwc(x, dtype=torch.bfloat16):
  return x.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()


def from_dev_nhwc(t, c):
  return t.float().cpu()[..., :c].permute(0, 3, 1, 2).contiguous()


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
def test_quantize_fp8_bit_exact(hip, dtype):
  ops = hip.ops
  g = torch.Generator().manual_seed(5)
  for scale, n in ((1.0, 1 << 20), (1e-4, 4096), (300.0, 65536), (0.0, 1024), (7.3e5, 16 * 999)):
    x = (torch.randn(n, generator=g) * scale).to(dtype)
    if scale:
      x[::97] = 0
      x[5] = -x.abs().max() * 1.0          # the maximum is negative
    xd = x.cuda()
    amax = ops.absmax(xd)
    q, sc = ops.quantize_fp8(xd)
    torch.cuda.synchronize()
    ref, s = L.quantize_fp8(x.float())
    assert float(amax.cpu()) == float(x.float().abs().max())
    assert sc.cpu().tolist() == [s, 1.0 / s], (sc.cpu().tolist(), s)
    assert torch.equal(q.cpu(), L.e4m3_bits(ref)), (scale, n)


# name, cin, cout, k, stride, border, upsample, H, W, B, c0
CONV_CASES = [
    ('vgg2_2', 128, 128, 3, 1, 'zero', False, 32, 32, 4, None),
    ('vgg3_2', 256, 256, 3, 1, 'zero', False, 16, 24, 3, None),
    ('unet_e2b', 128, 128, 4, 1, 'reflection', False, 32, 32, 2, None),
    ('unet_up', 128, 64, 4, 1, 'reflection', True, 16, 16, 2, None),
    ('unet_cat64', 128, 64, 4, 1, 'reflection', False, 32, 32, 2, 64),
    ('unet_cat128', 256, 128, 4, 1, 'reflection', False, 16, 16, 2, 128),
    ('unet_cat_uneven', 256, 64, 3, 1, 'zero', False, 16, 16, 2, 80),
    ('disc3', 128, 256, 4, 2, 'reflection', False, 32, 32, 4, None),
    ('disc5_splitk', 512, 1024, 4, 2, 'reflection', False, 8, 8, 2, None),
    ('disc6_tail', 1024, 1024, 4, 1, 'reflection', False, 5, 7, 3, None),
    ('big_m', 128, 64, 3, 1, 'zero', False, 128, 128, 4, None),
]


def _layer(ops, name, cin, cout, k, stride, border, up, gen):
  wt = torch.randn(cout, cin, k, k, generator=gen) / math.sqrt(cin * k * k)
  bias = torch.randn(cout, generator=gen) * 0.1
  pads, mode = O.same_padding(k, stride), border
  layer = ops.ConvLayer(torch.nn.Parameter(wt.clone().cuda()), torch.nn.Parameter(bias.clone().cuda()), stride, pads, mode,
                        torch.bfloat16, upsample=up)
  layer.fp8 = True
  return layer, wt, bias, pads, mode


@pytest.mark.gpu
def test_fp8_conv_exact_on_integer_operands(hip):
  """Integers up to 8 are exact in e4m3 and every partial sum is exact in fp32: the fp8 convolution must
  equal the integer convolution bit for bit, for every lane / K position (asymmetric operands)."""
  ops = hip.ops
  g = torch.Generator().manual_seed(2)
  for cin, cout, k, stride, c0 in ((128, 128, 3, 1, None), (256, 64, 4, 2, None), (256, 128, 3, 1, 64)):
    x = torch.randint(-8, 9, (2, cin, 12, 20), generator=g).float()
    wt = torch.randint(-8, 9, (cout, cin, k, k), generator=g).float()
    pads = O.same_padding(k, stride)
    layer = ops.ConvLayer(torch.nn.Parameter(wt.clone().cuda()), None, stride, pads, 'reflection', torch.bfloat16)
    layer.fp8 = True
    x0, x1 = (to_dev_nhwc(x), None) if c0 is None else (to_dev_nhwc(x[:, :c0]), to_dev_nhwc(x[:, c0:]))
    log = ops.LAUNCH_LOG = []
    try:
      y, _ = ops.conv_forward(layer, x0, x1, False, 1.0, False, torch.float32)
    finally:
      ops.LAUNCH_LOG = None
    assert log[-1][1].startswith('gconv_fp8_kernel'), log
    ref = F.conv2d(O.pad2d(x, pads, 'reflection'), wt, None, stride=stride)
    assert torch.equal(from_dev_nhwc(y, cout), ref), (cin, cout, k, stride, c0)


@pytest.mark.gpu
@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_fp8_conv_vs_oracle_on_rounded_operands(hip, case):
  ops = hip.ops
  name, cin, cout, k, stride, border, up, h, w, b, c0 = case
  g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 100000)
  layer, wt, bias, pads, mode = _layer(ops, name, cin, cout, k, stride, border, up, g)
  x = (torch.randn(b, cin, h, w, generator=g) * 1.7).bfloat16().float()
  x0, x1 = (to_dev_nhwc(x), None) if c0 is None else (to_dev_nhwc(x[:, :c0]), to_dev_nhwc(x[:, c0:]))
  # oracle: the reference convolution on the fp8-rounded operands
  xq, sx = L.quantize_fp8(x)
  wq, sw = L.quantize_fp8(wt)
  prep = (lambda t: F.interpolate(t, scale_factor=2, mode='nearest')) if up else (lambda t: t)
  ref8 = F.conv2d(O.pad2d(prep(xq), pads, mode), wq, None, stride=stride) / (sx * sw) + bias.view(1, -1, 1, 1)
  ref8 = F.leaky_relu(ref8, 0.2)
  ref32 = F.leaky_relu(F.conv2d(O.pad2d(prep(x), pads, mode), wt, bias, stride=stride), 0.2)
  log = ops.LAUNCH_LOG = []
  try:
    y32, _ = ops.conv_forward(layer, x0, x1, True, 0.2, False, torch.float32)
    y16, stats = ops.conv_forward(layer, x0, x1, True, 1.0, True, None)
    torch.cuda.synchronize()
  finally:
    ops.LAUNCH_LOG = None
  assert all(e[1].startswith('gconv_fp8_kernel') for e in log), log
  e32 = rel_l2(from_dev_nhwc(y32, cout), ref8)
  fmt = rel_l2(ref8, ref32)
  print('%-16s fp8 conv vs oracle(fp8 operands) %.2e | format error vs fp32 conv %.2e | %s sk%d' %
        (name, e32, fmt, log[0][1], log[0][2]))
  assert e32 < 5e-5, (name, e32)
  assert fmt < 6e-2, (name, fmt)
  # bf16 output + BatchNorm partial sums from the epilogue (no activation on this call)
  pre8 = F.conv2d(O.pad2d(prep(xq), pads, mode), wq, None, stride=stride) / (sx * sw) + bias.view(1, -1, 1, 1)
  assert rel_l2(from_dev_nhwc(y16, cout), pre8) < 3e-3
  s1, s2 = stats.cpu().double().reshape(2, layer.cout_p, -1).sum(-1)     # partial sums are [2][C][rows]
  assert torch.allclose(s1[:cout], pre8.double().sum((0, 2, 3)), rtol=1e-4, atol=1e-2)
  assert torch.allclose(s2[:cout], (pre8.double() ** 2).sum((0, 2, 3)), rtol=1e-4, atol=1e-2)


@pytest.mark.gpu
def test_fp8_layer_backward_is_the_bf16_path(hip):
  """ConvLayer.fp8 only changes the forward product: dgrad / wgrad launches and results are those of the
  bf16 layer (the ConvAct autograd node saves the bf16 input)."""
  ops = hip.ops
  g = torch.Generator().manual_seed(9)
  outs = []
  for fp8 in (False, True):
    gen = torch.Generator().manual_seed(11)
    layer, wt, bias, pads, mode = _layer(ops, 'l', 128, 128, 3, 1, 'zero', False, gen)
    layer.fp8 = fp8
    x = to_dev_nhwc(torch.randn(2, 128, 16, 16, generator=gen)).requires_grad_(True)
    gy = to_dev_nhwc(torch.randn(2, 128, 16, 16, generator=gen))
    y = ops.ConvAct.apply(x, None, layer.weight, layer.bias, layer, 1.0, None)
    y.backward(gy)
    ops.join_wgrad_stream()
    torch.cuda.synchronize()
    outs.append((y.detach().float().cpu(), x.grad.float().cpu(), layer.weight.grad.cpu().clone()))
  (y16, gx16, gw16), (y8, gx8, gw8) = outs
  assert 5e-3 < rel_l2(y8, y16) < 6e-2
  # no activation on this layer, so nothing of the forward's output enters the backward: identical launches
  assert torch.equal(gx8, gx16) and torch.equal(gw8, gw16)


# ---------------------------------------------------------------------------------------------------
# The frozen VGG stack in fp8 (ops.Fp8Chain): the patch-structured 3 x 3 kernel with e4m3fn operands, the fp8 copy of an
# output written by the producing kernel's epilogue, delayed scaling.  The perceptual loss of the reference
# (models/vgg_loss.py:43-65, models/vgg.py:35) has no fp8 form: the oracle is the reference's conv / ReLU / max-pool on
# operands rounded as fp8.hip rounds them.
# ---------------------------------------------------------------------------------------------------
def _frozen_layer(ops, cin, cout, gen, integer=False, density=1.0):
  if integer:
    wt = torch.randint(-4, 5, (cout, cin, 3, 3), generator=gen).float()
    if density < 1.0:
      wt = wt * (torch.rand(wt.shape, generator=gen) < density)
    bias = torch.randint(-3, 4, (cout,), generator=gen).float()
  else:
    wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (cin * 9))
    bias = torch.randn(cout, generator=gen) * 0.1
  layer = ops.ConvLayer(torch.nn.Parameter(wt.clone().cuda()), torch.nn.Parameter(bias.clone().cuda()), 1, (1, 1, 1, 1), 'zero',
                        torch.bfloat16, frozen=True)
  return layer, wt, bias


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(64, 128, 128, 40, 24, 3), (128, 256, 256, 16, 16, 5), (256, 128, 128, 32, 48, 2)],
                         ids=['64-128-128', '128-256-256', '256-128-128'])
def test_fp8_patch_conv_chain_exact_on_integer_operands(hip, shape):
  """Two frozen 3 x 3 layers chained through the fp8 copy: layer A (bf16 operands) writes y_A and, from the same epilogue,
  its fp8 copy (scale 1) and the running maximum; layer B multiplies that copy with fp8 weights.  With small integers
  every product and every fp32 partial sum is exact, so
    * the fp8 copy equals y_A element for element (e4m3 holds the integers 0..16 exactly; larger ones are rounded by
      the restated rule), the maximum is max |y_A|;
    * y_B equals the integer convolution of the ROUNDED copy with B's weights, rounded once to bf16 -- bit for bit: this
      pins the operand lane map of v_mfma_scale_f32_16x16x128_f8f6f4 in the patch kernel and the K order of its LDS
      images for every tap and channel chunk (ragged tiles: the map is not a multiple of 16)."""
  ops = hip.ops
  c0, c1, c2, h, w, b = shape
  g = torch.Generator().manual_seed(c0 + c1 + h)
  la, wa, ba = _frozen_layer(ops, c0, c1, g, integer=True, density=0.1)
  lb, wb, bb = _frozen_layer(ops, c1, c2, g, integer=True)
  # sparse small-integer input and first-layer weights: y_A stays within a few tens
  x = (torch.randint(-2, 3, (b, c0, h, w), generator=g) * (torch.rand(b, c0, h, w, generator=g) < 0.05)).float()
  plan = [('conv', la, 0.0), ('conv', lb, 0.0)]
  chain = ops.Fp8Chain(plan, torch.device('cuda'))
  assert chain.slots == {0: 0}
  chain.ready = True                                   # scales given: 1.0 (integers)
  xd = to_dev_nhwc(x)
  log = ops.LAUNCH_LOG = []
  try:
    ya, yq = ops.frozen_conv_forward(la, xd, 0.0, None, 0, 0, chain)
    yb, _ = ops.frozen_conv_forward(lb, ya, 0.0, yq, chain.dq_scale_ptr(0), None, chain)
    torch.cuda.synchronize()
  finally:
    ops.LAUNCH_LOG = None
  names = [e[1] for e in log]
  assert names[0].startswith('pconv2_kernel<3, 3,') and names[0].endswith('false>'), names
  assert names[1].startswith('pconv2_kernel<3, 3,') and names[1].endswith('true>'), names
  ra = torch.relu(F.conv2d(x, wa, ba, padding=1))
  assert torch.equal(from_dev_nhwc(ya, c1), ra.bfloat16().float())
  ya_host = from_dev_nhwc(ya, c1)
  q_ref = L.e4m3_round(ya_host)                          # scale 1
  assert torch.equal(yq.cpu()[..., :c1].permute(0, 3, 1, 2), L.e4m3_bits(q_ref))
  assert float(chain.amax[0].cpu()) == float(ya_host.abs().max())
  wq, sw = L.quantize_fp8(wb)
  assert torch.equal(wq / sw, wb)                        # small integers: exact in e4m3
  rb = torch.relu(F.conv2d(q_ref, wb, bb, padding=1))
  assert torch.equal(from_dev_nhwc(yb, c2), rb.bfloat16().float())


@pytest.mark.gpu
def test_fp8_patch_conv_vs_oracle_and_copy_is_quantize_of_output(hip):
  """Random operands at a VGG shape (conv3_x: 256 -> 256, 64 x 64, 4 images): the fp8 copy a layer's epilogue writes is
  bit for bit csmri_quantize_fp8's rounding of its stored bf16 output with the chain's scale; the consuming fp8 layer agrees
  with the reference convolution on the SAME rounded operands within bf16 output rounding (relative L2 <= 3e-3), and its
  distance to the un-quantised convolution is the format's (< 6e-2)."""
  ops = hip.ops
  g = torch.Generator().manual_seed(31)
  la, wa, ba = _frozen_layer(ops, 128, 256, g)
  lb, wb, bb = _frozen_layer(ops, 256, 256, g)
  x = torch.relu(torch.randn(4, 128, 64, 64, generator=g)).bfloat16().float()
  chain = ops.Fp8Chain([('conv', la, 0.0), ('conv', lb, 0.0)], torch.device('cuda'))
  xd = to_dev_nhwc(x)
  # pass 1 (bf16, collects the maximum), scales for pass 2 with one bit of headroom
  ya1, yq1 = ops.frozen_conv_forward(la, xd, 0.0, None, 0, 0, chain)
  assert yq1 is None
  chain.finish()
  torch.cuda.synchronize()
  ya_host = from_dev_nhwc(ya1, 256)
  amax = float(ya_host.abs().max())
  s = L.fp8_scale(amax) / 2.0
  assert chain.scales.cpu().tolist() == [[s, 1.0 / s]] and float(chain.amax[0].cpu()) == 0.0
  ya, yq = ops.frozen_conv_forward(la, xd, 0.0, None, 0, 0, chain)
  yb, _ = ops.frozen_conv_forward(lb, ya, 0.0, yq, chain.dq_scale_ptr(0), None, chain)
  torch.cuda.synchronize()
  assert torch.equal(ya, ya1) and float(chain.amax[0].cpu()) == amax
  qa = L.e4m3_round(ya_host * s)
  assert torch.equal(yq.cpu().permute(0, 3, 1, 2), L.e4m3_bits(qa))
  wq, sw = L.quantize_fp8(wb)
  ref8 = torch.relu(F.conv2d(qa, wq, None, padding=1) / (s * sw) + bb.view(1, -1, 1, 1))
  ref32 = torch.relu(F.conv2d(ya_host, wb, bb, padding=1))
  e8, fmt = rel_l2(from_dev_nhwc(yb, 256), ref8), rel_l2(ref8, ref32)
  print('fp8 patch conv vs oracle on the same fp8 operands %.2e | format error vs the bf16-operand convolution %.2e' % (e8, fmt))
  assert e8 < 3e-3 and fmt < 6e-2, (e8, fmt)


@pytest.mark.gpu
def test_fp8_copies_saturate_when_activations_outgrow_the_delayed_scale(hip):
  """Delayed scaling takes step t's scale from step t - 1's maximum with one bit of headroom, so max * scale lies in
  [64, 128): a tensor that grows 8x between two forwards (train -> validation switch, a differently scaled slice) lands
  at 512 ... 1024, outside e4m3fn (448).  The producing epilogues (patch-conv copy, max-pool copy) clamp before
  v_cvt_pk_fp8_f32: the copy is the SATURATED rounding of y * scale (no NaN byte 0x7f / 0xff), the consuming fp8 layer's
  output is finite, and the maximum collected in this pass repairs the next pass's scale."""
  ops = hip.ops
  g = torch.Generator().manual_seed(77)
  la, wa, ba = _frozen_layer(ops, 128, 128, g)
  lb, wb, bb = _frozen_layer(ops, 128, 128, g)
  x = torch.relu(torch.randn(2, 128, 32, 32, generator=g)).bfloat16().float()
  plan = [('conv', la, 0.0), ('pool', None, None), ('conv', lb, 0.0)]
  chain = ops.Fp8Chain([('conv', la, 0.0), ('conv', lb, 0.0)], torch.device('cuda'))
  xd = to_dev_nhwc(x)
  ops.frozen_conv_forward(la, xd, 0.0, None, 0, 0, chain)          # pass 1: bf16, collects the maximum
  chain.finish()
  torch.cuda.synchronize()
  s = float(chain.scales[0, 0].cpu())
  xd8 = to_dev_nhwc(x * 8.0)                                       # pass 2: 8x larger activations, last pass's scale
  ya, yq = ops.frozen_conv_forward(la, xd8, 0.0, None, 0, 0, chain)
  yb, _ = ops.frozen_conv_forward(lb, ya, 0.0, yq, chain.dq_scale_ptr(0), None, chain)
  # the max-pool's copy with the same stale scale
  yp, _, ypq = ops.maxpool2_fwd(ya, chain.q_scale_ptr(0), chain.amax_ptr(0), want_q=True)
  torch.cuda.synchronize()
  ya_host = from_dev_nhwc(ya, 128)
  assert float(ya_host.max()) * s > 448.0, 'the case must overflow the stale scale'
  for name, q, src in (('conv copy', yq, ya_host), ('pool copy', ypq, F.max_pool2d(ya_host, 2, 2))):
    bits = q.cpu().permute(0, 3, 1, 2)
    assert int(((bits & 0x7f) == 0x7f).sum()) == 0, name + ': NaN bytes in the fp8 copy'
    want = L.e4m3_bits(L.e4m3_round(torch.clamp(src * s, -448.0, 448.0)))
    assert torch.equal(bits, want), name
    assert int((bits == 0x7e).sum()) > 0, name + ': nothing saturated (0x7e = 448)'
  assert bool(torch.isfinite(yb.float()).all())
  chain.finish()
  torch.cuda.synchronize()
  assert float(chain.scales[0, 0].cpu()) <= s / 4.0                # the next pass's scale follows the new maximum


@pytest.mark.gpu
def test_maxpool_fp8_copy_is_quantize_of_output(hip):
  ops = hip.ops
  g = torch.Generator().manual_seed(8)
  x = torch.relu(torch.randn(3, 128, 24, 40, generator=g) * 3.0).bfloat16().float()
  sc = torch.tensor([4.0, 0.25], device='cuda')
  am = torch.zeros(1, device='cuda')
  y, arg, yq = ops.maxpool2_fwd(to_dev_nhwc(x), sc.data_ptr(), am.data_ptr(), want_q=True)
  torch.cuda.synchronize()
  ref = F.max_pool2d(x, 2, 2)
  assert torch.equal(from_dev_nhwc(y, 128), ref)
  assert torch.equal(yq.cpu().permute(0, 3, 1, 2), L.e4m3_bits(L.e4m3_round(ref * 4.0)))
  assert float(am.cpu()) == float(ref.abs().max())


@pytest.mark.gpu
def test_vgg19_fp8_chain_features_and_gradient(hip):
  """models.vgg.VGG19 with compute_dtype 'fp8' (frozen stack, ops.Fp8Chain), 128 x 128, 2 + 2 images: the first call
  runs bf16 and leaves scales behind (delayed scaling), the second multiplies fp8 operands in conv2_2 .. conv5_4 (13
  launches of the fp8 patch kernel).  relu5_4 of the fp8 pass against the fp32 CPU oracle (O.vgg_features, reference
  models/vgg.py:58-80) within the format's error (relative L2 < 0.15, 13 layers of ~3.8 % each; the bf16 pass: < 0.02);
  the gradient of the MSE feature loss w.r.t. the prediction (bf16 backward on the saved activations): see below."""
  import sys
  from conftest import PKG
  if PKG not in sys.path:
    sys.path.insert(0, PKG)
  from models.utils import set_default_compute_dtype
  from models.vgg import VGG19
  ops = hip.ops
  set_default_compute_dtype('fp8')
  try:
    vgg = VGG19(seed=3).cuda()
  finally:
    set_default_compute_dtype('bf16')
  assert vgg.fp8
  PV = {k: v.detach().cpu().float() for k, v in vgg.state_dict().items() if k.startswith('blocks.')}
  g = torch.Generator().manual_seed(17)
  p = torch.rand(2, 3, 128, 128, generator=g)
  t = torch.rand(2, 3, 128, 128, generator=g)
  mean, std = torch.tensor(O.VGG_MEAN).view(1, 3, 1, 1), torch.tensor(O.VGG_STD).view(1, 3, 1, 1)

  res = []
  for it in range(2):
    pd = ops.ToNHWC.apply(((p - mean) / std).cuda().requires_grad_(True), torch.bfloat16, 8)
    pd.retain_grad()
    td = ops.ToNHWC.apply(((t - mean) / std).cuda(), torch.bfloat16, 8)
    log = ops.LAUNCH_LOG = []
    try:
      fp, ft = vgg.features_pair(pd, td)
      loss = ((fp[0].float() - ft[0].float()) ** 2).mean()
      loss.backward()
      torch.cuda.synchronize()
    finally:
      ops.LAUNCH_LOG = None
    n8 = sum(1 for e in log if e[1].startswith('pconv2_kernel') and e[1].endswith('true>'))
    res.append((fp[0].detach().float().cpu(), ft[0].detach().float().cpu(), pd.grad.float().cpu()[..., :3], float(loss), n8))
  assert res[0][4] == 0 and res[1][4] == 13, (res[0][4], res[1][4])
  assert vgg._fp8_chain.ready and not vgg._fp8_chain.disabled and len(vgg._fp8_chain.slots) == 13
  pr = ((p - mean) / std).requires_grad_(True)
  fo_p, fo_t = O.vgg_features(PV, pr * std + mean), O.vgg_features(PV, t)
  lo = F.mse_loss(fo_p, fo_t.detach())
  lo.backward()
  go = pr.grad.permute(0, 2, 3, 1)
  for name, (fp, ft, gp, loss, n8) in zip(('bf16 pass', 'fp8 pass'), res):
    ef = rel_l2(fp.permute(0, 3, 1, 2)[:, :512], fo_p.detach())
    cos = float((gp * go).sum() / (gp.norm() * go.norm() + 1e-30))
    print('VGG19 %-9s relu5_4 rel_l2 vs fp32 oracle %.3e | loss %.5e (oracle %.5e) | grad cos %.4f | %d fp8 launches' %
          (name, ef, loss, float(lo), cos, n8))
    assert ef < (0.02 if name == 'bf16 pass' else 0.15), (name, ef)
    # the gradient: bf16 is the established path; for fp8 only the sign of the correlation is asserted -- the MSE of two
    # feature maps whose difference is of the size of the fp8 noise has no direction left (0.39 here, 0.6 / 0.25 for a
    # prediction 30 / 40 dB from its target: DESIGN.md 3.5), which is the measured reason the variant stays opt-in
    assert cos >= (0.90 if name == 'bf16 pass' else 0.2), (name, cos)
    assert abs(loss - float(lo)) <= (0.05 if name == 'bf16 pass' else 1.0) * float(lo)
  g16, g8 = res[0][2], res[1][2]
  cos88 = float((g16 * g8).sum() / (g16.norm() * g8.norm() + 1e-30))
  print('VGG19 fp8 pass against the bf16 pass: relu5_4 rel_l2 %.3e, grad cos %.4f' % (rel_l2(res[1][0], res[0][0]), cos88))
