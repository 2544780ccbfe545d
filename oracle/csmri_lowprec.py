"""Storage-format emulation for the CPU oracle (TEST INFRASTRUCTURE -- never imported by the product).

`emulate('bf16' | 'fp16')` makes oracle/csmri_oracle.py round, in fp32 arithmetic otherwise unchanged,
every tensor a 16-bit-storage implementation keeps in its compute dtype -- convolution outputs,
activation (BatchNorm + LeakyReLU / ReLU) outputs, max-pool outputs, the packed weight copies -- and
the gradients of the same tensors on the way back (accumulation, BatchNorm statistics, losses and
weight gradients stay fp32, as on the HIP path).  The difference between such a run and the plain
fp32 oracle is the error FLOOR of that storage format on a given step: no implementation that stores
activations in bf16 can be closer to the fp32 reference than this, whatever its kernels do.  The GPU
parity tests use it to state bf16 bounds that are about the format's conditioning (the GAN step's
gradients are cancellation-heavy sums) separately from kernel correctness
(which the per-layer tests pin to pure output rounding)."""
import contextlib
import types

import torch
import torch.nn.functional as RealF

import csmri_oracle as O

_STATE = {'dt': None, 'gscale': 1.0, 'fp8': False}


# ---- fp8 (OCP e4m3fn) operand rounding of the fp8 convolution variant -------------------------------------------
# Restates csrc/fp8.hip: one power-of-two scale per tensor, 2^(7 - floor(log2 amax)), so the scaled values stay
# below 256 (< 448, the e4m3fn maximum), then round-to-nearest-even onto the e4m3fn grid: 3 mantissa bits for
# |v| >= 2^-6, the fixed 2^-9 grid of the subnormals below.  Arithmetic is exact in fp32 (powers of two only), so
# the bytes the HIP kernel writes can be checked bit for bit (tests/test_fp8.py).
def e4m3_round(v):
  """fp32 tensor -> nearest e4m3fn value (as fp32), ties to even, |v| clamped to 448."""
  v = v.to(torch.float32).clamp(-448.0, 448.0)
  _, e = torch.frexp(v)                                  # |v| = m * 2^e, m in [0.5, 1)
  quantum = torch.ldexp(torch.ones_like(v), e.clamp(min=-5) - 4)
  return torch.round(v / quantum) * quantum              # torch.round: half to even


def fp8_scale(amax):
  amax = float(amax)
  if not (amax > 0.0) or amax == float('inf'):
    return 1.0
  e = max(int(torch.frexp(torch.tensor(amax, dtype=torch.float32))[1]) - 1, -100)
  return 2.0 ** (7 - e)


def e4m3_bits(v):
  """Byte encoding (sign, 4 exponent bits bias 7, 3 mantissa bits) of values already on the e4m3fn grid."""
  v = v.to(torch.float32)
  sign = (torch.signbit(v)).to(torch.int32) << 7
  a = v.abs()
  m, e = torch.frexp(a)
  normal = a >= 2.0 ** -6
  exp_field = torch.where(normal, e - 1 + 7, torch.zeros_like(e))
  mant = torch.where(normal, torch.round((m * 2 - 1) * 8), torch.round(a * 2.0 ** 9)).to(torch.int32)
  return (sign | (exp_field.to(torch.int32) << 3) | mant).to(torch.uint8)


def quantize_fp8(t, amax=None):
  """(values on the e4m3fn grid, scale): what csmri_quantize_fp8 stores for tensor t."""
  scale = fp8_scale(t.abs().max() if amax is None else amax)
  return e4m3_round(t.to(torch.float32) * scale), scale


def fp8_eligible(x, w):
  """Shape rule of the fp8 variant (include/csmri_hip.h): input channels % 128, output channels % 64."""
  return x.shape[1] % 128 == 0 and w.shape[0] % 64 == 0


class _Fp8Conv(torch.autograd.Function):
  """Forward product on fp8-rounded operands; backward as the 16-bit path (input gradient with the
  bf16-rounded weights, weight gradient with the bf16 input) -- what ops.conv_forward / conv_dgrad /
  conv_wgrad do when ConvLayer.fp8 is set."""

  @staticmethod
  def forward(ctx, x, w, stride, padding):
    xq, sx = quantize_fp8(x)
    wq, sw = quantize_fp8(w)
    ctx.save_for_backward(x, w)
    ctx.conf = (stride, padding)
    return RealF.conv2d(xq, wq, None, stride, padding) / (sx * sw)

  @staticmethod
  def backward(ctx, g):
    x, w = ctx.saved_tensors
    stride, padding = ctx.conf
    gx = torch.nn.grad.conv2d_input(x.shape, _q(w), g, stride, padding) if ctx.needs_input_grad[0] else None
    gw = torch.nn.grad.conv2d_weight(x, w.shape, g, stride, padding) if ctx.needs_input_grad[1] else None
    return gx, gw, None, None


def _q(t):
  dt = _STATE['dt']
  if dt is None:
    return t
  if dt == torch.float16:
    t = t.clamp(-65504.0, 65504.0)
  return t.to(dt).to(torch.float32)


class _Round(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x):
    return _q(x)

  @staticmethod
  def backward(ctx, g):
    s = _STATE['gscale']
    return _q(g * s) / s


def _rnd(x):
  return _Round.apply(x)


def _patched_functional():
  q = types.SimpleNamespace(**{n: getattr(RealF, n) for n in dir(RealF) if not n.startswith('_')})

  def conv2d(x, w, b=None, **kw):
    if _STATE['fp8'] and w.requires_grad and fp8_eligible(x, w) and not set(kw) - {'stride', 'padding'}:
      y = _Fp8Conv.apply(x, w, kw.get('stride', 1), kw.get('padding', 0))
      return _rnd(y if b is None else y + b.view(1, -1, 1, 1))
    wq = _rnd(w) if w.requires_grad else _q(w)
    return _rnd(RealF.conv2d(x, wq, b, **kw))
  q.conv2d = conv2d
  q.leaky_relu = lambda x, s=0.01, *a, **k: _rnd(RealF.leaky_relu(x, s))
  q.relu = lambda x, *a, **k: _rnd(RealF.relu(x))
  q.max_pool2d = lambda *a, **k: _rnd(RealF.max_pool2d(*a, **k))
  return q


@contextlib.contextmanager
def emulate(mode, gscale=1.0, fp8=False):
  """mode: 'bf16', 'fp16' or None (no-op).  gscale: static power-of-two gradient scale applied
  before rounding gradient tensors and removed after (fp16 needs one; exact for bf16).  fp8: the
  forward products of the trainable convolutions whose shape the fp8 variant accepts run on
  e4m3fn-rounded operands (per-tensor power-of-two scales); everything else as `mode`."""
  if mode is None:
    yield
    return
  _STATE['dt'] = {'bf16': torch.bfloat16, 'fp16': torch.float16}[mode]
  _STATE['gscale'] = float(gscale)
  _STATE['fp8'] = bool(fp8)
  O.F = _patched_functional()
  try:
    yield
  finally:
    O.F = RealF
    _STATE['dt'] = None
    _STATE['gscale'] = 1.0
    _STATE['fp8'] = False
