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

_STATE = {'dt': None, 'gscale': 1.0}


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
    wq = _rnd(w) if w.requires_grad else _q(w)
    return _rnd(RealF.conv2d(x, wq, b, **kw))
  q.conv2d = conv2d
  q.leaky_relu = lambda x, s=0.01, *a, **k: _rnd(RealF.leaky_relu(x, s))
  q.relu = lambda x, *a, **k: _rnd(RealF.relu(x))
  q.max_pool2d = lambda *a, **k: _rnd(RealF.max_pool2d(*a, **k))
  return q


@contextlib.contextmanager
def emulate(mode, gscale=1.0):
  """mode: 'bf16', 'fp16' or None (no-op).  gscale: static power-of-two gradient scale applied
  before rounding gradient tensors and removed after (fp16 needs one; exact for bf16)."""
  if mode is None:
    yield
    return
  _STATE['dt'] = {'bf16': torch.bfloat16, 'fp16': torch.float16}[mode]
  _STATE['gscale'] = float(gscale)
  O.F = _patched_functional()
  try:
    yield
  finally:
    O.F = RealF
    _STATE['dt'] = None
    _STATE['gscale'] = 1.0
