"""Host-side operator layer over libcsmri_hip.so.

PyTorch is used here for device memory (caching allocator), streams and the
autograd tape only -- every tensor op on the training path below is a call into
the C-ABI library.  Activations are NHWC tensors ``[B,H,W,Cp]`` (Cp = channels
padded to a multiple of 8, pad channels zero) in the compute dtype
(torch.bfloat16 or torch.float32); channel slices of wider buffers are allowed
(``stride(3)==1``, dense in B,H,W).
"""
import ctypes as C
import os

import torch

from . import lib
from .lib import F32, BF16, BORDER_ZERO, BORDER_REFLECT

FP8 = 2               # CSMRI_FP8: OCP e4m3fn operands of the fp8 convolution variant

_EPOCH = [0]          # bumped whenever trainable weights change (optimizer step / load)


def bump_weight_epoch():
  _EPOCH[0] += 1


def _weight_epoch(layer):
  """Packed copies of a layer's weights are current while this value is unchanged: the global epoch
  (loads, anything unknown) and the epoch of the layer's PackGroup (its own optimizer's steps)."""
  if layer.frozen:
    return -1
  group = getattr(layer, 'group', None)
  return (_EPOCH[0], group.epoch if group is not None else 0)


def stream():
  return torch.cuda.current_stream().cuda_stream


def dt_of(t):
  if t.dtype == torch.bfloat16:
    return BF16
  if t.dtype == torch.float32:
    return F32
  raise TypeError('unsupported dtype %s' % t.dtype)


def torch_dtype(dt):
  return torch.bfloat16 if dt == BF16 else torch.float32


def pad8(c):
  return (c + 7) // 8 * 8


def is_nhwc(t):
  return (t.dim() == 4 and t.stride(3) == 1 and t.stride(1) == t.shape[2] * t.stride(2) and
          t.stride(0) == t.shape[1] * t.stride(1))


def as_nhwc(t):
  return t if is_nhwc(t) else t.contiguous()


def ptr(t):
  return 0 if t is None else t.data_ptr()


def _need_gpu(t):
  if not t.is_cuda:
    raise RuntimeError('csmri_hip ops need device tensors (no CPU fallback)')


# ----------------------------------------------------------------------------
# layout converters
# ----------------------------------------------------------------------------


def nchw_to_nhwc(x, dtype, cpad=None, split=False):
  """fp32 NCHW [B,C,H,W] -> NHWC [B,H,W,Cpad] of ``dtype`` (zero padded).  ``split`` (2-channel images into 8 bf16
  channels): CSMRI_BF16_SPLIT -- channels 2,3 hold what the bf16 rounding of channels 0,1 dropped."""
  _need_gpu(x)
  x = x.contiguous().float()
  b, c, h, w = x.shape
  cp = pad8(c) if cpad is None else cpad
  out = torch.empty(b, h, w, cp, dtype=dtype, device=x.device)
  split = bool(split) and dtype == torch.bfloat16 and c == 2 and cp == 8 and (h * w) % 4 == 0
  lib.call('csmri_nchw_to_nhwc', x.data_ptr(), b, c, h, w, out.data_ptr(), lib.BF16_SPLIT if split else dt_of(out), cp, cp,
           stream())
  return out


def nhwc_to_nchw(x, c_real):
  """NHWC (any supported dtype, strided channels ok) -> fp32 NCHW [B,c_real,H,W]."""
  _need_gpu(x)
  x = as_nhwc(x)
  b, h, w, _ = x.shape
  out = torch.empty(b, c_real, h, w, dtype=torch.float32, device=x.device)
  lib.call('csmri_nhwc_to_nchw', x.data_ptr(), dt_of(x), x.stride(2), b, c_real, h, w,
           out.data_ptr(), stream())
  return out


class ToNHWC(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x, dtype, cpad, split=False):
    ctx.c = x.shape[1]
    return nchw_to_nhwc(x, dtype, cpad, split)

  @staticmethod
  def backward(ctx, g):
    return nhwc_to_nchw(g, ctx.c), None, None, None


class ToNCHW(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x, c_real):
    ctx.dtype, ctx.cp = x.dtype, x.shape[3]
    return nhwc_to_nchw(x, c_real)

  @staticmethod
  def backward(ctx, g):
    return nchw_to_nhwc(g, ctx.dtype, ctx.cp), None


def mask_to_u8(mask):
  """[B,2,H,W] fp32 {0,1} -> uint8 [B,H,W]; bit-exact (integer test != 0)."""
  _need_gpu(mask)
  mask = mask.contiguous().float()
  b, _, h, w = mask.shape
  out = torch.empty(b, h, w, dtype=torch.uint8, device=mask.device)
  lib.call('csmri_mask_to_u8', mask.data_ptr(), b, h, w, out.data_ptr(), stream())
  return out


# ----------------------------------------------------------------------------
# convolution layer object (geometry + packed weights)
# ----------------------------------------------------------------------------


class ConvLayer(object):
  """Geometry and packed-weight cache of one nn.Conv2d-equivalent.

  ``weight``: fp32 tensor/Parameter [Cout,Cin,KH,KW] (reference layout, state-dict
  contract).  ``pads`` = (left, right, top, bottom) as produced by the reference's
  SAME-padding rule (models/utils.py:58-85).  ``border``: 'zero' | 'reflection'.
  """

  def __init__(self, weight, bias, stride, pads, border, dtype, upsample=False, frozen=False):
    self.weight, self.bias = weight, bias
    self.cout, self.cin, self.kh, self.kw = weight.shape
    self.stride, self.pads = stride, tuple(pads)
    self.border = BORDER_REFLECT if border == 'reflection' else BORDER_ZERO
    self.upsample = bool(upsample)
    self.dtype = dtype                      # torch dtype of activations / packed weights
    self.frozen = frozen
    self.cin_p, self.cout_p = pad8(self.cin), pad8(self.cout)
    self._packs = {}
    self._bias_pad = None
    self.train_weights = True               # False: skip wgrad (e.g. D during the G backward)
    self.fp8 = False                        # True: forward products on fp8 operands where the shape allows
    self._pack8 = None

  # -- packs ---------------------------------------------------------------
  def _pack(self, mode):
    key = mode
    ent = self._packs.get(key)
    epoch = _weight_epoch(self)
    if ent is not None and ent[0] == epoch and ent[1].device == self.weight.device:
      return ent[1], ent[2], ent[3], ent[4]
    group = getattr(self, 'group', None)
    if group is not None and not self.frozen and ent is not None and group.repack(mode):
      ent = self._packs[key]
      return ent[1], ent[2], ent[3], ent[4]
    dt = BF16 if self.dtype == torch.bfloat16 else F32
    nbytes = lib.raw('csmri_pack_weight_bytes')(mode, dt, self.cout, self.cin, self.kh, self.kw)
    buf = ent[1] if ent is not None and ent[1].device == self.weight.device else \
        torch.empty(nbytes, dtype=torch.uint8, device=self.weight.device)
    kp, cs, tw = C.c_int(0), C.c_longlong(0), C.c_int(0)
    w = self.weight.detach()
    assert w.is_contiguous() and w.dtype == torch.float32
    lib.call('csmri_pack_weight', mode, dt, w.data_ptr(), self.cout, self.cin, self.kh, self.kw,
             buf.data_ptr(), C.byref(kp), C.byref(cs), C.byref(tw), stream())
    self._packs[key] = (epoch, buf, kp.value, cs.value, tw.value)
    return buf, kp.value, cs.value, tw.value

  def fp8_ok(self, x0, x1):
    """The fp8 variant's shape rules (include/csmri_hip.h, csmri_gconv_desc.in_dequant)."""
    if not self.fp8 or self.dtype != torch.bfloat16 or self.cin_p % 128 or self.cout_p % 64:
      return False
    if not x0.is_contiguous() or (x1 is not None and (not x1.is_contiguous() or x0.shape[3] % 16)):
      return False
    return True

  def _pack_fp8(self):
    """Forward pack in fp8: the fp32 forward pack quantised with one power-of-two scale."""
    epoch = _weight_epoch(self)
    ent = self._pack8
    if ent is not None and ent[0] == epoch and ent[1].device == self.weight.device:
      return ent[1], ent[2], ent[3]
    dev = self.weight.device
    nbytes = lib.raw('csmri_pack_weight_bytes')(0, F32, self.cout, self.cin, self.kh, self.kw)
    tmp = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    kp, cs, tw = C.c_int(0), C.c_longlong(0), C.c_int(0)
    w = self.weight.detach()
    assert w.is_contiguous() and w.dtype == torch.float32
    lib.call('csmri_pack_weight', 0, F32, w.data_ptr(), self.cout, self.cin, self.kh, self.kw,
             tmp.data_ptr(), C.byref(kp), C.byref(cs), C.byref(tw), stream())
    if ent is not None and ent[1].device == dev:
      q, scales = ent[1], ent[3]
    else:
      q = torch.empty(tmp.numel(), dtype=torch.uint8, device=dev)
      scales = torch.empty(3, dtype=torch.float32, device=dev)   # amax, scale, 1/scale
    lib.call('csmri_absmax', F32, tmp.data_ptr(), tmp.numel(), scales.data_ptr(), stream())
    lib.call('csmri_quantize_fp8', F32, tmp.data_ptr(), q.data_ptr(), tmp.numel(), scales.data_ptr(),
             scales.data_ptr() + 4, stream())
    self._pack8 = (epoch, q, kp.value, scales)
    return q, kp.value, scales

  def bias_padded(self):
    if self.bias is None:
      return None
    epoch = _weight_epoch(self)
    if self._bias_pad is None or self._bias_pad[0] != epoch or \
        self._bias_pad[1].device != self.bias.device:
      group = getattr(self, 'group', None)
      if group is not None and self._bias_pad is not None and \
          self._bias_pad[1].device == self.bias.device and group.refresh_biases():
        return self._bias_pad[1]
      bp = self._bias_pad[1] if self._bias_pad is not None and \
          self._bias_pad[1].device == self.bias.device else \
          torch.zeros(max(self.cout_p, 128), dtype=torch.float32, device=self.bias.device)
      bp[:self.cout].copy_(self.bias.detach())
      self._bias_pad = (epoch, bp)
    return self._bias_pad[1]

  def out_hw(self, h, w):
    pl, pr, pt, pb = self.pads
    hv, wv = (2 * h, 2 * w) if self.upsample else (h, w)
    return ((hv + pt + pb - self.kh) // self.stride + 1,
            (wv + pl + pr - self.kw) // self.stride + 1)


class PackGroup(object):
  """The conv layers of one trainable network.  After an optimizer step every layer's packed
  weights of a mode are stale; the first layer that notices re-packs ALL of them with one
  multi-tensor launch (csmri_pack_weight_multi) instead of one launch per layer."""

  def __init__(self, layers):
    self.layers = [l for l in layers if not l.frozen]
    for l in self.layers:
      l.group = self
    self._tables = {}
    self.epoch = 0          # bumped by the optimizer that owns these weights (FlatAdam.pack_groups)

  def bump(self):
    self.epoch += 1

  def modes(self):
    return sorted(set(m for l in self.layers for m in l._packs))

  def repack_stale(self, modes=None):
    """Re-pack now (current stream) every mode whose packs are stale, instead of at first use."""
    for mode in (self.modes() if modes is None else modes):
      members = [l for l in self.layers if mode in l._packs]
      if members and any(l._packs[mode][0] != _weight_epoch(l) for l in members):
        self.repack(mode)
    if any(l._bias_pad is not None and l._bias_pad[0] != _weight_epoch(l) for l in self.layers):
      self.refresh_biases()

  def refresh_biases(self):
    """One multi-tensor copy of every stale member bias into its zero-padded fp32 buffer (the fallback: a re-pack
    launch carries the biases along, see ensure_table)."""
    members = [l for l in self.layers if l.bias is not None and l._bias_pad is not None and
               l._bias_pad[1].device == l.bias.device and l._bias_pad[0] != _weight_epoch(l)]
    if not members:
      return False
    torch._foreach_copy_([l._bias_pad[1][:l.cout] for l in members],
                         [l.bias.detach() for l in members])
    for l in members:
      l._bias_pad = (_weight_epoch(l), l._bias_pad[1])
    return True

  @staticmethod
  def _has_bias_pad(l):
    return l.bias is not None and l._bias_pad is not None and l._bias_pad[1].device == l.bias.device

  def ensure_table(self, mode):
    """The device-side item table of ``mode``'s multi-layer re-pack (built by a host-to-device copy: must exist
    before a stream capture that re-packs).  Returns ((sig, table, n, bias_layers), members) or None when no member
    owns a pack of that mode.  The item of a layer whose LOWEST pack mode this is also carries its bias refresh, and
    the table of the group's lowest mode gets one bias-only item per layer that has a bias buffer but no pack at all:
    after an optimizer step the re-pack launches are the only ones the weights need (the biases used to cost a
    multi-tensor copy launch of their own, 25 us for the U-Net's)."""
    members = [l for l in self.layers if mode in l._packs]
    if not members:
      return None
    carried = [l for l in members if PACK_BIAS and self._has_bias_pad(l) and min(l._packs) == mode]
    bias_only = []
    if PACK_BIAS and mode == min(self.modes()):
      bias_only = [l for l in self.layers if not l._packs and self._has_bias_pad(l)]
    sig = tuple((id(l), l.weight.data_ptr(), l._packs[mode][1].data_ptr()) for l in members) + \
        tuple((id(l), l.bias.data_ptr(), l._bias_pad[1].data_ptr()) for l in carried + bias_only)
    tab = self._tables.get(mode)
    if tab is None or tab[0] != sig:
      arr = (lib.PackItem * (len(members) + len(bias_only)))()
      for it, l in zip(arr, members + bias_only):
        it.mode, it.dtype = -1, (BF16 if l.dtype == torch.bfloat16 else F32)
        if l in members:
          it.w, it.out, it.mode = l.weight.data_ptr(), l._packs[mode][1].data_ptr(), mode
        it.Cout, it.Cin, it.KH, it.KW = l.cout, l.cin, l.kh, l.kw
        if l in carried or l in bias_only:
          it.bias, it.bias_out = l.bias.data_ptr(), l._bias_pad[1].data_ptr()
      host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
      tab = (sig, host.to(members[0].weight.device), len(arr), carried + bias_only)
      self._tables[mode] = tab
    return tab, members

  def repack(self, mode):
    """Re-pack every member that already owns a pack of ``mode`` (buffers exist after the
    first step).  Returns False if the table cannot be used (caller packs individually)."""
    got = self.ensure_table(mode)
    if got is None:
      return False
    tab, members = got
    lib.call('csmri_pack_weight_multi', tab[1].data_ptr(), tab[2], stream())
    for l in members:
      e = l._packs[mode]
      l._packs[mode] = (_weight_epoch(l),) + tuple(e[1:])
    for l in tab[3]:
      l._bias_pad = (_weight_epoch(l), l._bias_pad[1])
    return True


FOLD_WINDOW = True      # reflection dgrads: centre written in place + border-only halo fold
PROFILE_SHAPES = False  # tools: append the problem shape to gconv labels
PROFILE = None        # bench.py sets this to a list to collect per-launch HIP event timings
GRAD_READY_HOOK = None  # data parallelism: called with the ConvLayer once its weight-gradient launch is issued
LAUNCH_LOG = None     # tests set this to a list: (kind, kernel instance name, splitk) per conv-library launch


class _Timed(object):
  """HIP-event bracket on the current stream around one library launch."""

  def __init__(self, label, flops):
    self.rec = None
    if PROFILE is not None:
      self.rec = (label, flops, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

  def __enter__(self):
    if self.rec is not None:
      self.rec[2].record()
    return self

  def __exit__(self, *exc):
    if self.rec is not None:
      self.rec[3].record()
      PROFILE.append(self.rec)
    return False


_SEEDS = {}


def backward_scalar(total):
  """``total.backward()`` with the seed gradient taken from a per-device constant instead of autograd's ones_like
  (one fill launch per backward pass, 5-12 us on the step's critical chain)."""
  if not SEED_CONST or not total.is_cuda or total.dim() != 0:
    return total.backward()
  key = (total.device, total.dtype)
  one = _SEEDS.get(key)
  if one is None:
    one = _SEEDS[key] = torch.ones((), dtype=total.dtype, device=total.device)
  total.backward(gradient=one)


def absmax(x):
  """max |x| of a contiguous fp32 / bf16 tensor as a device scalar [1] (csmri_absmax)."""
  _need_gpu(x)
  assert x.is_contiguous()
  out = torch.empty(1, dtype=torch.float32, device=x.device)
  lib.call('csmri_absmax', dt_of(x), x.data_ptr(), x.numel(), out.data_ptr(), stream())
  return out


def quantize_fp8(x, amax=None):
  """(q, scales): q = e4m3fn(x * scale) as uint8 of x's shape, scales = device [scale, 1/scale] with the
  power-of-two scale 2^(7 - floor(log2 amax)) (csmri_quantize_fp8; amax defaults to max |x|)."""
  _need_gpu(x)
  assert x.is_contiguous() and x.numel() % 16 == 0
  if amax is None:
    amax = absmax(x)
  q = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
  scales = torch.empty(2, dtype=torch.float32, device=x.device)
  lib.call('csmri_quantize_fp8', dt_of(x), x.data_ptr(), q.data_ptr(), x.numel(), amax.data_ptr(),
           scales.data_ptr(), stream())
  return q, scales


SPLITK_OVERRIDE = None   # tools/sk_sweep.py only: force the split-K factor of csmri_gconv launches
GCONV_FLAGS = int(os.environ.get('CSMRI_GCONV_FLAGS', '0'))   # or-ed into csmri_gconv_desc.flags (A/B switches of the conv dispatch: 2 / 16 = gpipe everywhere / nowhere)


def _gconv_run(d, want_stats, flops=0.0):
  d.flags = GCONV_FLAGS
  splitk = lib.raw('csmri_gconv_suggest_splitk')(C.byref(d))
  if SPLITK_OVERRIDE is not None and splitk >= 1 and not want_stats:
    splitk = SPLITK_OVERRIDE
  if want_stats:
    splitk = 1
  d.splitk = splitk
  keep = []
  dev = torch.device('cuda', torch.cuda.current_device())
  if splitk > 1:
    nbytes = lib.raw('csmri_gconv_slab_bytes')(C.byref(d))
    slab = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    d.slab = slab.data_ptr()
    keep.append(slab)
  stats = None
  if want_stats:
    rows = lib.raw('csmri_gconv_stats_rows')(C.byref(d))
    stats = torch.empty(rows, 2, d.Cout, dtype=torch.float32, device=dev)
    d.stats_partial = stats.data_ptr()
  if LAUNCH_LOG is not None:
    nm = C.create_string_buffer(96)
    lib.call('csmri_gconv_kernel_name', C.byref(d), nm, 96)
    LAUNCH_LOG.append(('gconv', nm.value.decode(), splitk))
  d.flags = GCONV_FLAGS
  if PROFILE is None:
    lib.call('csmri_gconv', C.byref(d), stream())
    return stats
  # profiling: bracket the main kernel alone, keyed by the instance name rocprofv3 reports
  name = C.create_string_buffer(96)
  lib.call('csmri_gconv_kernel_name', C.byref(d), name, 96)
  d.flags = 1 | GCONV_FLAGS                   # CSMRI_GCONV_DEFER_REDUCE
  label = name.value.decode()
  if PROFILE_SHAPES:
    label += ' B%d %dx%d Cin%d Cout%d t%dx%d s%d ncls%d sk%d' % (
        d.B, d.Ho, d.Wo, d.Cin, d.Cout, d.TH, d.TW, d.in_s, max(1, d.nclass), splitk)
  with _Timed(label, flops):
    lib.call('csmri_gconv', C.byref(d), stream())
  if splitk > 1:
    with _Timed('gconv_reduce_kernel', 0.0):
      lib.call('csmri_gconv_reduce', C.byref(d), stream())
  return stats


def conv_forward(layer, x0, x1=None, use_bias=True, act_slope=1.0, want_stats=False,
                 out_dtype=None):
  """y = act(conv(pad(up(cat(x0,x1)))) + bias).  Returns (y, stats_partial|None)."""
  _need_gpu(x0)
  x0 = as_nhwc(x0)
  b, h, w, c0 = x0.shape
  cin = c0
  if x1 is not None:
    x1 = as_nhwc(x1)
    assert x1.shape[:3] == x0.shape[:3] and x1.dtype == x0.dtype
    cin += x1.shape[3]
  assert cin == layer.cin_p, (cin, layer.cin_p)
  assert x0.dtype == layer.dtype
  ho, wo = layer.out_hw(h, w)
  odt = out_dtype or layer.dtype
  y = torch.empty(b, ho, wo, layer.cout_p, dtype=odt, device=x0.device)
  pl, pr, pt, pb = layer.pads
  d = lib.GConvDesc()
  keep8 = None
  if layer.fp8_ok(x0, x1):
    # fp8 variant: both operands rounded to e4m3fn with per-tensor power-of-two scales (a concatenated
    # input shares one scale: the larger of the two maxima); accumulate / epilogue / output unchanged
    wp, kp, wsc = layer._pack_fp8()
    tw = layer.kw
    amax = absmax(x0)
    if x1 is not None:
      amax = torch.maximum(amax, absmax(x1))
    q0, xsc = quantize_fp8(x0, amax)
    q1 = quantize_fp8(x1, amax)[0] if x1 is not None else None
    keep8 = (q0, q1, xsc, wsc)
    d.dtype, d.out_dtype = FP8, dt_of(y)
    d.in0, d.in0_pix_stride = q0.data_ptr(), q0.stride(2)
    if q1 is not None:
      d.in1, d.in1_pix_stride, d.c0 = q1.data_ptr(), q1.stride(2), c0
    d.in_dequant, d.w_dequant = xsc.data_ptr() + 4, wsc.data_ptr() + 8
  else:
    wp, kp, _, tw = layer._pack(0)
    d.dtype, d.out_dtype = dt_of(x0), dt_of(y)
    d.in0, d.in0_pix_stride = x0.data_ptr(), x0.stride(2)
    if x1 is not None:
      d.in1, d.in1_pix_stride, d.c0 = x1.data_ptr(), x1.stride(2), c0
  d.B, d.Hin, d.Win, d.Cin = b, h, w, cin
  d.upsample, d.border = int(layer.upsample), layer.border
  d.TH, d.TW, d.in_s = layer.kh, tw, layer.stride     # tw >= kw: zero taps pad few-channel filters
  d.dy0, d.dy_step, d.dx0, d.dx_step = -pt, 1, -pl, 1
  d.w, d.Kp, d.nclass, d.w_class_stride = wp.data_ptr(), kp, 1, 0
  d.out, d.out_pix_stride, d.Hout_t, d.Wout_t = y.data_ptr(), y.stride(2), ho, wo
  d.Ho, d.Wo, d.out_sy, d.out_sx, d.out_oy, d.out_ox = ho, wo, 1, 1, 0, 0
  d.Cout = layer.cout_p
  d.cin_real, d.cout_real = (layer.cin if x1 is None else 0), layer.cout
  bias = layer.bias_padded() if use_bias else None
  d.bias = ptr(bias)
  d.act_slope = float(act_slope)
  stats = _gconv_run(d, want_stats, 2.0 * b * ho * wo * layer.cout * layer.cin * layer.kh * layer.kw)
  return y, stats


def conv_dgrad(layer, gy, in_hw, g_src=None, g_slope=1.0):
  """Gradient w.r.t. the (concatenated) conv input.  gy: [B,Ho,Wo,Cout_p].
  Returns [B,H,W,Cin_p].  Optionally multiplies by lrelu'(g_src) (producer's
  activation) on the way out."""
  gy = as_nhwc(gy)
  b, ho, wo, cg = gy.shape
  assert cg == layer.cout_p and gy.dtype == layer.dtype
  h, w = in_hw
  pl, pr, pt, pb = layer.pads
  dev = gy.device
  d = lib.GConvDesc()
  d.dtype = d.out_dtype = dt_of(gy)
  d.in0, d.in0_pix_stride = gy.data_ptr(), gy.stride(2)
  d.B, d.Hin, d.Win, d.Cin = b, ho, wo, layer.cout_p
  d.upsample, d.border = 0, BORDER_ZERO
  d.in_s = 1
  d.Cout = layer.cin_p
  d.cin_real, d.cout_real = layer.cout, layer.cin         # (roles swapped: the K side is the conv's output channels)
  d.act_slope = 1.0
  direct = layer.border == BORDER_ZERO and not layer.upsample
  if layer.stride == 1:
    # flipped-tap pack: the input-gradient is a plain correlation over dY
    wp, kp, _, tw = layer._pack(3)
    d.w, d.Kp, d.nclass, d.w_class_stride = wp.data_ptr(), kp, 1, 0
    d.TH, d.TW = layer.kh, tw
    d.dy_step = d.dx_step = 1
    if direct:
      out = torch.empty(b, h, w, layer.cin_p, dtype=gy.dtype, device=dev)
      d.dy0, d.dx0 = pt - (layer.kh - 1), pl - (layer.kw - 1)
      d.Ho, d.Wo, d.Hout_t, d.Wout_t = h, w, h, w
      if g_src is not None:
        g_src = as_nhwc(g_src)
        d.g_src, d.g_pix_stride, d.g_slope, d.g_dtype = g_src.data_ptr(), g_src.stride(2), g_slope, dt_of(g_src)
    else:
      hv, wv = (2 * h, 2 * w) if layer.upsample else (h, w)
      hp, wpad = hv + pt + pb, wv + pl + pr
      out = torch.empty(b, hp, wpad, layer.cin_p, dtype=gy.dtype, device=dev)
      d.dy0, d.dx0 = -(layer.kh - 1), -(layer.kw - 1)
      d.Ho, d.Wo, d.Hout_t, d.Wout_t = hp, wpad, hp, wpad
    d.out_sy = d.out_sx = 1
    d.out_oy = d.out_ox = 0
  elif layer.stride == 2:
    if direct or layer.upsample or layer.kh % 2 or layer.kw % 2:
      raise RuntimeError('csmri_hip: stride-2 dgrad implemented for even kernels with reflection padding')
    wp, kp, cs, _ = layer._pack(2)
    d.w, d.Kp, d.nclass, d.w_class_stride = wp.data_ptr(), kp, 4, cs
    d.TH, d.TW = layer.kh // 2, layer.kw // 2
    d.dy_step = d.dx_step = -1
    d.dy0 = d.dx0 = 0
    hp, wpad = h + pt + pb, w + pl + pr
    assert hp % 2 == 0 and wpad % 2 == 0
    out = torch.empty(b, hp, wpad, layer.cin_p, dtype=gy.dtype, device=dev)
    d.Ho, d.Wo, d.Hout_t, d.Wout_t = hp // 2, wpad // 2, hp, wpad
    d.out_sy = d.out_sx = 2
    d.out_oy = d.out_ox = 0
  else:
    raise RuntimeError('unsupported stride')
  flops = 2.0 * b * ho * wo * layer.cout * layer.cin * layer.kh * layer.kw
  gs = as_nhwc(g_src) if g_src is not None else None
  if not direct and not layer.upsample and h >= pt + pb + 2 and w >= pl + pr + 2 and FOLD_WINDOW:
    # reflection padding, no upsampling: the kernel writes the un-padded centre straight into dx
    # and only the halo positions into `out`; a border-only kernel mirrors the halo back
    dx = torch.empty(b, h, w, layer.cin_p, dtype=gy.dtype, device=dev)
    d.out, d.out_pix_stride = dx.data_ptr(), dx.stride(2)
    d.out_halo, d.halo_pix_stride = out.data_ptr(), out.stride(2)
    d.win_y0, d.win_x0, d.win_h, d.win_w = pt, pl, h, w
    if gs is not None:
      d.g_src, d.g_pix_stride, d.g_slope, d.g_dtype = gs.data_ptr(), gs.stride(2), g_slope, dt_of(gs)
    _gconv_run(d, False, flops)
    lib.call('csmri_fold_halo', dt_of(out), out.data_ptr(), dx.data_ptr(), dx.stride(2), b, h, w,
             layer.cin_p, pt, pb, pl, pr, ptr(gs), gs.stride(2) if gs is not None else 0,
             float(g_slope), stream())
    return dx
  d.out, d.out_pix_stride = out.data_ptr(), out.stride(2)
  _gconv_run(d, False, flops)
  if direct:
    return out
  dx = torch.empty(b, h, w, layer.cin_p, dtype=gy.dtype, device=dev)
  lib.call('csmri_fold_pad_grad', dt_of(out), out.data_ptr(), dx.data_ptr(), dx.stride(2), b, h, w,
           layer.cin_p, pt, pb, pl, pr, int(layer.upsample), ptr(gs),
           gs.stride(2) if gs is not None else 0, float(g_slope), stream())
  return dx


def conv_wgrad(layer, x0, x1, gy, accumulate=True):
  """Accumulates dW (and db) into layer.weight.grad / layer.bias.grad (fp32)."""
  x0 = as_nhwc(x0)
  gy = as_nhwc(gy)
  b, h, w, c0 = x0.shape
  _, ho, wo, _ = gy.shape
  wgt = layer.weight
  if wgt.grad is None:
    wgt.grad = torch.zeros_like(wgt)
  if layer.bias is not None and layer.bias.grad is None:
    layer.bias.grad = torch.zeros_like(layer.bias)
  pl, pr, pt, pb = layer.pads
  d = lib.WGradDesc()
  d.dtype = dt_of(x0)
  d.in0, d.in0_pix_stride = x0.data_ptr(), x0.stride(2)
  if x1 is not None:
    x1 = as_nhwc(x1)
    d.in1, d.in1_pix_stride, d.c0 = x1.data_ptr(), x1.stride(2), c0
  d.B, d.Hin, d.Win, d.Cin = b, h, w, layer.cin_p
  d.upsample, d.border = int(layer.upsample), layer.border
  d.KH, d.KW, d.stride, d.pad_t, d.pad_l = layer.kh, layer.kw, layer.stride, pt, pl
  d.dy, d.dy_pix_stride, d.Ho, d.Wo, d.Cout = gy.data_ptr(), gy.stride(2), ho, wo, layer.cout_p
  d.Cin_real, d.Cout_real = layer.cin, layer.cout
  assert wgt.grad.is_contiguous()
  d.dw = wgt.grad.data_ptr()
  d.db = layer.bias.grad.data_ptr() if layer.bias is not None else 0
  # lazily zeroed gradients (FlatAdam.lazy_zero): the first write after zero_grad() overwrites
  fresh = getattr(wgt, '_grad_fresh', False)
  wgt._kernel_grad = True                 # (FlatAdam.lazy_zero: these gradients are written here, not by autograd)
  if layer.bias is not None:
    layer.bias._kernel_grad = True
  if fresh:
    wgt._grad_fresh = False
    if layer.bias is not None:
      layer.bias._grad_fresh = False
  d.accumulate = int(accumulate and not fresh)
  d.splitk = lib.raw('csmri_wgrad_suggest_splitk')(C.byref(d))
  nbytes = lib.raw('csmri_wgrad_slab_bytes')(C.byref(d))

  kname = None
  if LAUNCH_LOG is not None or PROFILE is not None:
    nm = C.create_string_buffer(96)
    lib.call('csmri_wgrad_kernel_name', C.byref(d), nm, 96)
    kname = nm.value.decode()
  if LAUNCH_LOG is not None:
    LAUNCH_LOG.append(('wgrad', kname, d.splitk))

  def launch():
    slab = torch.empty(nbytes // 4, dtype=torch.float32, device=x0.device)
    d.slab = slab.data_ptr()
    label = kname or 'wgrad'            # the instance name rocprofv3 reports (bench.py keys its table by it)
    if PROFILE_SHAPES:
      label += ' B%d %dx%d->%dx%d Cin%d Cout%d k%d s%d up%d refl%d sk%d' % (
          b, h, w, ho, wo, layer.cin_p, layer.cout_p, layer.kh, layer.stride, int(layer.upsample),
          int(layer.border == BORDER_REFLECT), d.splitk)
    # the slab reduction of every layer of this backward pass in ONE launch at its end (csmri_wgrad_finish_multi)
    # instead of one or two small launches behind each main kernel -- when somebody will flush the queue (the
    # end-of-backward callback) and nobody needs this gradient earlier (data parallelism starts a bucket per layer)
    fq = _WGRAD['finish']
    want = WGRAD_FINISH_MULTI == '1' or (WGRAD_FINISH_MULTI == 'auto' and _WGRAD['stream'] is None)
    defer = want and PROFILE is None and GRAD_READY_HOOK is None and _ensure_flush_callback()
    if defer and any(e[0].dw == d.dw for e in fq):
      _finish_wgrads()                     # a second weight gradient of the same layer: its accumulation comes after
    d.defer_finish = int(bool(defer))
    with _Timed(label, 2.0 * b * ho * wo * layer.cout * layer.cin * layer.kh * layer.kw):
      lib.call('csmri_wgrad', C.byref(d), stream())
    if defer:
      fq.append((d, slab, wgt.grad, layer.bias.grad if layer.bias is not None else None,
                 torch.cuda.current_stream()))

  if _WGRAD['stream'] is None:
    launch()
    return
  # weight gradients are only consumed by the optimizer: run them on a side stream next to the
  # data-gradient chain (a layer always uses the same side stream, so its accumulation stays race-free)
  side = _side_stream_of(layer)
  if WGRAD_DEFER <= 0:
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      launch()
    for t in (x0, x1, gy):
      if t is not None:
        t.record_stream(side)
    _mark_pending(side)
    return
  # Deferred issue: the operands are ready NOW (event), but the launch itself is issued one weight gradient later.
  # In a captured graph a node's first-issued successor stays on its hardware queue; issuing the side-stream launch
  # right here made it the first successor of the node the backward's critical chain continues from, and that chain
  # paid a cross-queue hand-off (10-15 us in the kernel trace) at every layer.
  ev = torch.cuda.Event()
  ev.record(torch.cuda.current_stream())
  q = _WGRAD['deferred']
  # whatever is still held back when this backward pass ends is issued then (autograd engine callback): a caller
  # that reads .grad after backward() + synchronize / join_wgrad_stream() sees every launch, as without deferral
  if not _ensure_flush_callback():      # not inside a backward pass (direct call): no deferral
    _issue_deferred_wgrad((ev, launch, (x0, x1, gy), layer))
    return
  q.append((ev, launch, (x0, x1, gy), layer))
  # (Releasing the held-back launches every N layers as the first successor of the chain's newest node, and two or
  # three weight-gradient streams, were measured slower the more they overlapped with the chain: DESIGN 9.0.)
  while len(q) > WGRAD_DEFER:
    _issue_deferred_wgrad(q.pop(0))


def _wgrad_deferred_mode():
  return WGRAD_DEFER > 0 and _WGRAD['stream'] is not None


_NAMED_STREAMS = {}


def named_stream(name, priority=0):
  """One HIP stream per role and device for the whole process.  torch hands out streams from a pool of 32 per
  priority, round-robin: runners that each create their own half-dozen side streams (tests build dozens of runners in
  one process) end up with two roles ALIASED to one stream, and a graph captured across such a pair crashed the
  runtime at replay.  Roles: 'wgrad', 'wgrad1'.., 'vgg', 'third', 'lookahead', 'metrics', 'capture', 'warmup'."""
  key = (name, torch.cuda.current_device(), priority)
  st = _NAMED_STREAMS.get(key)
  if st is None:
    st = _NAMED_STREAMS[key] = torch.cuda.Stream(priority=priority)
  return st


def _mark_pending(side):
  if not any(st is side for st in _WGRAD['pending']):
    _WGRAD['pending'].append(side)


def _side_stream_of(layer):
  """The side stream of the weight-gradient launches: ONE for all layers, so a layer's accumulations stay ordered."""
  return _WGRAD['stream']


def _issue_deferred_wgrad(d):
  ev, launch, tensors, layer = d
  side = _side_stream_of(layer)
  side.wait_event(ev)
  with torch.cuda.stream(side):
    launch()
  for t in tensors:
    if t is not None:
      t.record_stream(side)
  _mark_pending(side)
  if GRAD_READY_HOOK is not None:
    GRAD_READY_HOOK(layer)


def _ensure_flush_callback():
  """Register _flush_deferred_wgrad to run when the current backward pass ends.  False outside a backward pass:
  nothing would flush, the caller must not hold anything back."""
  # Registered on EVERY call (the callback is idempotent and costs a Python call): a "registered for this pass" flag
  # would survive a backward pass that raised before its callbacks ran and silently disable the flush of the next one.
  try:
    torch.autograd.Variable._execution_engine.queue_callback(_flush_deferred_wgrad)
  except RuntimeError:
    return False
  return True


def _finish_wgrads():
  """The deferred slab reductions of the weight gradients issued so far, one launch (on the stream they ran on)."""
  fq = _WGRAD['finish']
  if not fq:
    return
  st = fq[0][4]
  arr = (lib.WGradDesc * len(fq))()
  for i, e in enumerate(fq):
    assert e[4] == st
    C.memmove(C.byref(arr[i]), C.byref(e[0]), C.sizeof(lib.WGradDesc))
  with torch.cuda.stream(st):
    lib.call('csmri_wgrad_finish_multi', arr, len(fq), st.cuda_stream)
  del fq[:]


def _flush_deferred_wgrad():
  q = _WGRAD['deferred']
  while q:
    _issue_deferred_wgrad(q.pop(0))
  _finish_wgrads()


WGRAD_DEFER = 2       # weight-gradient launches held back behind the data-gradient chain (0 = none); +7 % on C3
# one slab-reduction launch per backward pass: 'auto' = when the weight gradients run on the main stream (RecNet MSE
# step: 30 launches of 12-16 us in a strictly serial step, +9.5 %); with the side stream of the GAN step the per-layer
# reductions read their slab while it is still in the Infinity Cache and the single launch measured 1 % slower
WGRAD_FINISH_MULTI = 'auto'       # auto | '1' | '0' (tests set it)
_WGRAD = {'stream': None, 'pending': [], 'deferred': [], 'finish': []}


def enable_wgrad_stream(on):
  """Route csmri_wgrad launches to a dedicated side stream (join with join_wgrad_stream)."""
  if on and _WGRAD['stream'] is None:
    _WGRAD['stream'] = named_stream('wgrad')
  elif not on:
    join_wgrad_stream()
    _WGRAD['stream'] = None


def join_wgrad_stream():
  """Make the current stream wait for every weight-gradient launch issued so far."""
  _flush_deferred_wgrad()
  # only streams that received launches since the last join: waiting on an idle stream from inside a graph capture
  # records an event outside the capture (the replay of such a graph crashed in the runtime)
  for st in _WGRAD['pending']:
    torch.cuda.current_stream().wait_stream(st)
  _WGRAD['pending'] = []


def act_bwd(gz, z, slope, gz2=None):
  """dy = (gz + gz2) * lrelu'(z); ``gz2``: a second gradient of z (see _two_grads)."""
  gz, z = as_nhwc(gz), as_nhwc(z)
  b, h, w, c = z.shape
  out = torch.empty(b, h, w, c, dtype=z.dtype, device=z.device)
  lib.call('csmri_act_bwd', dt_of(z), gz.data_ptr(), gz.stride(2), z.data_ptr(), z.stride(2),
           out.data_ptr(), out.stride(2), b * h * w, c, float(slope), ptr(gz2),
           gz2.stride(2) if gz2 is not None else 0, stream())
  return out


def _two_grads(ctx, g_a, g_b, dtype):
  """The gradients of the two aliases a ``tap`` replay node returned (either may be missing) as (g, g2) in the layer's
  dtype: g2 is summed with g inside the first backward kernel (csmri_bn_bwd_* / csmri_act_bwd ``dz2``), not by an add
  launch of its own -- on the generator step every discriminator feature has two consumers, the next layer and the
  feature-matching loss, and that add sat on the critical chain of the backward six times."""
  if g_a is None:
    g_a, g_b = g_b, None
  g_a = as_nhwc(g_a)
  if g_a.dtype != dtype:
    g_a = g_a.to(dtype)
  if g_b is not None:
    g_b = as_nhwc(g_b)
    if g_b.dtype != dtype:
      g_b = g_b.to(dtype)
  return g_a, g_b


class ConvAct(torch.autograd.Function):
  """pad -> conv -> (+bias) -> LeakyReLU/ReLU, one kernel.  inputs: x0, x1|None,
  weight, bias (the Parameters, so autograd schedules the backward)."""

  @staticmethod
  def forward(ctx, x0, x1, weight, bias, layer, act_slope, out_dtype):
    y, _ = conv_forward(layer, x0, x1, True, act_slope, False, out_dtype)
    ctx.layer, ctx.act_slope = layer, act_slope
    ctx.c0 = x0.shape[3]
    ctx.in_hw = (x0.shape[1], x0.shape[2])
    ctx.save_for_backward(x0, x1, y)
    # captured at forward time: a later pass may toggle layer.train_weights
    ctx.w_req = weight.requires_grad and layer.train_weights
    return y

  @staticmethod
  def backward(ctx, gy, gy2=None):
    layer = ctx.layer
    x0, x1, y = ctx.saved_tensors
    gy, gy2 = _two_grads(ctx, gy, gy2, layer.dtype)
    if ctx.act_slope != 1.0:
      g = act_bwd(gy, y, ctx.act_slope, gy2)
    else:
      g = gy if gy2 is None else gy + gy2
    gx0 = gx1 = None                  # data gradient first, weight gradient second (see ConvBnAct._finish_backward)
    if ctx.needs_input_grad[0] or (x1 is not None and ctx.needs_input_grad[1]):
      gx = conv_dgrad(layer, g, ctx.in_hw)
      if x1 is None:
        gx0 = gx
      else:
        gx0, gx1 = gx[..., :ctx.c0], gx[..., ctx.c0:]
    if ctx.w_req:
      conv_wgrad(layer, x0, x1, g)
      if GRAD_READY_HOOK is not None and not _wgrad_deferred_mode():
        GRAD_READY_HOOK(layer)
    return gx0, gx1, None, None, None, None, None


class ConvActReplay(torch.autograd.Function):
  """A sub-batch of a ConvAct call that has ALREADY run on a larger stacked batch (the discriminator's grouped pass):
  forward launches nothing and returns the sub-batch's rows of the stored output; backward is ConvAct's own on those
  rows.  ``w_req``: whether this sub-batch contributes weight gradients."""

  @staticmethod
  def forward(ctx, x0, x1, weight, bias, layer, act_slope, y_holder, w_req, tap=False, nchw_out=0):
    y_rows = y_holder[0]               # (in a list: a tensor argument returned as it is would be aliased to the input)
    ctx.layer, ctx.act_slope = layer, act_slope
    ctx.c0 = x0.shape[3]
    ctx.in_hw = (x0.shape[1], x0.shape[2])
    ctx.save_for_backward(x0, x1, y_rows)
    ctx.w_req = bool(w_req) and weight.requires_grad
    ctx.nchw_out = nchw_out
    if nchw_out:
      # the output in both of the forms it leaves the library in -- the device-layout rows and the fp32 NCHW API
      # tensor of ``nchw_out`` channels (ToNCHW's result) --: the backward converts, sums and rounds the two
      # gradients in one launch (csmri_nchw_to_nhwc_add) where ToNCHW's backward, autograd's add and a dtype cast
      # were three
      assert not tap and act_slope == 1.0
      ctx.set_materialize_grads(False)
      return y_rows[:], nhwc_to_nchw(y_rows, nchw_out)
    if not tap:
      return y_rows
    # ``tap``: two aliases of the output, one per consumer, so that the backward receives their gradients
    # separately (_two_grads) instead of autograd's sum
    ctx.set_materialize_grads(False)
    return y_rows[:], y_rows[:]

  @staticmethod
  def backward(ctx, gy, gy2=None):
    if ctx.nchw_out and gy2 is not None:
      layer = ctx.layer
      g_nchw = gy2.contiguous().float()
      b, c, h, w = g_nchw.shape
      cp = ctx.saved_tensors[2].shape[3]
      add = as_nhwc(gy) if gy is not None else None
      g = torch.empty(b, h, w, cp, dtype=layer.dtype, device=g_nchw.device)
      lib.call('csmri_nchw_to_nhwc_add', g_nchw.data_ptr(), b, c, h, w, g.data_ptr(), dt_of(g), cp, cp, ptr(add),
               dt_of(add) if add is not None else 0, add.stride(2) if add is not None else 0, stream())
      gy, gy2 = g, None
    return ConvAct.backward(ctx, gy, gy2)[:4] + (None,) * 6


# RecNet conv blocks of the supported shape run as one launch (csmri_convblock_fused_fwd); tests turn it off for A/B
FUSED_CONVBLOCK = True


def convblock_fused_forward(x, plan, out_dtype_last, need_acts, out_complex=False, x_split=False):
  """The whole conv block in one launch (csrc/convblock.hip) when its shape is the supported one, else None.
  Returns [x, a1, a2, y] (a1 / a2 = None when ``need_acts`` is false: nothing of them reaches HBM).
  ``out_complex``: y is the dense interleaved complex fp32 image [B,H,W,2] instead of the padded [B,H,W,8]."""
  if len(plan) != 3 or x.dtype != torch.bfloat16 or not is_nhwc(x):
    return None
  layers = [l for l, _ in plan]
  slopes = [s for _, s in plan]
  if slopes[0] != slopes[1] or slopes[2] != 1.0 or any(l.bias is None or l.fp8 for l in layers):
    return None
  d = lib.ConvBlockDesc()
  d.dtype = BF16
  l0 = layers[0]
  d.num_convs, d.num_filters, d.kernel_size = 3, l0.cout, l0.kh
  d.num_inputs, d.num_outputs, d.border = l0.cin, layers[2].cout, l0.border
  d.slope = float(slopes[0])
  chain_ok = all(l.kh == l.kw == l0.kh and l.stride == 1 and not l.upsample and l.border == l0.border and
                 l.dtype == torch.bfloat16 and tuple(l.pads) == (1, 1, 1, 1) for l in layers) and \
      layers[1].cin == layers[1].cout == l0.cout and layers[2].cin == l0.cout and x.shape[3] == l0.cin_p
  if not chain_ok or not lib.raw('csmri_convblock_fused_supported')(C.byref(d)):
    return None
  _need_gpu(x)
  b, h, w, _ = x.shape
  keep = []
  for i, l in enumerate(layers):
    wp, kp, _, _ = l._pack(0)
    bp = l.bias_padded()
    d.w[i], d.Kp[i], d.bias[i] = wp.data_ptr(), kp, bp.data_ptr()
    keep += [wp, bp]
  d.x, d.x_pix_stride, d.B, d.H, d.W = x.data_ptr(), x.stride(2), b, h, w
  d.slope = float(slopes[0])
  d.x_split = int(bool(x_split))              # x is CSMRI_BF16_SPLIT: layer 1 multiplies hi + lo
  a1 = a2 = None
  if need_acts:
    a1 = torch.empty(b, h, w, layers[0].cout_p, dtype=torch.bfloat16, device=x.device)
    a2 = torch.empty(b, h, w, layers[1].cout_p, dtype=torch.bfloat16, device=x.device)
    d.act[0], d.act[1] = a1.data_ptr(), a2.data_ptr()
    d.act_pix_stride[0], d.act_pix_stride[1] = a1.stride(2), a2.stride(2)
  if out_complex:
    assert layers[2].cout == 2 and out_dtype_last == torch.float32
    y = torch.empty(b, h, w, 2, dtype=torch.float32, device=x.device)
  else:
    y = torch.empty(b, h, w, layers[2].cout_p, dtype=out_dtype_last or torch.bfloat16, device=x.device)
  d.out, d.out_dtype, d.out_pix_stride = y.data_ptr(), dt_of(y), y.stride(2)
  if LAUNCH_LOG is not None:
    LAUNCH_LOG.append(('convblock', 'convblock_fwd_kernel<%s>' % ('true' if need_acts else 'false'), 1))
  flops = 2.0 * b * h * w * 9 * (l0.cin * l0.cout + l0.cout * l0.cout + l0.cout * layers[2].cout)
  with _Timed('convblock_fwd_kernel', flops):
    lib.call('csmri_convblock_fused_fwd', C.byref(d), stream())
  return [x, a1, a2, y]


FUSED_CONVBLOCK_BWD = True        # tests turn it off for A/B against the per-layer backward


def _grad_targets(layer):
  """(dw, db, accumulate) of a layer's weight / bias gradient, honouring FlatAdam.lazy_zero like conv_wgrad."""
  wgt = layer.weight
  if wgt.grad is None:
    wgt.grad = torch.zeros_like(wgt)
  if layer.bias is not None and layer.bias.grad is None:
    layer.bias.grad = torch.zeros_like(layer.bias)
  fresh = getattr(wgt, '_grad_fresh', False)
  wgt._kernel_grad = True
  if layer.bias is not None:
    layer.bias._kernel_grad = True
  if fresh:
    wgt._grad_fresh = False
    if layer.bias is not None:
      layer.bias._grad_fresh = False
  assert wgt.grad.is_contiguous()
  return wgt.grad, (layer.bias.grad if layer.bias is not None else None), int(not fresh)


def convblock_fused_backward(plan, saved, g, need_dx, out_complex, x_split=False):
  """The whole backward of a fused conv block in one launch (csrc/convblock_bwd.hip) + the slab reduction of its three
  weight gradients; returns (ok, dx).  ``saved`` = [x, a1, a2, _]; ``g``: gradient of the block output."""
  if not FUSED_CONVBLOCK_BWD or len(plan) != 3:
    return False, None
  layers = [l for l, _ in plan]
  slopes = [s for _, s in plan]
  x, a1, a2 = saved[0], saved[1], saved[2]
  l0 = layers[0]
  if a1 is None or a2 is None or x.dtype != torch.bfloat16 or a1.dtype != torch.bfloat16 or a2.dtype != torch.bfloat16:
    return False, None
  if slopes[0] != slopes[1] or slopes[2] != 1.0 or any(l.bias is None or l.fp8 or l.dtype != torch.bfloat16 for l in layers):
    return False, None
  if not all(l.weight.requires_grad and l.train_weights for l in layers):
    return False, None
  ok = all(l.kh == l.kw == 3 and l.stride == 1 and not l.upsample and l.border == BORDER_ZERO and
           tuple(l.pads) == (1, 1, 1, 1) for l in layers) and \
      (l0.cin, l0.cout, layers[1].cin, layers[1].cout, layers[2].cin, layers[2].cout) == (2, 32, 32, 32, 32, 2)
  if not ok or not (is_nhwc(x) and is_nhwc(a1) and is_nhwc(a2)) or not (0.0 <= slopes[0] <= 1.0):
    return False, None
  g = as_nhwc(g)
  b, h, w, _ = x.shape
  if out_complex:
    if g.dtype != torch.float32 or g.shape[3] != 2 or not g.is_contiguous():
      return False, None
  elif g.shape[3] < 8 or g.dtype not in (torch.float32, torch.bfloat16):
    return False, None
  d = lib.ConvBlockBwdDesc()
  d.dtype = BF16
  d.num_convs, d.num_filters, d.kernel_size, d.num_inputs, d.num_outputs, d.border = 3, 32, 3, 2, 2, BORDER_ZERO
  d.x, d.x_pix_stride, d.B, d.H, d.W = x.data_ptr(), x.stride(2), b, h, w
  d.act[0], d.act[1] = a1.data_ptr(), a2.data_ptr()
  d.act_pix_stride[0], d.act_pix_stride[1] = a1.stride(2), a2.stride(2)
  d.gy, d.gy_dtype, d.gy_pix_stride = g.data_ptr(), dt_of(g), g.stride(2)
  keep = [g]
  for i, l in enumerate(layers):
    wp, kp, _, _ = l._pack(3)
    d.wd[i], d.Kp[i] = wp.data_ptr(), kp
    keep.append(wp)
  d.slope = float(slopes[0])
  dx = None
  if need_dx:
    dx = torch.empty(b, h, w, l0.cin_p, dtype=torch.bfloat16, device=x.device)
    d.dx, d.dx_pix_stride = dx.data_ptr(), dx.stride(2)
  z = lib.raw('csmri_convblock_fused_bwd_splits')(b, h, w)
  d.splits, d.want_db = z, 1
  # split images on both sides of the block: the weight gradient of layer 1 sees hi + lo of x, and dX leaves with the
  # part its bf16 rounding drops in channels 2,3 (the DC adjoint adds the two)
  d.x_split = d.dx_split = int(bool(x_split))
  descs = []
  for i, l in enumerate(layers):
    wd = lib.WGradDesc()
    wd.dtype, wd.B, wd.Hin, wd.Win, wd.Cin = BF16, b, h, w, l.cin_p
    wd.KH, wd.KW, wd.stride, wd.pad_t, wd.pad_l = 3, 3, 1, 1, 1
    wd.Ho, wd.Wo, wd.Cout, wd.Cin_real, wd.Cout_real = h, w, l.cout_p, l.cin, l.cout
    wd.splitk, wd.defer_finish = z, 2
    dw, db, acc = _grad_targets(l)
    wd.dw, wd.db, wd.accumulate = dw.data_ptr(), db.data_ptr(), acc
    slab = torch.empty(lib.raw('csmri_wgrad_slab_bytes')(C.byref(wd)) // 4, dtype=torch.float32, device=x.device)
    wd.slab = slab.data_ptr()
    wd.in0, wd.dy = x.data_ptr(), g.data_ptr()          # (only their presence is checked by the reduction)
    d.slab[i] = slab.data_ptr()
    descs.append((wd, slab, dw, db))
  if LAUNCH_LOG is not None:
    LAUNCH_LOG.append(('convblock', 'convblock_bwd_kernel', z))
  # algorithmic FLOPs of the launch: the three weight gradients + the data gradients of layers 3 and 2 (+ layer 1's
  # when dX is wanted), 2 per MAC
  per_pix = 9.0 * (l0.cin * l0.cout + layers[1].cin * layers[1].cout + layers[2].cin * layers[2].cout)
  flops = 2.0 * b * h * w * (2.0 * per_pix - (0.0 if need_dx else 9.0 * l0.cin * l0.cout))
  with _Timed('convblock_bwd_kernel', flops):
    lib.call('csmri_convblock_fused_bwd', C.byref(d), stream())
  fq = _WGRAD['finish']
  want = WGRAD_FINISH_MULTI == '1' or (WGRAD_FINISH_MULTI == 'auto' and _WGRAD['stream'] is None)
  if want and GRAD_READY_HOOK is None and _ensure_flush_callback():
    if any(e[0].dw == wd.dw for e in fq for wd, _, _, _ in descs):
      _finish_wgrads()
    for wd, slab, dw, db in descs:
      fq.append((wd, slab, dw, db, torch.cuda.current_stream()))
  else:
    arr = (lib.WGradDesc * 3)()
    for i, (wd, _, _, _) in enumerate(descs):
      C.memmove(C.byref(arr[i]), C.byref(wd), C.sizeof(lib.WGradDesc))
    lib.call('csmri_wgrad_finish_multi', arr, 3, stream())
    if GRAD_READY_HOOK is not None:
      for l in layers:
        GRAD_READY_HOOK(l)
  return True, dx


class ConvActStack(torch.autograd.Function):
  """A chain of (pad -> conv -> +bias -> LeakyReLU) layers as ONE autograd node (RecNet's ConvBlock,
  reference models/recnet.py:29-62).  Same kernels as ConvAct per layer; in the backward the activation
  derivative of layer i is applied in the epilogue of layer i+1's data-gradient kernel (its input IS layer
  i's activated output, whose sign is the pre-activation's), so no separate act_bwd pass reads and
  writes the gradient tensor between two layers.

  ``plan``: list of (ConvLayer, slope); ``params``: the layers' weight / bias Parameters (only so that
  autograd schedules this node; their gradients are accumulated into .grad by the kernels)."""

  @staticmethod
  def forward(ctx, x, plan, out_dtype_last, *params):
    # out_dtype_last = ('complex', torch.float32): a block that ends in 2 channels (re, im) without activation
    # returns the dense interleaved complex fp32 image [B,H,W,2] -- what the DC layer reads and what its adjoint
    # hands back -- instead of the channel-padded [B,H,W,8] (4x the bytes written, read, and returned as gradient)
    n = len(plan)
    out_complex = isinstance(out_dtype_last, tuple)
    x_split = False
    if out_complex:
      # ('complex', torch.float32[, 'split']): 'split' = x is a CSMRI_BF16_SPLIT image (the fused kernels multiply
      # hi + lo and return dX split as well; the per-layer fallback reads plain bf16: its weights are zero on the lo
      # channels)
      x_split = len(out_dtype_last) > 2 and out_dtype_last[2] == 'split'
      out_dtype_last = out_dtype_last[1]
      assert out_dtype_last == torch.float32 and plan[-1][0].cout == 2 and plan[-1][1] == 1.0
    ctx.x_split = x_split
    saved = None
    if FUSED_CONVBLOCK:
      need_acts = any(ctx.needs_input_grad[i] for i in (0,) + tuple(range(3, 3 + len(params))))
      saved = convblock_fused_forward(x, plan, out_dtype_last, need_acts, out_complex, x_split)
    if saved is None:
      saved, cur = [x], x
      for i, (layer, slope) in enumerate(plan):
        cur, _ = conv_forward(layer, cur, None, True, slope, False, out_dtype_last if i == n - 1 else None)
        saved.append(cur)
      if out_complex:
        saved[-1] = copy_channels(saved[-1], 2, torch.float32)
    cur = saved[-1]
    ctx.out_complex = out_complex
    if out_complex:
      saved = list(saved[:-1]) + [None]        # (the un-activated last output is not needed by the backward)
    ctx.plan = plan
    ctx.w_req = [layer.weight.requires_grad and layer.train_weights for layer, _ in plan]
    ctx.save_for_backward(*saved)
    return cur

  @staticmethod
  def backward(ctx, gy):
    plan, saved = ctx.plan, ctx.saved_tensors
    n = len(plan)
    g = as_nhwc(gy)
    last_layer, last_slope = plan[-1]
    if n == 3 and all(ctx.w_req) and saved[1] is not None:
      ok, dx = convblock_fused_backward(plan, saved, g, ctx.needs_input_grad[0], ctx.out_complex, ctx.x_split)
      if ok:
        return (dx, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)
    if ctx.out_complex:              # [B,H,W,2] fp32 -> the data-gradient kernels' padded layout, one pass
      g = copy_channels(g, last_layer.cout_p, last_layer.dtype)
    elif g.dtype != last_layer.dtype:
      g = g.to(last_layer.dtype)
    if last_slope != 1.0:
      g = act_bwd(g, saved[n], last_slope)
    for i in range(n - 1, -1, -1):
      layer, _ = plan[i]
      xin = saved[i]
      in_hw = (xin.shape[1], xin.shape[2])
      g_out = g                          # data gradient first, weight gradient second (see ConvBnAct._finish_backward)
      if i > 0:
        prev_slope = plan[i - 1][1]
        g = conv_dgrad(layer, g_out, in_hw, g_src=xin if prev_slope != 1.0 else None, g_slope=prev_slope)
      elif ctx.needs_input_grad[0]:
        g = conv_dgrad(layer, g_out, in_hw)
      else:
        g = None
      if ctx.w_req[i]:
        conv_wgrad(layer, xin, None, g_out)
        if GRAD_READY_HOOK is not None and not _wgrad_deferred_mode():
          GRAD_READY_HOOK(layer)
    return (g, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)


# ----------------------------------------------------------------------------
# conv + BatchNorm(train/eval) + LeakyReLU (+ Dropout2d mask)
# ----------------------------------------------------------------------------


class BNState(object):
  """Affine params + running buffers of one nn.BatchNorm2d (fp32 tensors)."""

  def __init__(self, weight, bias, running_mean, running_var, eps=1e-5, momentum=0.1):
    self.weight, self.bias = weight, bias
    self.running_mean, self.running_var = running_mean, running_var
    self.eps, self.momentum = eps, momentum


BN_SMALL = True       # small feature maps: one-launch BatchNorm forward / backward (tests turn it off for A/B)
# A/B switches of the gradient fan-in fusions (tools/bench_toggle.py); all on in the product
FANIN_TAPS = True     # grouped discriminator pass: replay nodes hand one alias per consumer, gradients summed in-kernel
FANIN_REFINE = True   # RefineCombine: aliases for the prediction's / the refinement's second consumer
FANIN_LOGITS = True   # final discriminator conv: NCHW + NHWC gradients converted and summed in one launch
SEED_CONST = True     # backward seed gradient from a constant
PACK_BIAS = True      # bias refresh inside the re-pack launch


def _bn_forward(y, stats, bn, c_real, slope, training, dropmask, groups=1):
  b, h, w, cp = y.shape
  dev = y.device
  if training and stats is None and BN_SMALL and lib.raw('csmri_bn_small_ok')(b * h * w, cp, groups):
    # small map: statistics, finalize and normalise/activate in one launch
    mean = torch.empty(1, cp, dtype=torch.float32, device=dev)
    invstd = torch.empty(1, cp, dtype=torch.float32, device=dev)
    snap = torch.empty(2, cp, dtype=torch.float32, device=dev)
    z = torch.empty(b, h, w, cp, dtype=y.dtype, device=dev)
    lib.call('csmri_bn_small_fwd', dt_of(y), y.data_ptr(), y.stride(2), z.data_ptr(), z.stride(2), b, h * w, cp,
             c_real, bn.weight.data_ptr(), bn.bias.data_ptr(), float(slope), ptr(dropmask), bn.eps, bn.momentum,
             mean.data_ptr(), invstd.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
             snap.data_ptr(), stream())
    return z, mean, invstd, snap
  if training:
    if stats is None:
      rows = groups * lib.raw('csmri_bn_stats_rows')(b * h * w // groups, cp)
      stats = torch.empty(rows, 2, cp, dtype=torch.float32, device=dev)
      lib.call('csmri_bn_stats', dt_of(y), y.data_ptr(), y.stride(2), b * h * w, cp, stats.data_ptr(),
               groups, stream())
    mean = torch.empty(groups, cp, dtype=torch.float32, device=dev)
    invstd = torch.empty(groups, cp, dtype=torch.float32, device=dev)
    snap = torch.empty(2, cp, dtype=torch.float32, device=dev)
    lib.call('csmri_bn_finalize', stats.data_ptr(), stats.shape[0], cp, c_real, b * h * w, bn.eps,
             bn.momentum, mean.data_ptr(), invstd.data_ptr(), bn.running_mean.data_ptr(),
             bn.running_var.data_ptr(), groups, stream())
  else:
    assert groups == 1
    snap = None
    mean = torch.zeros(cp, dtype=torch.float32, device=dev)
    invstd = torch.zeros(cp, dtype=torch.float32, device=dev)
    mean[:c_real] = bn.running_mean
    invstd[:c_real] = torch.rsqrt(bn.running_var + bn.eps)
  z = torch.empty(b, h, w, cp, dtype=y.dtype, device=dev)
  lib.call('csmri_bn_act', dt_of(y), y.data_ptr(), y.stride(2), z.data_ptr(), z.stride(2), b, h * w, cp,
           c_real, mean.data_ptr(), invstd.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(),
           float(slope), ptr(dropmask), ptr(snap), groups, stream())
  return z, mean, invstd, snap


class ConvBnAct(torch.autograd.Function):
  """pad -> conv(no bias) -> BatchNorm2d -> LeakyReLU -> channel dropout mask.

  The conv epilogue emits per-channel partial sums (no extra pass for the batch
  statistics); backward = two-pass BN backward, wgrad, dgrad (+ reflect fold).
  ``dropmask``: [B,Cp] fp32 in {0, 1/(1-p)} or None.
  ``groups``: the batch holds that many equal sub-batches normalised independently (as if the
  module had been called once per sub-batch, in order)."""

  @staticmethod
  def run_forward(x0, x1, layer, bn, slope, training, dropmask, groups=1):
    """The forward launches; returns (y, z, mean, invstd, snap)."""
    ho, wo = layer.out_hw(x0.shape[1], x0.shape[2])
    m = x0.shape[0] * ho * wo
    small = m < 32768
    assert x0.shape[0] % groups == 0
    # epilogue partial rows cover 64 consecutive positions: a row must not straddle two groups
    fused_stats = training and not small and (m // groups) % 64 == 0
    y, stats = conv_forward(layer, x0, x1, False, 1.0, fused_stats, None)
    if stats is not None and stats.shape[0] % groups:
      stats = None
    z, mean, invstd, snap = _bn_forward(y, stats, bn, layer.cout, slope, training, dropmask, groups)
    return y, z, mean, invstd, snap

  @staticmethod
  def forward(ctx, x0, x1, weight, gamma, beta, layer, bn, slope, training, dropmask, groups=1):
    y, z, mean, invstd, snap = ConvBnAct.run_forward(x0, x1, layer, bn, slope, training, dropmask, groups)
    ctx.layer, ctx.bn, ctx.slope, ctx.training, ctx.groups = layer, bn, slope, training, groups
    ctx.c0 = x0.shape[3]
    ctx.in_hw = (x0.shape[1], x0.shape[2])
    # z is not kept for the BN backward: the activation sign is recomputed from y and `snap`
    ctx.save_for_backward(x0, x1, y, snap, mean, invstd, dropmask)
    ctx.w_req = weight.requires_grad and layer.train_weights
    return z

  @staticmethod
  def backward(ctx, gz, gz2=None):
    layer, bn = ctx.layer, ctx.bn
    x0, x1, y, snap, mean, invstd, dropmask = ctx.saved_tensors
    if not ctx.training:
      raise RuntimeError('backward through eval-mode BatchNorm is not on the training path')
    gz, gz2 = _two_grads(ctx, gz, gz2, layer.dtype)
    gz2_ptr, gz2_ps = (gz2.data_ptr(), gz2.stride(2)) if gz2 is not None else (0, 0)
    b, h, w, cp = y.shape
    dev = y.device
    groups = ctx.groups
    gy = torch.empty(b, h, w, cp, dtype=y.dtype, device=dev)
    want_affine = ctx.w_req
    if want_affine:
      if bn.weight.grad is None:
        bn.weight.grad = torch.zeros_like(bn.weight)
      if bn.bias.grad is None:
        bn.bias.grad = torch.zeros_like(bn.bias)
    acc_affine = 1
    if want_affine:
      bn.weight._kernel_grad = bn.bias._kernel_grad = True
    if want_affine and getattr(bn.weight, '_grad_fresh', False):     # lazily zeroed gradients: first write overwrites
      acc_affine = 0
      bn.weight._grad_fresh = False
      bn.bias._grad_fresh = False
    if BN_SMALL and lib.raw('csmri_bn_small_ok')(b * h * w, cp, groups):
      lib.call('csmri_bn_small_bwd', dt_of(y), gz.data_ptr(), gz.stride(2), gz2_ptr, gz2_ps, y.data_ptr(), y.stride(2),
               gy.data_ptr(), gy.stride(2), b, h * w, cp, layer.cout, mean.data_ptr(), invstd.data_ptr(),
               bn.weight.data_ptr(), float(ctx.slope), ptr(dropmask), snap.data_ptr(),
               bn.weight.grad.data_ptr() if want_affine else 0, bn.bias.grad.data_ptr() if want_affine else 0,
               acc_affine, stream())
      return ConvBnAct._finish_backward(ctx, gy, x0, x1, want_affine)
    rows = groups * lib.raw('csmri_bn_stats_rows')(b * h * w // groups, cp)
    partial = torch.empty(rows + groups, 2, cp, dtype=torch.float32, device=dev)
    lib.call('csmri_bn_bwd_reduce', dt_of(y), gz.data_ptr(), gz.stride(2), y.data_ptr(), y.stride(2),
             0, 0, b, h * w, cp, mean.data_ptr(), invstd.data_ptr(),
             float(ctx.slope), ptr(dropmask), partial.data_ptr(), snap.data_ptr(), groups, gz2_ptr, gz2_ps,
             stream())
    lib.call('csmri_bn_bwd_apply', dt_of(y), gz.data_ptr(), gz.stride(2), y.data_ptr(), y.stride(2),
             0, 0, gy.data_ptr(), gy.stride(2), b, h * w, cp, layer.cout,
             mean.data_ptr(), invstd.data_ptr(), bn.weight.data_ptr(), float(ctx.slope), ptr(dropmask),
             partial.data_ptr(), rows,
             bn.weight.grad.data_ptr() if want_affine else 0,
             bn.bias.grad.data_ptr() if want_affine else 0, acc_affine, snap.data_ptr(), groups, gz2_ptr, gz2_ps,
             stream())
    return ConvBnAct._finish_backward(ctx, gy, x0, x1, want_affine)

  @staticmethod
  def _finish_backward(ctx, gy, x0, x1, want_affine):
    layer = ctx.layer
    # ISSUE ORDER: the data gradient (the backward's critical chain) first, the weight gradient (side stream) second.
    # In a captured graph the first successor of a node stays on its hardware queue and the second one pays a
    # cross-queue hand-off (~10 us in the kernel trace): with the weight gradient issued first, every layer's
    # data gradient paid it (tools/trace_timeline.py).
    gx0 = gx1 = None
    if ctx.needs_input_grad[0] or (x1 is not None and ctx.needs_input_grad[1]):
      gx = conv_dgrad(layer, gy, ctx.in_hw)
      if x1 is None:
        gx0 = gx
      else:
        gx0, gx1 = gx[..., :ctx.c0], gx[..., ctx.c0:]
    if want_affine:
      conv_wgrad(layer, x0, x1, gy)
      if GRAD_READY_HOOK is not None and not _wgrad_deferred_mode():
        GRAD_READY_HOOK(layer)
    return gx0, gx1, None, None, None, None, None, None, None, None, None


class ConvBnActReplay(torch.autograd.Function):
  """Groups [lo, hi) of a grouped ConvBnAct pass that has ALREADY run (``rec`` = the tuple ConvBnAct.run_forward
  returned for the whole stacked batch, ``n`` images per group): forward launches nothing and returns those groups'
  rows of the stored activation; backward is ConvBnAct's own two-pass BatchNorm backward / data gradient / weight
  gradient on those rows with those groups' statistics.  This is how ONE discriminator pass over
  [pool-fake; real; current-fake] serves the two backward passes of a training step (the discriminator loss
  differentiates groups 0-1 w.r.t. the weights, the generator loss group 2 w.r.t. its input): reference
  training/adversarial_runner.py:332,338,354 are three module calls whose results do not depend on each other."""

  @staticmethod
  def forward(ctx, x0, x1, weight, gamma, beta, layer, bn, slope, dropmask_rows, rec, n, lo, hi, w_req, tap=False):
    y, z, mean, invstd, snap = rec
    ctx.layer, ctx.bn, ctx.slope, ctx.training, ctx.groups = layer, bn, slope, True, hi - lo
    ctx.c0 = x0.shape[3]
    ctx.in_hw = (x0.shape[1], x0.shape[2])
    ctx.save_for_backward(x0, x1, y[lo * n:hi * n], snap, mean[lo:hi], invstd[lo:hi], dropmask_rows)
    ctx.w_req = bool(w_req) and weight.requires_grad
    if not tap:
      return z[lo * n:hi * n]
    ctx.set_materialize_grads(False)           # two aliases, one per consumer (see ConvActReplay)
    return z[lo * n:hi * n], z[lo * n:hi * n]

  @staticmethod
  def backward(ctx, gz, gz2=None):
    return ConvBnAct.backward(ctx, gz, gz2)[:5] + (None,) * 10


def maxpool2_fwd(x, q_scale_ptr=0, amax_ptr=0, want_q=False):
  """MaxPool2d(2,2) on NHWC; returns (y, argmax uint8), or (y, argmax, y_q) when an fp8 copy of y (csmri_maxpool2_q:
  scale at device address ``q_scale_ptr``) and / or the running |y| maximum (``amax_ptr``) is asked for."""
  x = as_nhwc(x)
  b, h, w, c = x.shape
  y = torch.empty(b, h // 2, w // 2, c, dtype=x.dtype, device=x.device)
  arg = torch.empty(b, h // 2, w // 2, c, dtype=torch.uint8, device=x.device)
  if want_q or amax_ptr:
    yq = torch.empty(b, h // 2, w // 2, c, dtype=torch.uint8, device=x.device) if want_q else None
    lib.call('csmri_maxpool2_q', dt_of(x), x.data_ptr(), x.stride(2), y.data_ptr(), y.stride(2),
             arg.data_ptr(), b, h, w, c, ptr(yq), c, q_scale_ptr if want_q else 0, amax_ptr, stream())
    return y, arg, yq
  lib.call('csmri_maxpool2', dt_of(x), x.data_ptr(), x.stride(2), y.data_ptr(), y.stride(2),
           arg.data_ptr(), b, h, w, c, stream())
  return y, arg


def maxpool2_bwd(gy, arg, shape, g_src=None, g_slope=1.0, g_add=None, g_pooled=None):
  """Gradient of MaxPool2d(2,2).  ``g_add``: a second gradient of the pool's input, added in the same pass;
  ``g_src`` (the output of the activation layer that fed the pool): the result also carries that activation's
  derivative (csmri_maxpool2_bwd_act); ``g_pooled``: the same with the pool's OUTPUT as the gate (same result, a
  quarter of the gate's bytes: csmri_maxpool2_bwd_pooled_gate)."""
  b, h, w, c = shape
  gy = as_nhwc(gy)
  gx = torch.empty(b, h, w, c, dtype=gy.dtype, device=gy.device)
  if g_pooled is not None:
    assert g_src is None and g_add is None and g_pooled.dtype == gy.dtype and tuple(g_pooled.shape) == (b, h // 2, w // 2, c)
    lib.call('csmri_maxpool2_bwd_pooled_gate', dt_of(gy), gy.data_ptr(), gy.stride(2), arg.data_ptr(), gx.data_ptr(),
             gx.stride(2), b, h, w, c, g_pooled.data_ptr(), g_pooled.stride(2), float(g_slope), stream())
    return gx
  if g_src is not None or g_add is not None:
    for t in (g_src, g_add):
      assert t is None or (t.dtype == gy.dtype and tuple(t.shape) == (b, h, w, c) and t.stride(3) == 1)
    lib.call('csmri_maxpool2_bwd_act', dt_of(gy), gy.data_ptr(), gy.stride(2), arg.data_ptr(),
             gx.data_ptr(), gx.stride(2), b, h, w, c, ptr(g_src), g_src.stride(2) if g_src is not None else 0,
             float(g_slope), ptr(g_add), g_add.stride(2) if g_add is not None else 0, stream())
    return gx
  lib.call('csmri_maxpool2_bwd', dt_of(gy), gy.data_ptr(), gy.stride(2), arg.data_ptr(),
           gx.data_ptr(), gx.stride(2), b, h, w, c, stream())
  return gx


class MaxPool2(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x):
    y, arg = maxpool2_fwd(x)
    ctx.save_for_backward(arg)
    ctx.shape = tuple(x.shape)
    return y

  @staticmethod
  def backward(ctx, gy):
    arg, = ctx.saved_tensors
    return maxpool2_bwd(gy, arg, ctx.shape)


class MaxPool2Skip(torch.autograd.Function):
  """x -> (MaxPool2d(2,2)(x), x): the U-Net encoder's pool together with the skip connection that leaves the same
  tensor (models/unet.py:58-72).  Backward adds the skip gradient while un-pooling (one pass instead of the
  un-pooling launch + autograd's accumulation of the two gradients)."""

  @staticmethod
  def forward(ctx, x):
    y, arg = maxpool2_fwd(x)
    ctx.save_for_backward(arg)
    ctx.shape = tuple(x.shape)
    return y, x.view_as(x)

  @staticmethod
  def backward(ctx, gy, gskip):
    arg, = ctx.saved_tensors
    if gy is None:
      return gskip
    if gskip is not None:
      gskip = as_nhwc(gskip)
      if gskip.dtype != gy.dtype or tuple(gskip.shape) != ctx.shape or gskip.stride(3) != 1:
        return maxpool2_bwd(gy, arg, ctx.shape) + gskip
    return maxpool2_bwd(gy, arg, ctx.shape, g_add=gskip)


class Fp8Chain(object):
  """fp8 (e4m3fn) forward of a frozen conv / ReLU / max-pool stack with DELAYED scaling (BASELINE config 5: the VGG19
  perceptual loss is half the step's FLOPs and its weights never change).  Every 3 x 3 layer with a multiple of 128 input
  channels reads an fp8 copy of its input that the PRODUCING kernel's epilogue wrote (csmri_gconv_desc.out_q /
  csmri_maxpool2_q: no quantisation pass) and fp8 weights quantised once; outputs, saved activations and the whole
  backward stay bf16.  The scale of tensor j at step t is derived from the |x| maximum the producers accumulated at step
  t - 1 (csmri_fp8_scales_update, one tiny launch per forward); the first forward runs in bf16 and only collects the
  maxima.  ``margin``: bits of headroom kept above last step's maximum."""

  def __init__(self, plan, device, margin=1):
    self.slots = {}                       # plan index of the producer -> slot of its output tensor
    convs = [i for i, it in enumerate(plan) if it[0] == 'conv']
    for i in convs:
      if self.layer_ok(plan[i][1]) and i > 0:
        self.slots[i - 1] = len(self.slots)        # the item in front of an fp8 layer produces its input
    n = max(1, len(self.slots))
    self.amax = torch.zeros(n, dtype=torch.float32, device=device)
    self.scales = torch.ones(n, 2, dtype=torch.float32, device=device)
    self.margin = int(margin)
    self.ready = False                    # True once a forward has left scales behind
    self.disabled = not self.slots
    self.steps = 0

  @staticmethod
  def layer_ok(layer):
    return (layer.frozen and layer.dtype == torch.bfloat16 and layer.kh == 3 and layer.kw == 3 and layer.stride == 1 and
            not layer.upsample and layer.border == BORDER_ZERO and layer.cin_p % 128 == 0 and layer.cout_p % 128 == 0)

  def q_scale_ptr(self, slot):
    return self.scales.data_ptr() + 8 * slot

  def dq_scale_ptr(self, slot):
    return self.scales.data_ptr() + 8 * slot + 4

  def amax_ptr(self, slot):
    return self.amax.data_ptr() + 4 * slot

  def finish(self):
    """End of a forward: this pass's maxima become the next pass's scales."""
    lib.call('csmri_fp8_scales_update', self.amax.data_ptr(), self.scales.data_ptr(), len(self.slots), self.margin,
             stream())
    self.ready = True
    self.steps += 1


def frozen_conv_forward(layer, x, slope, xq=None, dq_ptr=0, out_slot=None, chain=None):
  """Bias + ReLU forward of one layer of a frozen stack: conv_forward, plus (chain given) fp8 operands when ``xq`` (the
  fp8 copy of x, dequantisation scale at ``dq_ptr``) is given, and an fp8 copy / the |y| maximum of the output for slot
  ``out_slot``.  Returns (y, y_q or None)."""
  if chain is None or (xq is None and out_slot is None):
    return conv_forward(layer, x, None, True, slope, False, None)[0], None
  _need_gpu(x)
  b, h, w, _ = x.shape
  ho, wo = layer.out_hw(h, w)
  y = torch.empty(b, ho, wo, layer.cout_p, dtype=layer.dtype, device=x.device)
  pl, pr, pt, pb = layer.pads
  d = lib.GConvDesc()
  keep = None
  if xq is not None:
    wq, kp, wsc = layer._pack_fp8()
    keep = (wq, wsc)
    d.dtype, d.in0, d.in0_pix_stride = FP8, xq.data_ptr(), xq.stride(2)
    d.in_dequant, d.w_dequant = dq_ptr, wsc.data_ptr() + 8
    wp = wq
  else:
    wp, kp, _, _ = layer._pack(0)
    d.dtype, d.in0, d.in0_pix_stride = dt_of(x), x.data_ptr(), x.stride(2)
  d.out_dtype = dt_of(y)
  d.B, d.Hin, d.Win, d.Cin = b, h, w, layer.cin_p
  d.upsample, d.border = 0, layer.border
  d.TH, d.TW, d.in_s = layer.kh, layer.kw, layer.stride
  d.dy0, d.dy_step, d.dx0, d.dx_step = -pt, 1, -pl, 1
  d.w, d.Kp, d.nclass, d.w_class_stride = wp.data_ptr(), kp, 1, 0
  d.out, d.out_pix_stride, d.Hout_t, d.Wout_t = y.data_ptr(), y.stride(2), ho, wo
  d.Ho, d.Wo, d.out_sy, d.out_sx, d.out_oy, d.out_ox = ho, wo, 1, 1, 0, 0
  d.Cout = layer.cout_p
  d.cin_real, d.cout_real = layer.cin, layer.cout
  bias = layer.bias_padded()
  d.bias = ptr(bias)
  d.act_slope = float(slope)
  yq = None
  if out_slot is not None:
    d.out_amax = chain.amax_ptr(out_slot)
    if chain.ready:
      yq = torch.empty(b, ho, wo, layer.cout_p, dtype=torch.uint8, device=x.device)
      d.out_q, d.out_q_pix_stride, d.out_q_scale = yq.data_ptr(), layer.cout_p, chain.q_scale_ptr(out_slot)
  _gconv_run(d, False, 2.0 * b * ho * wo * layer.cout * layer.cin * layer.kh * layer.kw)
  return y, yq


POOLED_GATE = True     # frozen stack: a pool's backward gates with the pooled tensor (tests turn it off for A/B)


class FrozenConvStackPair(torch.autograd.Function):
  """A frozen conv/ReLU/max-pool stack (VGG19 features) applied to a (prediction, target)
  pair as ONE batched forward: the stack has no batch-coupled op, so concatenating the two
  halves along the batch is exact, halves the launches and doubles M of the small late
  layers.  Only the prediction half is differentiated (input gradient only -- the weights
  are frozen): backward walks the stack on the first half of the saved activations.

  ``plan``: list of ('conv', ConvLayer, slope) / ('pool', None, None); ``taps``: indices into
  the plan after which a feature map is returned.  Returns 2*len(taps) tensors:
  prediction features then target features."""

  @staticmethod
  def forward(ctx, p_in, t_in, plan, taps, cabs=None, chain=None):
    # cabs = (dtype, mode): p_in / t_in are interleaved complex fp32 images [B,H,W,2]; their magnitudes (ComplexAbs
    # of that mode) are written straight into the two halves of the batched input (no torch.cat), and the backward
    # ends with the magnitude's derivative
    b = p_in.shape[0]
    ctx.cabs = cabs
    if cabs is not None:
      _need_gpu(p_in)
      x = torch.empty(2 * b, p_in.shape[1], p_in.shape[2], 8, dtype=cabs[0], device=p_in.device)
      complex_abs_raw(p_in.detach(), cabs[0], cabs[1], x[:b])
      complex_abs_raw(t_in.detach().contiguous(), cabs[0], cabs[1], x[b:])
      ctx.cabs_x = p_in.detach()
    else:
      x = torch.cat((p_in, t_in), 0)
    saved, shapes, feats = [], [], []
    pooled = {}
    if chain is not None and chain.disabled:
      chain = None
    xq, xq_slot = None, None              # fp8 copy of x and its slot in the chain (Fp8Chain)
    for i, (kind, layer, slope) in enumerate(plan):
      out_slot = chain.slots.get(i) if chain is not None else None
      if kind == 'conv':
        try:
          y, yq = frozen_conv_forward(layer, x, slope, xq if (chain is not None and Fp8Chain.layer_ok(layer)) else None,
                                      chain.dq_scale_ptr(xq_slot) if xq is not None else 0, out_slot, chain)
        except lib.CsmriError as e:
          # ONLY "this shape is outside the fp8-capable kernel" (tiny feature maps) turns the chain off -- for good, and
          # said aloud: the stack then runs bf16.  Anything else (a launch failure, out of memory) is an error.
          if chain is None or e.code != lib.E_UNSUPPORTED:
            raise
          import logging
          logging.getLogger(__name__).warning(
              'fp8 chain disabled: layer %d (%d -> %d channels at %d x %d) is outside the fp8 patch kernel; the frozen '
              'stack runs bf16', i, layer.cin, layer.cout, x.shape[1], x.shape[2])
          chain.disabled, chain, xq, out_slot = True, None, None, None
          y, yq = frozen_conv_forward(layer, x, slope)
        saved.append(y)
        shapes.append((x.shape[1], x.shape[2]))
        x, xq, xq_slot = y, yq, out_slot
      else:
        shapes.append((b,) + tuple(x.shape[1:]))
        if out_slot is not None:
          x, arg, xq = maxpool2_fwd(x, chain.q_scale_ptr(out_slot), chain.amax_ptr(out_slot), want_q=chain.ready)
          xq_slot = out_slot
        else:
          x, arg = maxpool2_fwd(x)
          xq, xq_slot = None, None
        saved.append(arg)
        pooled[i] = x                      # the gate of this pool's backward (the producer's ReLU at the routed position)
      if i in taps:
        feats.append(x)
    if chain is not None:
      chain.finish()
    ctx.plan, ctx.taps, ctx.b, ctx.shapes = plan, taps, b, shapes
    ctx.pool_keys = sorted(pooled) if POOLED_GATE else []
    ctx.save_for_backward(*(saved + [pooled[k] for k in ctx.pool_keys]))
    outs = [f[:b] for f in feats] + [f[b:] for f in feats]
    ctx.set_materialize_grads(False)        # no zero fills for the target half / unused feature maps
    ctx.mark_non_differentiable(*outs[len(feats):])
    return tuple(outs)

  @staticmethod
  def backward(ctx, *gouts):
    plan, taps, b = ctx.plan, sorted(ctx.taps), ctx.b
    saved = ctx.saved_tensors
    npk = len(ctx.pool_keys)
    pooled = dict(zip(ctx.pool_keys, saved[len(saved) - npk:])) if npk else {}
    gmap = {t: gouts[j] for j, t in enumerate(taps)}
    g = None
    act_done = False       # True when g already carries the activation derivative of plan[i]
    for i in range(len(plan) - 1, -1, -1):
      if i in gmap and gmap[i] is not None:
        gi = as_nhwc(gmap[i])
        if g is not None and act_done:
          # a feature tap joins here: its gradient still needs this layer's act derivative
          gi = act_bwd(gi, saved[i][:b], plan[i][2]) if plan[i][0] == 'conv' and plan[i][2] != 1.0 else gi
        g = gi if g is None else g + gi
      if g is None:
        continue
      kind, layer, slope = plan[i]
      if kind == 'conv':
        y = saved[i][:b]
        if g.dtype != layer.dtype:
          g = g.to(layer.dtype)
        gp = g if (act_done or slope == 1.0) else act_bwd(g, y, slope)
        act_done = False
        if i == 0 and not ctx.needs_input_grad[0]:
          return None, None, None, None, None, None
        # fuse the producer's activation derivative into this dgrad's epilogue when the
        # producer is the previous conv of the stack (its output IS this layer's input)
        prev = plan[i - 1] if i > 0 else None
        if prev is not None and prev[0] == 'conv' and prev[2] != 1.0 and \
            layer.border == BORDER_ZERO and not layer.upsample and layer.stride == 1:
          g = conv_dgrad(layer, gp, ctx.shapes[i], g_src=saved[i - 1][:b], g_slope=prev[2])
          act_done = True
        else:
          g = conv_dgrad(layer, gp, ctx.shapes[i])
      else:
        prev = plan[i - 1] if i > 0 else None
        if prev is not None and prev[0] == 'conv' and prev[2] != 1.0 and \
            saved[i - 1].dtype == g.dtype:
          # the pool's producer is an activated conv: its derivative rides on the un-pooling pass
          if i in pooled and pooled[i].dtype == g.dtype:
            g = maxpool2_bwd(g, saved[i][:b], ctx.shapes[i], g_pooled=pooled[i][:b], g_slope=prev[2])
          else:
            g = maxpool2_bwd(g, saved[i][:b], ctx.shapes[i], g_src=saved[i - 1][:b], g_slope=prev[2])
          act_done = True
        else:
          g = maxpool2_bwd(g, saved[i][:b], ctx.shapes[i])
          act_done = False
    if ctx.cabs is not None and g is not None:
      xc = ctx.cabs_x
      g = as_nhwc(g)
      dx = torch.empty_like(xc)
      lib.call('csmri_complex_abs_bwd', xc.data_ptr(), xc.shape[0] * xc.shape[1] * xc.shape[2], g.data_ptr(),
               dt_of(g), g.stride(2), 3 if ctx.cabs[1] == 3 else 1, ctx.cabs[1], dx.data_ptr(), 0, stream())
      g = dx
    return g, None, None, None, None, None


# ----------------------------------------------------------------------------
# data consistency
# ----------------------------------------------------------------------------


def copy_channels(x, c_dst, dtype=None):
  """Channel slice / zero-pad / cast of an NHWC tensor into a dense [B,H,W,c_dst]."""
  x = as_nhwc(x)
  b, h, w, c = x.shape
  out = torch.empty(b, h, w, c_dst, dtype=dtype or x.dtype, device=x.device)
  lib.call('csmri_copy_channels', x.data_ptr(), dt_of(x), x.stride(2), min(c, c_dst), out.data_ptr(),
           dt_of(out), c_dst, c_dst, b * h * w, stream())
  return out


def image_pool_exchange(x, pool, plan, out=None):
  """One query of the image history pool (utils/image_pool.py): x [n,...], pool [pool_size+1,...] (updated in
  place), plan int64 [5,n] on the device.  Returns the images handed to the discriminator (in ``out`` if given:
  a dense tensor of x's shape that does not overlap x)."""
  _need_gpu(x)
  x = x.contiguous()
  n = x.shape[0]
  assert pool.is_contiguous() and pool.dtype == x.dtype and tuple(pool.shape[1:]) == tuple(x.shape[1:])
  assert plan.is_contiguous() and plan.dtype == torch.int64 and tuple(plan.shape) == (5, n)
  if out is None:
    out = torch.empty_like(x)
  assert out.is_contiguous() and out.dtype == x.dtype and out.shape == x.shape and out.data_ptr() != x.data_ptr()
  lib.call('csmri_image_pool_exchange', x.data_ptr(), pool.data_ptr(), out.data_ptr(), plan.data_ptr(), n,
           x[0].numel() * x.element_size(), stream())
  return out


def undersample(img, mask_u8):
  """Forward model on the device: img interleaved complex fp32 [B,H,W,2], mask uint8 [B,H,W]
  -> (kspace, inp), both [B,H,W,2] fp32: kspace = m*FFT2(img), inp = IFFT2(kspace) (ortho)."""
  _need_gpu(img)
  img = img.contiguous().float()
  b, h, w, _ = img.shape
  ks = torch.empty(b, h, w, 2, dtype=torch.float32, device=img.device)
  inp = torch.empty(b, h, w, 2, dtype=torch.float32, device=img.device)
  lib.call('csmri_undersample', img.data_ptr(), mask_u8.contiguous().data_ptr(), ks.data_ptr(), inp.data_ptr(),
           b, h, w, stream())
  return ks, inp


def fft2(x, inverse=False, ortho=True):
  """Batched 2-D FFT of interleaved complex [B,H,W,2] (reference Fft2d / Ifft2d, myfft.py:78-128): fp32, or
  bf16 storage with fp32 arithmetic (csmri_fft2_bf16)."""
  _need_gpu(x)
  x = x.contiguous()
  assert x.dtype in (torch.float32, torch.bfloat16) and x.shape[-1] == 2
  b, h, w, _ = x.shape
  out = torch.empty_like(x)
  lib.call('csmri_fft2' if x.dtype == torch.float32 else 'csmri_fft2_bf16', x.data_ptr(), out.data_ptr(), b, h, w,
           int(inverse), int(ortho), stream())
  return out


class Fft2d(torch.autograd.Function):
  """orthoFFT2 with the adjoint as backward (myfft.py:78-102: grad_x = orthoIFFT2(grad_k))."""

  @staticmethod
  def forward(ctx, x, inverse=False):
    ctx.inverse = inverse
    return fft2(x, inverse, True)

  @staticmethod
  def backward(ctx, g):
    return fft2(g.contiguous(), not ctx.inverse, True), None


def dc_raw(x, k0, mask_u8, pad_dtype=None, out_fp32=False, x_split=False):
  """x: interleaved complex fp32 [B,H,W,2] or channels 0,1 of a [B,H,W,8] fp32
  conv output; k0: dense [B,H,W,2]; mask uint8 [B,H,W].  Returns (out [B,H,W,2]
  fp32, channel-padded copy [B,H,W,8] of pad_dtype or None).  x_split: a bf16 x read with out_fp32 is a
  CSMRI_BF16_SPLIT tensor (channels (0,1) + (2,3)); otherwise only channels 0,1 are read."""
  _need_gpu(x)
  x = as_nhwc(x)
  assert x.dtype in (torch.float32, torch.bfloat16) and x.shape[3] >= 2
  b, h, w, _ = x.shape
  out = torch.empty(b, h, w, 2, dtype=x.dtype, device=x.device)
  out_pad = None
  pad_code = 0
  if pad_dtype is not None:
    split = isinstance(pad_dtype, tuple)               # (torch.bfloat16, 'split'): CSMRI_BF16_SPLIT padded copy
    out_pad = torch.empty(b, h, w, 8, dtype=pad_dtype[0] if split else pad_dtype, device=x.device)
    pad_code = lib.BF16_SPLIT if split else dt_of(out_pad)
  if x.dtype == torch.bfloat16 and out_fp32:      # bf16 input (a padded gradient), fp32 arithmetic and result
    out = torch.empty(b, h, w, 2, dtype=torch.float32, device=x.device)
    lib.call('csmri_dc_in_bf16', x.data_ptr(), lib.BF16_SPLIT if x_split else lib.BF16, x.stride(2), ptr(k0),
             mask_u8.data_ptr(), out.data_ptr(), ptr(out_pad), pad_code, b, h, w, stream())
  elif x.dtype == torch.bfloat16:     # bf16 image storage (k0 stays fp32): the "bf16 cFFT" of config 5
    lib.call('csmri_dc_bf16', x.data_ptr(), x.stride(2), ptr(k0), mask_u8.data_ptr(), out.data_ptr(),
             ptr(out_pad), pad_code, b, h, w, stream())
  else:
    lib.call('csmri_dc', x.data_ptr(), x.stride(2), ptr(k0), mask_u8.data_ptr(), out.data_ptr(),
             ptr(out_pad), pad_code, 0, b, h, w, stream())
  return out, out_pad


class DataConsistency(torch.autograd.Function):
  """out = orthoIFFT2((1-m) orthoFFT2(x) + k0); backward = adjoint (k0 := 0).
  x: the dense complex fp32 [B,H,W,2] output of a conv block, or channels 0,1 of a [B,H,W,8] one (read in place).
  Returns out fp32 [B,H,W,2] and, if pad_dtype is given, the same result as a channel-padded [B,H,W,8] tensor (the
  next conv block's input layout).  BOTH outputs are differentiable views of the one result: the gradient of the
  padded copy (what the next block's data-gradient kernel writes, bf16) goes through the adjoint as it is
  (csmri_dc_in_bf16), without a conversion to the dense fp32 layout first."""

  @staticmethod
  def forward(ctx, x, k0, mask_u8, pad_dtype):
    out, out_pad = dc_raw(x, k0, mask_u8, pad_dtype)
    ctx.save_for_backward(mask_u8)
    ctx.cx, ctx.xdt = x.shape[3], x.dtype
    # the padded copy handed out is a split image: the gradient that comes back for it is one too (the fused backward
    # writes dX in the format of its x)
    ctx.pad_split = isinstance(pad_dtype, tuple)
    ctx.set_materialize_grads(False)       # (no zero fills for whichever of the two outputs is unused)
    if out_pad is None:
      return out
    return out, out_pad

  @staticmethod
  def backward(ctx, g, gpad=None):
    if g is None and gpad is None:
      return None, None, None, None
    mask_u8, = ctx.saved_tensors
    if g is None and gpad.dtype == torch.bfloat16 and ctx.xdt == torch.float32 and is_nhwc(gpad):
      gx, gp = dc_raw(gpad, None, mask_u8, ctx.xdt if ctx.cx == 8 else None, out_fp32=True, x_split=ctx.pad_split)
      return (gp if ctx.cx == 8 else gx), None, None, None
    if gpad is not None:                   # both outputs used (or an unusual dtype): sum in the dense layout
      gpn = as_nhwc(gpad)
      if ctx.pad_split and gpn.shape[3] >= 4:          # split gradient: hi + lo (rare path: plain torch ops)
        gp2 = (gpn[..., 0:2].float() + gpn[..., 2:4].float()).to(ctx.xdt).contiguous()
      else:
        gp2 = copy_channels(gpn, 2, ctx.xdt)
      g = gp2 if g is None else as_nhwc(g).to(ctx.xdt) + gp2
    g = as_nhwc(g)
    if g.dtype != ctx.xdt:
      g = copy_channels(g, 2, ctx.xdt)
    gx, gp = dc_raw(g, None, mask_u8, ctx.xdt if ctx.cx == 8 else None)
    return (gp if ctx.cx == 8 else gx), None, None, None


# ----------------------------------------------------------------------------
# complex magnitude, refinement combine, losses, metric, optimizer
# ----------------------------------------------------------------------------


def complex_abs_raw(x, dtype, mode, out=None):
  """|x| of interleaved complex fp32 [B,H,W,2] into NHWC [B,H,W,8] (no autograd); ``out``: a dense [B,H,W,8] view to
  write into (e.g. one half of a batch that two calls fill, instead of a torch.cat of their results)."""
  _need_gpu(x)
  assert x.is_contiguous() and x.dtype == torch.float32
  b, h, w, _ = x.shape
  if out is None:
    out = torch.empty(b, h, w, 8, dtype=dtype, device=x.device)
  assert out.is_contiguous() and out.dtype == dtype and tuple(out.shape) == (b, h, w, 8)
  lib.call('csmri_complex_abs', x.data_ptr(), b * h * w, out.data_ptr(), dt_of(out), 8, 8, mode, stream())
  return out


class ComplexAbs(torch.autograd.Function):
  """|x| of interleaved complex fp32 [B,H,W,2] -> NHWC [B,H,W,8] of ``dtype``.
  mode 0: channel 0 = |x|; mode 3: channels 0..2 = (|x|-mean_c)/std_c (VGG input).
  ``holder``: optional one-element list with a dense [B,H,W,8] tensor of ``dtype`` that receives the result (e.g. one
  group of a stacked batch, instead of a torch.cat of results); it is also what the call returns.  ``filled``: that
  tensor already holds the result (no launch; only the autograd link is made)."""

  @staticmethod
  def forward(ctx, x, dtype, mode, holder=None, filled=False):
    assert x.is_contiguous() and x.dtype == torch.float32
    b, h, w, _ = x.shape
    if filled:                       # the holder's tensor already IS |x| (complex_abs_raw wrote it): link only
      out = holder[0]
      assert out.dtype == dtype and tuple(out.shape) == (b, h, w, 8)
    else:
      out = complex_abs_raw(x, dtype, mode, holder[0] if holder is not None else None)
    ctx.save_for_backward(x)
    ctx.mode = mode
    return out

  @staticmethod
  def backward(ctx, g):
    x, = ctx.saved_tensors
    g = as_nhwc(g)
    b, h, w, _ = x.shape
    dx = torch.empty_like(x)
    lib.call('csmri_complex_abs_bwd', x.data_ptr(), b * h * w, g.data_ptr(), dt_of(g), g.stride(2),
             3 if ctx.mode == 3 else 1, ctx.mode, dx.data_ptr(), 0, stream())
    return dx, None, None, None, None


class RefineCombine(torch.autograd.Function):
  """RefinementWrapper 'real-penalty-add' tail (refinement_wrapper.py:169-194):
  pred = cat(unscale(scale(pre_real) + s*u), pre_imag).  pre: fp32 [B,H,W,2]
  (no grad), u: NHWC [B,H,W,8] channel 0, scale: fp32 [1].
  Returns (pred [B,H,W,2] fp32, scaled [B,H,W] fp32, pred_b, u_b): pred_b is a second alias of pred and u_b an alias
  of u, one per consumer (pred: the discriminator input and the VGG loss; u: this node and the feature penalty), so
  that the backward receives those gradients separately and sums them inside its one kernel; the scale parameter's
  gradient is written by that launch as well (lazy-zero protocol of FlatAdam, like the weight-gradient kernels).
  Four torch launches of the generator backward's head -- two adds, a zero fill, an accumulate -- are gone."""

  @staticmethod
  def forward(ctx, pre, u, scale):
    b, h, w, _ = pre.shape
    u = as_nhwc(u)
    mm = torch.empty(lib.raw('csmri_minmax_floats')(b), dtype=torch.float32, device=pre.device)
    lib.call('csmri_minmax_real', pre.data_ptr(), b, h * w, mm.data_ptr(), stream())
    pred = torch.empty_like(pre)
    scaled = torch.empty(b, h, w, dtype=torch.float32, device=pre.device)
    lib.call('csmri_refine_combine', pre.data_ptr(), u.data_ptr(), dt_of(u), u.stride(2),
             scale.data_ptr(), mm.data_ptr(), b, h * w, pred.data_ptr(), scaled.data_ptr(), stream())
    ctx.save_for_backward(u, scale, mm)
    ctx.scale_param = scale
    ctx.mark_non_differentiable(scaled)
    ctx.set_materialize_grads(False)
    return pred, scaled, pred[:], u[:]

  @staticmethod
  def backward(ctx, gpred, gscaled, gpred_b, gu_b):
    u, scale, mm = ctx.saved_tensors
    b, h, w, cp = u.shape
    if gpred is None:
      gpred, gpred_b = gpred_b, None
    if gpred is None:                       # only the alias of u was used: its gradient passes through
      return None, gu_b, None
    gpred = gpred.contiguous()
    if gpred_b is not None:
      gpred_b = gpred_b.contiguous()
    if gu_b is not None:
      gu_b = as_nhwc(gu_b)
      if gu_b.dtype != u.dtype:
        gu_b = gu_b.to(u.dtype)
    # (pad channels must be exact zeros: the kernel writes a pixel of 8 channels whole, wider ones are cleared here)
    du = (torch.empty if cp == 8 else torch.zeros)(b, h, w, cp, dtype=u.dtype, device=u.device)
    part = torch.empty(1026, dtype=torch.float32, device=u.device)
    p = ctx.scale_param
    direct = isinstance(p, torch.nn.Parameter) and p.requires_grad and ctx.needs_input_grad[2]
    acc = 1
    if direct:
      if p.grad is None:
        p.grad = torch.zeros_like(p)
      p._kernel_grad = True
      if getattr(p, '_grad_fresh', False):
        acc, p._grad_fresh = 0, False
    lib.call('csmri_refine_combine_bwd', gpred.data_ptr(), u.data_ptr(), dt_of(u), u.stride(2),
             scale.data_ptr(), mm.data_ptr(), b, h * w, du.data_ptr(), dt_of(du), du.stride(2),
             part.data_ptr(), ptr(gpred_b), ptr(gu_b), gu_b.stride(2) if gu_b is not None else 0,
             p.grad.data_ptr() if direct else 0, acc, stream())
    return None, du, (None if direct else part[:1])


class MeanLoss(torch.autograd.Function):
  """mean |a-b| (kind 0) or mean (a-b)^2 (kind 1) over the real channels of NHWC
  tensors; b may be None (= 0) and never receives a gradient (targets detached)."""

  @staticmethod
  def forward(ctx, a, b, kind, c_real):
    a = as_nhwc(a)
    if b is not None:
      b = as_nhwc(b)
      assert b.shape == a.shape and b.dtype == a.dtype
    bb, h, w, cp = a.shape
    res = torch.empty(1, dtype=torch.float32, device=a.device)
    work = torch.empty(lib.raw('csmri_loss_work_bytes')() // 8, dtype=torch.float64, device=a.device)
    lib.call('csmri_loss', kind, dt_of(a), a.data_ptr(), a.stride(2), ptr(b),
             b.stride(2) if b is not None else 0, bb * h * w, c_real, res.data_ptr(), work.data_ptr(),
             stream())
    ctx.save_for_backward(a, b)
    ctx.kind, ctx.c_real = kind, c_real
    return res.reshape(())

  @staticmethod
  def backward(ctx, g):
    a, b = ctx.saved_tensors
    bb, h, w, cp = a.shape
    coeff = g.reshape(1).float().contiguous()
    ga = torch.empty(bb, h, w, cp, dtype=a.dtype, device=a.device)
    lib.call('csmri_loss_bwd', ctx.kind, dt_of(a), a.data_ptr(), a.stride(2), ptr(b),
             b.stride(2) if b is not None else 0, bb * h * w, cp, ctx.c_real, coeff.data_ptr(), 1.0,
             ga.data_ptr(), ga.stride(2), 0, stream())
    return ga, None, None, None


class MultiMeanLoss(torch.autograd.Function):
  """sum_i weights[i] * mean|a_i - b_i| (kind 0) or mean (a_i - b_i)^2 (kind 1) over the real
  channels of n NHWC tensor pairs, one launch pair forward and one launch backward.
  Call: MultiMeanLoss.apply(kind, weights, chans, a_0..a_{n-1}, b_0..b_{n-1}); the b_i are
  targets (no gradient)."""

  @staticmethod
  def _items(kind, weights, chans, a_list, b_list, grads=None):
    n = len(a_list)
    arr = (lib.LossItem * n)()
    for i in range(n):
      a, b = a_list[i], b_list[i]
      bb, h, w, cp = a.shape
      it = arr[i]
      it.a, it.a_pix_stride = a.data_ptr(), a.stride(2)
      it.b, it.b_pix_stride = ptr(b), (b.stride(2) if b is not None else 0)
      it.npix, it.C, it.C_real, it.weight = bb * h * w, cp, chans[i], float(weights[i])
      it.dtype_plus1 = dt_of(a) + 1                   # per item: a feature list may mix bf16 maps and fp32 logits
      assert b is None or b.dtype == a.dtype
      if grads is not None:
        it.ga, it.ga_pix_stride = grads[i].data_ptr(), grads[i].stride(2)
    return arr

  @staticmethod
  def forward(ctx, kind, weights, chans, *tensors):
    n = len(tensors) // 2
    a_list = [as_nhwc(t) for t in tensors[:n]]
    b_list = [as_nhwc(t) if t is not None else None for t in tensors[n:]]
    dev = a_list[0].device
    res = torch.empty(1 + n, dtype=torch.float32, device=dev)
    work = torch.empty(lib.raw('csmri_loss_multi_work_bytes')(n) // 8, dtype=torch.float64, device=dev)
    arr = MultiMeanLoss._items(kind, weights, chans, a_list, b_list)
    lib.call('csmri_loss_multi', kind, dt_of(a_list[0]), arr, n, res.data_ptr(), work.data_ptr(), stream())
    ctx.save_for_backward(*(a_list + [b for b in b_list if b is not None]))
    ctx.has_b = [b is not None for b in b_list]
    ctx.kind, ctx.weights, ctx.chans, ctx.n = kind, list(weights), list(chans), n
    return res[0]

  @staticmethod
  def backward(ctx, g):
    saved = list(ctx.saved_tensors)
    n = ctx.n
    a_list, rest = saved[:n], saved[n:]
    b_list = [rest.pop(0) if hb else None for hb in ctx.has_b]
    coeff = g.reshape(1).float().contiguous()
    grads = [torch.empty(a.shape, dtype=a.dtype, device=a.device) for a in a_list]
    arr = MultiMeanLoss._items(ctx.kind, ctx.weights, ctx.chans, a_list, b_list, grads)
    lib.call('csmri_loss_multi_bwd', ctx.kind, dt_of(a_list[0]), arr, n, coeff.data_ptr(), stream())
    return (None, None, None) + tuple(grads) + (None,) * n


class BCELogits(torch.autograd.Function):
  """mean BCE(sigmoid(logits), target) with torch's log clamp (adversarial_loss.py)."""

  @staticmethod
  def forward(ctx, logits, target):
    lg = logits.contiguous().float()
    res = torch.empty(1, dtype=torch.float32, device=lg.device)
    lib.call('csmri_bce_logits', lg.data_ptr(), lg.numel(), float(target), 0, res.data_ptr(), stream())
    ctx.save_for_backward(lg)
    ctx.target = float(target)
    return res.reshape(())

  @staticmethod
  def backward(ctx, g):
    lg, = ctx.saved_tensors
    coeff = g.reshape(1).float().contiguous()
    out = torch.empty_like(lg)
    lib.call('csmri_bce_logits_bwd', lg.data_ptr(), lg.numel(), ctx.target, coeff.data_ptr(), 1.0,
             out.data_ptr(), 0, stream())
    return out, None


class WeightedSum(torch.autograd.Function):
  """total = sum_i w_i * loss_i of n <= 16 scalar loss tensors (fp32, list order): the runners'
  `torch.sum(torch.cat(losses) * weights)` (reference adversarial_runner.py:314-320) as one launch, one more backward.
  Call: WeightedSum.apply(weights_tuple, *losses)."""

  @staticmethod
  def forward(ctx, weights, *losses):
    n = len(losses)
    L = lib.ScalarList()
    vals = [v.detach().reshape(1).float().contiguous() for v in losses]
    for i in range(n):
      L.v[i], L.w[i] = vals[i].data_ptr(), float(weights[i])
    L.n = n
    out = torch.empty(1, dtype=torch.float32, device=vals[0].device)
    lib.call('csmri_weighted_sum', C.byref(L), out.data_ptr(), stream())
    ctx.weights, ctx.n = [float(w) for w in weights], n
    ctx.shapes = [(v.shape, v.dtype) for v in losses]
    return out.reshape(())

  @staticmethod
  def backward(ctx, g):
    L = lib.ScalarList()
    for i in range(ctx.n):
      L.w[i] = ctx.weights[i]
    L.n = ctx.n
    gg = g.reshape(1).float().contiguous()
    out = torch.empty(ctx.n, dtype=torch.float32, device=g.device)
    lib.call('csmri_weighted_sum_bwd', C.byref(L), gg.data_ptr(), out.data_ptr(), stream())
    return (None,) + tuple(out[i].reshape(shape).to(dt) if dt != torch.float32 else out[i].reshape(shape)
                           for i, (shape, dt) in enumerate(ctx.shapes))


def weighted_sum(losses, weights):
  """weights: host floats (list / 1-D CPU or device tensor read ONCE at construction by the caller)."""
  _need_gpu(losses[0])
  return WeightedSum.apply(tuple(float(w) for w in weights), *losses)


class BCELogitsPair(torch.autograd.Function):
  """mean BCE(sigmoid(l[:b]), t_first) + mean BCE(sigmoid(l[b:]), t_second) on the logits of a batched [first; second]
  discriminator pass (adversarial_loss.py:71-85 GANLoss 'disc'): one launch forward, one backward, no slicing of the
  logits in the autograd graph."""

  @staticmethod
  def forward(ctx, logits, t_first, t_second):
    lg = logits.contiguous().float()
    assert lg.shape[0] % 2 == 0
    res = torch.empty(3, dtype=torch.float32, device=lg.device)
    lib.call('csmri_bce_logits_pair', lg.data_ptr(), lg.numel() // 2, float(t_first), float(t_second), res.data_ptr(),
             stream())
    ctx.save_for_backward(lg)
    ctx.t = (float(t_first), float(t_second))
    return res[0]

  @staticmethod
  def backward(ctx, g):
    lg, = ctx.saved_tensors
    coeff = g.reshape(1).float().contiguous()
    out = torch.empty_like(lg)
    lib.call('csmri_bce_logits_pair_bwd', lg.data_ptr(), lg.numel() // 2, ctx.t[0], ctx.t[1], coeff.data_ptr(),
             out.data_ptr(), stream())
    return out, None, None


def psnr_mean(mse):
  """mean_b 10 log10(1 / mse_b) as a device scalar (csmri_psnr_mean)."""
  out = torch.empty(1, dtype=torch.float32, device=mse.device)
  m = mse.contiguous().float()
  lib.call('csmri_psnr_mean', m.data_ptr(), m.numel(), out.data_ptr(), stream())
  return out.reshape(())


def disc_accuracy(prob_fake, prob_real):
  """csmri_disc_accuracy: either argument may be None; [B, ...] fp32 probabilities, B <= 64."""
  ref = prob_fake if prob_fake is not None else prob_real
  pf = prob_fake.detach().contiguous().float() if prob_fake is not None else None
  pr = prob_real.detach().contiguous().float() if prob_real is not None else None
  b = ref.shape[0]
  out = torch.empty(1, dtype=torch.float32, device=ref.device)
  lib.call('csmri_disc_accuracy', ptr(pf), ptr(pr), b, ref[0].numel(), out.data_ptr(), stream())
  return out.reshape(())


def sigmoid_prob(logits):
  lg = logits.detach().contiguous().float()
  prob = torch.empty_like(lg)
  res = torch.empty(1, dtype=torch.float32, device=lg.device)
  lib.call('csmri_bce_logits', lg.data_ptr(), lg.numel(), 0.0, prob.data_ptr(), res.data_ptr(), stream())
  return prob


def psnr_mse(pred, target):
  """per-image MSE of clamp(|.|,0,1); pred/target interleaved complex [B,H,W,2]."""
  _need_gpu(pred)
  b, h, w, _ = pred.shape
  # converted copies are bound to locals so they outlive the launch (a temporary freed before the
  # kernel runs could be handed to the next allocation)
  p, t = pred.contiguous().float(), target.contiguous().float()
  assert p.shape == t.shape and p.shape[-1] == 2
  buf = torch.empty(b * 33, dtype=torch.float32, device=pred.device)   # results + 32 partials/image
  lib.call('csmri_psnr_mse', p.data_ptr(), t.data_ptr(), b, h * w, buf.data_ptr(), stream())
  return buf[:b]


def ssim(pred, target):
  """per-image SSIM of clamp(|.|,0,1) (11x11 gaussian window); pred/target interleaved complex
  [B,H,W,2] fp32.  Returns a [B] fp32 tensor."""
  _need_gpu(pred)
  b, h, w, _ = pred.shape
  out = torch.empty(b, dtype=torch.float32, device=pred.device)
  work = torch.empty(lib.raw('csmri_ssim_work_bytes')(b, h, w) // 8, dtype=torch.float64, device=pred.device)
  p, t = pred.contiguous().float(), target.contiguous().float()
  assert p.shape == t.shape and p.shape[-1] == 2
  lib.call('csmri_ssim', p.data_ptr(), t.data_ptr(), b, h, w, out.data_ptr(), work.data_ptr(), stream())
  return out


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0):
  """In-place Adam on flat fp32 buffers."""
  lib.call('csmri_adam', p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr),
           float(beta1), float(beta2), float(eps), int(step), float(grad_scale), stream())


def adam_step_dev(p, g, m, v, lr, beta1, beta2, eps, step_dev, grad_scale=1.0):
  """Adam with the step counter (int32 device tensor, steps already taken) on the device --
  the form that can be captured into a hipGraph and replayed.  ``lr``: a float, or a 1-element fp32 DEVICE tensor
  (csmri_adam_dev_lr: a captured graph then follows a learning-rate schedule without being captured again)."""
  if torch.is_tensor(lr):
    assert lr.is_cuda and lr.dtype == torch.float32 and lr.numel() == 1
    lib.call('csmri_adam_dev_lr', p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr.data_ptr(),
             float(beta1), float(beta2), float(eps), step_dev.data_ptr(), float(grad_scale), stream())
    return
  lib.call('csmri_adam_dev', p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr),
           float(beta1), float(beta2), float(eps), step_dev.data_ptr(), float(grad_scale), stream())
