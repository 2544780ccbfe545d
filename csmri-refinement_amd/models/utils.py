"""Padding / bias rules shared by the models.

The reference materialises padding with nn.ZeroPad2d / nn.ReflectionPad2d
(models/utils.py:58-85); here the rule only yields the (left, right, top,
bottom) amounts and the border mode -- the gather-conv kernel applies them in
its address computation."""
import math
import os

import torch
import torch.nn as nn

import csmri_hip
from csmri_hip import ops

# 'fp8' (BASELINE config 5): bf16 activations / backward; the FROZEN VGG19 of the perceptual loss multiplies e4m3fn
# operands from conv2_2 on (ops.Fp8Chain: weights quantised once, activations quantised by the producing kernel's
# epilogue with delayed scaling) and the data-consistency layers keep their images in bf16; everything else exactly the
# bf16 path.  FP8_TRAINABLE (or CSMRI_FP8_TRAINABLE=1) additionally moves the forward products of the TRAINABLE
# convolutions to fp8 wherever the shape allows (ops.ConvLayer.fp8_ok): quantisation passes per launch, measured 12 %
# slower than bf16 end to end -- kept as an opt-in, not part of 'fp8'.
COMPUTE_DTYPES = {'bf16': torch.bfloat16, 'fp32': torch.float32,
                  'bfloat16': torch.bfloat16, 'float32': torch.float32, 'fp8': torch.bfloat16}
_DEFAULT_DTYPE = [torch.bfloat16]
_FP8_FORWARD = [False]
FP8_TRAINABLE = os.environ.get('CSMRI_FP8_TRAINABLE', '0') == '1'


def set_default_compute_dtype(name_or_dtype):
  _DEFAULT_DTYPE[0] = COMPUTE_DTYPES.get(name_or_dtype, name_or_dtype)
  _FP8_FORWARD[0] = name_or_dtype == 'fp8'


def default_fp8_forward():
  return _FP8_FORWARD[0]


def set_fp8_forward(module, on=True):
  """Switch the fp8 forward variant of every trainable convolution under ``module``."""
  n = 0
  for m in module.modules():
    if isinstance(m, ConvParams):
      m.fp8 = bool(on)
      if m.layer is not None and not m.layer.frozen:
        m.layer.fp8 = bool(on)
        n += 1
  return n


def default_compute_dtype():
  return _DEFAULT_DTYPE[0]


def same_padding(kernel_size, stride, dilation=1):
  """SAME padding of the reference (models/utils.py:75-85): total =
  ceil((k_eff-1)/stride); odd totals put the extra pixel right/bottom."""
  assert stride in (1, 2), 'Formula only works for stride 1 or 2'
  if dilation != 1:
    raise NotImplementedError('dilated convolutions are not on the hot path')
  k_eff = kernel_size + (kernel_size - 1) * (dilation - 1)
  total = int(math.ceil((k_eff - 1.0) / stride))
  lo = total // 2
  hi = lo if total % 2 == 0 else lo + 1
  return (lo, hi, lo, hi)


def need_bias(use_norm_layers, norm_layer):
  """models/utils.py:47-55."""
  if not use_norm_layers or use_norm_layers == 'not-first' or norm_layer == 'instance':
    return True
  return False


class ConvParams(nn.Module):
  """Holds the fp32 master weights of one convolution under the reference's
  state-dict names (``weight`` [Cout,Cin,KH,KW], optional ``bias``)."""

  def __init__(self, cin, cout, k, bias=True):
    super(ConvParams, self).__init__()
    self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
    self.bias = nn.Parameter(torch.empty(cout)) if bias else None
    # torch's nn.Conv2d default init (kept where the reference keeps it)
    nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
    if self.bias is not None:
      bound = 1.0 / math.sqrt(cin * k * k)
      nn.init.uniform_(self.bias, -bound, bound)
    self.kind = 'conv'
    self.layer = None

  def make_layer(self, stride, pads, border, dtype, upsample=False, frozen=False):
    self.layer = ops.ConvLayer(self.weight, self.bias, stride, pads, border, dtype,
                               upsample=upsample, frozen=frozen)
    self._layer_args = (stride, pads, border, dtype, upsample, frozen)
    self.layer.fp8 = bool(getattr(self, 'fp8', _FP8_FORWARD[0] and FP8_TRAINABLE)) and not frozen
    return self.layer

  def _apply(self, fn, *a, **k):
    out = super(ConvParams, self)._apply(fn, *a, **k)
    if self.layer is not None:       # parameters may have been re-created (.cuda())
      self.make_layer(*self._layer_args)
    return out


class BNParams(nn.Module):
  """nn.BatchNorm2d parameters/buffers under the reference's names (weight, bias,
  running_mean, running_var).  ``num_batches_tracked`` (present in checkpoints written
  by torch >= 0.4.1, absent in the reference's torch-0.3.1 ones) is kept as a host
  counter and only materialised in state_dict()."""

  def __init__(self, c, eps=1e-5, momentum=0.1):
    super(BNParams, self).__init__()
    self.weight = nn.Parameter(torch.ones(c))
    self.bias = nn.Parameter(torch.zeros(c))
    self.register_buffer('running_mean', torch.zeros(c))
    self.register_buffer('running_var', torch.ones(c))
    self.eps, self.momentum = eps, momentum
    self.kind = 'batchnorm'
    self.batches_tracked = 0

  def state(self, training=False, groups=1):
    if training:
      self.batches_tracked += groups
    return ops.BNState(self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                       self.momentum)

  def _save_to_state_dict(self, destination, prefix, keep_vars):
    super(BNParams, self)._save_to_state_dict(destination, prefix, keep_vars)
    destination[prefix + 'num_batches_tracked'] = torch.tensor(self.batches_tracked, dtype=torch.long)

  def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
    key = prefix + 'num_batches_tracked'
    if key in state_dict:
      self.batches_tracked = int(state_dict.pop(key))
    super(BNParams, self)._load_from_state_dict(state_dict, prefix, *args, **kwargs)


def ensure_pack_group(module):
  """All trainable conv layers of ``module`` re-pack their weights with one launch per mode
  (ops.PackGroup).  Cheap identity check so that .cuda()/.to() re-creations are picked up."""
  layers = [m.layer for m in module.modules() if isinstance(m, ConvParams) and m.layer is not None
            and not m.layer.frozen]
  key = tuple(id(l) for l in layers)
  if getattr(module, '_pack_group_key', None) != key:
    module._pack_group = ops.PackGroup(layers)
    module._pack_group_key = key
  return module._pack_group


def trainable_pack_groups(model):
  """The PackGroups of ``model``'s sub-networks if together they hold EVERY trainable conv layer
  of it (then an optimizer step on the model's weights only has to invalidate these), else None."""
  owners = [m for m in model.modules() if getattr(m, '_pack_group', None) is not None]
  groups = [ensure_pack_group(m) for m in owners]
  covered = set(id(l) for g in groups for l in g.layers)
  for m in model.modules():
    if isinstance(m, ConvParams) and m.layer is not None and not m.layer.frozen and id(m.layer) not in covered:
      return None
  return groups


def refresh_packs(model):
  """After weights were replaced wholesale (checkpoint load): bring every EXISTING packed copy up
  to date now and in place, frozen layers included -- captured graphs keep reading those buffers."""
  owners = []
  for m in model.modules():
    if getattr(m, '_pack_group', None) is not None:
      owners.append(m)
    if isinstance(m, ConvParams) and m.layer is not None and m.layer.frozen:
      layer = m.layer
      for mode in list(layer._packs):
        layer._packs[mode] = (None,) + tuple(layer._packs[mode][1:])
        layer._pack(mode)
      if layer._bias_pad is not None:
        layer._bias_pad = (None, layer._bias_pad[1])
        layer.bias_padded()
  for m in owners:
    ensure_pack_group(m).repack_stale()


def prepare_packs_for_capture(model):
  """Before a hipGraph capture: make sure the device-side item table of every pack group's multi-layer re-pack exists
  (it is built by a host-to-device copy at first use -- not allowed while a stream captures).  Freshness of the
  packs is left as it is: the captured step re-packs exactly where an eager step would."""
  for m in model.modules():
    if getattr(m, '_pack_group', None) is not None:
      g = ensure_pack_group(m)
      for mode in g.modes():
        g.ensure_table(mode)


def freeze(module):
  for p in module.parameters():
    p.requires_grad = False
  for m in module.modules():
    if isinstance(m, ConvParams) and m.layer is not None:
      m.layer.frozen = True
      m._layer_args = m._layer_args[:5] + (True,)      # survives the re-creation in _apply (.cuda())
