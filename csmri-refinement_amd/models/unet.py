"""U-Net of the refinement generator, MI355X-native.

Implements the path the hot-path configs use (configs/2-refinement.json:31-47):
reflection-padded k x k stride-1 convs without bias, BatchNorm, LeakyReLU,
MaxPool2 between encoder scales, 'nn-resize-conv' upsampling with BN+activation
on the upsampling path only, 1x1 head.  Drop-in for reference models/unet.py:
same constructor arguments and state-dict keys (encode_units.{s}.encode.{1,2,5,6},
concat_decode_units.{s}.upsample.{2,3}, ...decode.0.encode.{1,2,5,6}, head.0).

Differences in HOW: each [pad, conv, BN, LeakyReLU] group is one fused operator
(conv kernel emits the BN partial sums; normalise+activation is a second pass);
nearest-upsampling and the skip concatenation are never materialised -- the conv
gathers from the low-resolution tensor / from two source tensors directly."""
import torch
import torch.nn as nn

from csmri_hip import ops
from models.utils import ensure_pack_group, ConvParams, BNParams, same_padding, default_compute_dtype, COMPUTE_DTYPES
from models.weight_inits import initialize_weights

REQUIRED_PARAMS = ['num_inputs', 'num_outputs', 'num_layers_per_scale', 'encode_filters',
                   'decode_filters', 'output_activation']
OPTIONAL_PARAMS = ['kernel_size', 'transposed_kernel_size', 'relu_leakiness', 'use_bn',
                   'upsampling_mode', 'padding', 'encoder_features', 'use_refinement',
                   'decoder_act_upsampling_only', 'compute_dtype']


def construct_model(conf, model_name, **kwargs):
  params = conf.to_param_dict(REQUIRED_PARAMS, OPTIONAL_PARAMS)
  model = UNET(**params)
  initialize_weights(model, conf.get_attr('weight_init', default={}))
  return model


class _ConvBnAct(object):
  """Container giving the reference's Sequential slot names to one fused group:
  slots (pad, conv, bn, act) -> conv at ``base+1``, BN at ``base+2``.  A plain object on
  purpose: the parameters are registered once, through the owner's ModuleDict."""

  def __init__(self, owner, base, cin, cout, k, padding, slope, dtype, upsample=False):
    self.conv = ConvParams(cin, cout, k, bias=False)
    self.conv.make_layer(1, same_padding(k, 1), padding, dtype, upsample=upsample)
    self.bn = BNParams(cout)
    self.slope = slope
    owner[str(base + 1)] = self.conv
    owner[str(base + 2)] = self.bn

  def run(self, x0, x1, training):
    return ops.ConvBnAct.apply(x0, x1, self.conv.weight, self.bn.weight, self.bn.bias,
                               self.conv.layer, self.bn.state(training), self.slope, training, None, 1)


class ConvEncodeUnit(nn.Module):
  def __init__(self, in_channels, num_layers, num_filters, kernel_size, relu_leakiness,
               downsample, padding, dtype):
    super(ConvEncodeUnit, self).__init__()
    self.downsample = downsample
    slots = {}
    self._groups = []
    cin = in_channels
    for i in range(num_layers):
      self._groups.append(_ConvBnAct(slots, 4 * i, cin, num_filters, kernel_size, padding,
                                     relu_leakiness, dtype))
      cin = num_filters
    self.encode = nn.ModuleDict(slots)

  def forward(self, x0, x1=None):
    x = self._groups[0].run(x0, x1, self.training)
    for g in self._groups[1:]:
      x = g.run(x, None, self.training)
    if self.downsample:
      return ops.MaxPool2Skip.apply(x)        # (pooled, skip): the two gradients meet inside the un-pooling pass
    return x


class ConvDecodeUnit(nn.Module):
  """'nn-resize-conv' decode unit with act_upsampling_only (unet.py:61-139)."""

  def __init__(self, in_channels, encoder_channels, num_filters, relu_leakiness, kernel_size,
               num_layers, padding, dtype):
    super(ConvDecodeUnit, self).__init__()
    slots = {}
    # reference slots: 0 Upsample, 1 pad, 2 conv, 3 BN, 4 act
    self._up = _ConvBnAct(slots, 1, in_channels, num_filters, kernel_size, padding,
                          relu_leakiness, dtype, upsample=True)
    self.upsample = nn.ModuleDict(slots)
    dec = {}
    if num_layers > 0:
      dec['0'] = ConvEncodeUnit(num_filters + encoder_channels, num_layers, num_filters,
                                kernel_size, relu_leakiness, False, padding, dtype)
    self.decode = nn.ModuleDict(dec)

  def forward(self, decode_path, encode_path=None):
    up = self._up.run(decode_path, None, self.training)
    if '0' not in self.decode:
      return up
    if encode_path is not None:
      # torch.cat((encode_path, x), dim=1) of the reference: two-source gather
      return self.decode['0'](encode_path, up)
    return self.decode['0'](up)


class UNET(nn.Module):
  DEFAULT_RELU_LEAKINESS = 0.1

  def __init__(self, num_inputs, num_outputs, num_layers_per_scale, encode_filters,
               decode_filters, output_activation, kernel_size=3, transposed_kernel_size=2,
               relu_leakiness=DEFAULT_RELU_LEAKINESS, use_bn=True, upsampling_mode='transposed',
               padding='zero', encoder_features=None, use_refinement=False,
               decoder_act_upsampling_only=False, compute_dtype=None):
    super(UNET, self).__init__()
    assert output_activation in ('softmax', 'tanh', 'none')
    unsupported = []
    if output_activation != 'none':
      unsupported.append('output_activation=%s' % output_activation)
    if upsampling_mode != 'nn-resize-conv':
      unsupported.append('upsampling_mode=%s' % upsampling_mode)
    if not use_bn:
      unsupported.append('use_bn=False')
    if not decoder_act_upsampling_only:
      unsupported.append('decoder_act_upsampling_only=False')
    if encoder_features is not None or use_refinement:
      unsupported.append('encoder_features/use_refinement')
    if unsupported:
      raise NotImplementedError('UNET options outside the hot path: ' + ', '.join(unsupported))
    dtype = COMPUTE_DTYPES.get(compute_dtype, compute_dtype) or default_compute_dtype()
    self.dtype = dtype
    if isinstance(relu_leakiness, float):
      relu_leakiness = (relu_leakiness, relu_leakiness)
    self.num_inputs, self.num_outputs = num_inputs, num_outputs

    n_enc = len(encode_filters)
    units, chans = [], []
    cin = num_inputs
    for s, f in enumerate(encode_filters):
      units.append(ConvEncodeUnit(cin, num_layers_per_scale, f, kernel_size, relu_leakiness[0],
                                  s != n_enc - 1, padding, dtype))
      chans.append(f)
      cin = f
    self.encode_units = nn.ModuleList(units)
    cdu = []
    for s, f in enumerate(decode_filters[:n_enc - 1]):
      cdu.append(ConvDecodeUnit(cin, chans[-(s + 2)], f, relu_leakiness[1], kernel_size,
                                num_layers_per_scale, padding, dtype))
      cin = f
    self.concat_decode_units = nn.ModuleList(cdu)
    du = []
    for f in decode_filters[n_enc - 1:]:
      du.append(ConvDecodeUnit(cin, 0, f, relu_leakiness[1], kernel_size, num_layers_per_scale,
                               padding, dtype))
      cin = f
    self.decode_units = nn.ModuleList(du)
    head = ConvParams(cin, num_outputs, 1, bias=True)
    head.make_layer(1, (0, 0, 0, 0), 'zero', dtype)
    self.head = nn.ModuleDict({'0': head})

  @staticmethod
  def weight_init_params(user_weight_init=None):
    return {'conv_weight': ('he_normal', UNET.DEFAULT_RELU_LEAKINESS),
            'batchnorm_weight': ('uniform', 0.98, 1.02)}

  def forward_nhwc(self, x):
    """x: NHWC [B,H,W,8] compute dtype -> NHWC [B,H,W,8] (num_outputs real channels)."""
    ensure_pack_group(self)
    skips = []
    for unit in self.encode_units:
      if unit.downsample:
        x, feat = unit(x)
        skips.append(feat)
      else:
        x = unit(x)
    for s, unit in enumerate(self.concat_decode_units):
      x = unit(x, skips[-(s + 1)])
    for unit in self.decode_units:
      x = unit(x)
    h = self.head['0']
    return ops.ConvAct.apply(x, None, h.weight, h.bias, h.layer, 1.0, None)

  def forward(self, inp):
    """inp: [B,num_inputs,H,W] fp32 -> [B,num_outputs,H,W] fp32 (reference API)."""
    x = ops.ToNHWC.apply(inp, self.dtype, ops.pad8(self.num_inputs))
    return ops.ToNCHW.apply(self.forward_nhwc(x), self.num_outputs)
