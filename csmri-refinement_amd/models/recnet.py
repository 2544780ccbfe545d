"""RecNet: cascade of conv blocks and k-space data-consistency layers
(Schlemper et al. deep cascade), MI355X-native.

Drop-in for the reference's models/recnet.py: same constructor arguments,
``forward(inp, kspace, mask)`` with [B,2,H,W] fp32 tensors, same state-dict keys
(``conv_blocks.{b}.layers.{1,4,7}.{weight,bias}``, recnet.py:29-62,121-134).
Internally activations are NHWC in the compute dtype, every conv is one
implicit-GEMM MFMA kernel with padding/bias/LeakyReLU fused, and each DC layer is
the fused FFT/merge/iFFT of csmri_dc whose last pass also emits the next block's
channel-padded input."""
import torch
import torch.nn as nn

from csmri_hip import ops
from models.utils import ConvParams, ensure_pack_group, same_padding, default_compute_dtype
from models.weight_inits import initialize_weights

RECNET_REQUIRED_PARAMS = ['num_blocks', 'num_convs', 'num_filters']
RECNET_OPTIONAL_PARAMS = ['num_final_outputs', 'dilations_per_conv', 'kernel_size',
                          'relu_leakiness', 'padding', 'use_refinement', 'skip_final_dc',
                          'return_intermediate_recs', 'compute_dtype', 'dc_storage', 'image_precision']


def construct_model(conf, model_name, **kwargs):
  params = conf.to_param_dict(RECNET_REQUIRED_PARAMS, RECNET_OPTIONAL_PARAMS)
  model = RecNet(**params)
  for block in model.conv_blocks:
    initialize_weights(block, conf.get_attr('weight_init', default={}))
  return model


class ConvBlock(nn.Module):
  """(pad, conv, LeakyReLU) x (n-1), pad, conv -- reference recnet.py:29-62.
  Sequential slot of conv i is 3*i+1 (pad, conv, act triplets)."""

  def __init__(self, num_convs, num_filters, kernel_size, relu_leakiness, padding='zero',
               num_inputs=2, num_outputs=2, dtype=None):
    super(ConvBlock, self).__init__()
    self.slope = relu_leakiness
    self.num_convs = num_convs
    dtype = dtype or default_compute_dtype()
    convs = {}
    cin = num_inputs
    for i in range(num_convs):
      cout = num_filters if i < num_convs - 1 else num_outputs
      cp = ConvParams(cin, cout, kernel_size, bias=True)
      cp.make_layer(1, same_padding(kernel_size, 1), padding, dtype)
      convs[str(3 * i + 1)] = cp
      cin = cout
    self.layers = nn.ModuleDict(convs)
    self.num_outputs = num_outputs
    self.out_dtype = torch.float32      # RecNet(dc_storage='bf16') switches the block's last conv to bf16 output

  def weight_init_params(self, user_weight_init=None):
    first = self.layers['1']
    return {'conv_weight': ('he_normal', RecNet.DEFAULT_RELU_LEAKINESS),
            first: {'weight': ('xavier', 1.0)}}

  def forward(self, x, split=False):
    """x: NHWC [B,H,W,8] compute dtype (``split``: a CSMRI_BF16_SPLIT image, bf16 hi + lo of the two real channels).
    Returns interleaved complex fp32 [B,H,W,2] when the block ends in 2 channels (feeds DC), else NHWC."""
    n = self.num_convs
    cps = [self.layers[str(3 * i + 1)] for i in range(n)]
    plan = [(cp.layer, 1.0 if i == n - 1 else self.slope) for i, cp in enumerate(cps)]
    params = [p for cp in cps for p in (cp.weight, cp.bias)]
    out = self.out_dtype
    if self.num_outputs == 2 and out == torch.float32 and x.is_cuda:
      out = ('complex', torch.float32, 'split') if split else ('complex', torch.float32)   # dense [B,H,W,2] fp32: what DC consumes
    return ops.ConvActStack.apply(x, plan, out, *params)


class RecNet(nn.Module):
  DEFAULT_RELU_LEAKINESS = 0.01

  def __init__(self, num_blocks, num_convs, num_filters, num_final_outputs=2,
               dilations_per_conv=1, kernel_size=3, relu_leakiness=DEFAULT_RELU_LEAKINESS,
               padding='zero', use_refinement=False, skip_final_dc=False,
               return_intermediate_recs=False, compute_dtype=None, dc_storage=None, image_precision=None):
    super(RecNet, self).__init__()
    if isinstance(num_filters, int):
      num_filters = [num_filters] * num_blocks
    assert len(num_filters) == num_blocks, 'Number of given filters must match number of blocks'
    if not (dilations_per_conv == 1 or all(d == 1 for d in dilations_per_conv)):
      raise NotImplementedError('dilated RecNet convolutions are outside the hot path')
    if num_final_outputs != 2:
      raise NotImplementedError('RecNet with num_final_outputs != 2 is outside the hot path')
    from models.utils import COMPUTE_DTYPES
    dtype = COMPUTE_DTYPES.get(compute_dtype, compute_dtype) or default_compute_dtype()
    self.dtype = dtype
    self.conv_blocks = nn.ModuleList([
        ConvBlock(num_convs, nf, kernel_size, relu_leakiness, padding, dtype=dtype)
        for nf in num_filters])
    # plain python list in the reference (recnet.py:128-134): no parameters/buffers
    self.dc_layers = list(range(num_blocks if not skip_final_dc else num_blocks - 1))
    self.use_refinement = use_refinement
    self.skip_final_dc = skip_final_dc
    self.return_intermediate_recs = return_intermediate_recs
    # 'bf16': the conv blocks hand the data-consistency layers bf16 images and the FFT passes keep their
    # images / intermediates in bf16 (csmri_dc_bf16, the "bf16 cFFT" of BASELINE config 5; default with
    # compute_dtype 'fp8').  Default 'fp32' keeps the image path of the cascade in fp32 (PSNR parity).
    from models.utils import default_fp8_forward
    if dc_storage is None:
      dc_storage = 'bf16' if (compute_dtype == 'fp8' or (compute_dtype is None and default_fp8_forward())) else 'fp32'
    assert dc_storage in ('fp32', 'bf16')
    self.dc_storage = dc_storage
    if dc_storage == 'bf16':
      assert dtype == torch.bfloat16 and not use_refinement, 'bf16 DC storage needs bf16 compute without refinement'
      for block in self.conv_blocks:
        block.out_dtype = torch.bfloat16
    # The image between two cascades (and its gradient on the way back) is the one 2-channel tensor of the block: in bf16
    # compute it travels as a channel-padded bf16 pixel of 8, 6 channels of which are padding.  'split' (default with
    # bf16 compute and the fp32 DC path) stores bf16 hi + lo of the two real channels in that same pixel
    # (CSMRI_BF16_SPLIT): 16 significant bits for the block input and for dX at no extra byte; 'bf16' rounds both to 8
    # bits as rounds 1-3 did.  Measured on the trained 5-cascade at 45 dB: the bf16 forward pass loses 0.10 dB to the fp32
    # one, 0.045 dB of it through this rounding (tests/c2_forward_floor.py, profiles/r04_c2_forward_floor.json).
    if image_precision is None:
      image_precision = 'split' if (dtype == torch.bfloat16 and dc_storage == 'fp32' and num_convs == 3) else 'bf16'
    assert image_precision in ('split', 'bf16')
    self.split_images = image_precision == 'split' and dtype == torch.bfloat16 and dc_storage == 'fp32'

  def forward(self, inp, kspace, mask):
    """inp, kspace, mask: [B,2,H,W] fp32 (re, im planes).  Returns [B,2,H,W]."""
    ensure_pack_group(self)          # trainable: one multi-layer re-pack per mode after an optimizer step (15 x 2 launches otherwise)
    split = self.split_images and inp.is_cuda
    x_pad = ops.ToNHWC.apply(inp, self.dtype, 8, split)      # conv input layout
    # interleaved complex copy of the input: only the residual form (use_refinement) and a cascade without any
    # block read it
    x_c = ops.ToNHWC.apply(inp, torch.float32, 2) if (self.use_refinement or not len(self.conv_blocks)) else None
    k0 = ops.nchw_to_nhwc(kspace, torch.float32, 2)
    m8 = ops.mask_to_u8(mask)
    recs = []
    nb = len(self.conv_blocks)
    for idx, block in enumerate(self.conv_blocks):
      y = block(x_pad, split)                                # fp32 [B,H,W,2] (re, im), or [B,H,W,8] with ch 0,1 = re,im
      if self.use_refinement:
        y = y + (x_c if y.shape[3] == 2 else _CastPad.apply(x_c, torch.float32))
      if idx < len(self.dc_layers):
        want_pad = ((self.dtype, 'split') if split else self.dtype) if idx < nb - 1 else None
        res = ops.DataConsistency.apply(y, k0, m8, want_pad)
        if want_pad is not None:
          x_c, x_pad = res                 # both differentiable (ops.DataConsistency)
        else:
          x_c = res
        if self.return_intermediate_recs:
          recs.append(ops.ToNCHW.apply(x_c, 2))
      else:
        x_c = y if y.shape[3] == 2 else _Slice2.apply(y)
        if idx < nb - 1:
          x_pad = _CastPad.apply(x_c, self.dtype)
    out = ops.ToNCHW.apply(x_c, 2)
    out._nhwc = x_c.detach()           # the same image in device layout (metrics read it instead of converting back)
    if self.return_intermediate_recs:
      return {'pred': out, 'reconstructions': recs}
    return out


class _CastPad(torch.autograd.Function):
  """[B,H,W,2] fp32 -> channel-padded [B,H,W,8] of ``dtype``."""

  @staticmethod
  def forward(ctx, x_c, dtype):
    return ops.copy_channels(x_c, 8, dtype)

  @staticmethod
  def backward(ctx, g):
    return ops.copy_channels(g, 2, torch.float32), None


class _Slice2(torch.autograd.Function):
  """channels 0,1 of a [B,H,W,8] fp32 tensor as dense interleaved complex."""

  @staticmethod
  def forward(ctx, y):
    return ops.copy_channels(y, 2, torch.float32)

  @staticmethod
  def backward(ctx, g):
    return ops.copy_channels(g, 8, torch.float32)
