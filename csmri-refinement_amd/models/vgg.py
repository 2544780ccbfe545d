"""VGG19 feature extractor (frozen) for the perceptual loss, MI355X-native.

Mirrors reference models/vgg.py:8-80: torchvision's VGG19 ``features`` split into
blocks at the max-pools, block b holding the convs before the (b+1)-th pool;
state-dict keys ``blocks.{b}.{features_idx}.{weight,bias}`` plus mean/std buffers.
torchvision and its ImageNet weights are not available offline: weights are
initialised like torchvision's non-pretrained VGG (kaiming_normal fan_out, bias 0)
from ``seed`` and can be overwritten with load_state_dict (SURVEY A-9).

Each conv3x3+bias+ReLU is one implicit-GEMM kernel; the ImageNet mean/std
normalisation is folded into the complex-magnitude kernel that builds the input."""
import torch
import torch.nn as nn

from csmri_hip import ops
from models.utils import ConvParams, default_compute_dtype

VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M',
             512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']


class VGG19(nn.Module):
  LAST_FEATURE_MAP = 4

  def __init__(self, output_blocks=(LAST_FEATURE_MAP,), requires_grad=False, seed=0, dtype=None):
    super(VGG19, self).__init__()
    assert len(output_blocks) >= 1, 'Need at least one output block'
    self.output_blocks = sorted(output_blocks)
    last = self.output_blocks[-1]
    assert last <= 5, 'VGG19 has at most 6 blocks'
    dtype = dtype or default_compute_dtype()
    self.dtype = dtype
    from models.utils import default_fp8_forward
    self.fp8 = bool(default_fp8_forward()) and not requires_grad     # compute_dtype 'fp8': ops.Fp8Chain in features_pair
    gen = torch.Generator().manual_seed(seed)
    blocks, plan = [dict()], []
    cin, idx, block = 3, 0, 0
    for v in VGG19_CFG:
      if v == 'M':
        if block == last:
          break
        block += 1
        blocks.append(dict())
        plan.append(('pool', None, block))
        idx += 1
        continue
      conv = ConvParams(cin, v, 3, bias=True)
      nn.init.kaiming_normal_(conv.weight, mode='fan_out', nonlinearity='relu', generator=gen)
      nn.init.zeros_(conv.bias)
      conv.make_layer(1, (1, 1, 1, 1), 'zero', dtype, frozen=not requires_grad)
      blocks[block][str(idx)] = conv
      plan.append(('conv', conv, block))
      cin = v
      idx += 2
    self.blocks = nn.ModuleList([nn.ModuleDict(b) for b in blocks])
    self._plan = plan
    # plan indices after which a requested block's feature map is complete
    self._taps = []
    for i, (kind, conv, block) in enumerate(plan):
      last_of_block = i + 1 == len(plan) or plan[i + 1][2] != block
      if last_of_block and block in self.output_blocks:
        self._taps.append(i)
    for p in self.parameters():
      p.requires_grad = requires_grad
    self.register_buffer('mean', torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
    self.register_buffer('std', torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))

  def load_pretrained(self, path):
    """Load a torchvision vgg19 state_dict (``features.{i}.weight/bias``; a checkpoint of this
    module, ``blocks.{b}.{i}.*``, is accepted too) into the blocks.  Every conv of the requested
    blocks must be present; classifier entries are ignored."""
    sd = torch.load(path, map_location='cpu')
    if isinstance(sd, dict) and 'state_dict' in sd:
      sd = sd['state_dict']
    own = {}
    for b, block in enumerate(self.blocks):
      for idx, conv in block.items():
        for leaf in ('weight', 'bias'):
          src = sd.get('features.%s.%s' % (idx, leaf), sd.get('blocks.%d.%s.%s' % (b, idx, leaf)))
          if src is None:
            raise KeyError('VGG19 weights file %s lacks features.%s.%s' % (path, idx, leaf))
          dst = getattr(conv, leaf)
          if tuple(src.shape) != tuple(dst.shape):
            raise ValueError('features.%s.%s: shape %s, expected %s' % (idx, leaf, tuple(src.shape), tuple(dst.shape)))
          own[(b, idx, leaf)] = src
    with torch.no_grad():
      for (b, idx, leaf), src in own.items():
        getattr(self.blocks[b][idx], leaf).copy_(src.to(torch.float32))
    from models.utils import refresh_packs
    refresh_packs(self)

  def forward_nhwc(self, x):
    """x: NHWC [B,H,W,8], already normalised (ComplexAbs mode 3).  Returns the list of
    NHWC feature maps of the requested blocks."""
    out, cur = [], 0
    for kind, conv, block in self._plan:
      if block != cur:
        if cur in self.output_blocks:
          out.append(x)
        cur = block
      if kind == 'pool':
        x = ops.MaxPool2.apply(x)
      else:
        x = ops.ConvAct.apply(x, None, conv.weight, conv.bias, conv.layer, 0.0, None)
    if cur in self.output_blocks:
      out.append(x)
    return out

  def features_pair(self, p_in, t_in, complex_input=False):
    """Features of a (prediction, target) pair in one batched pass; only the prediction
    half carries gradient.  Returns (pred_feats, target_feats).  ``complex_input``: p_in / t_in are the
    interleaved complex images [B,H,W,2] fp32 and the normalised magnitude (ComplexAbs mode 3) is part of the op."""
    plan = [('conv', conv.layer, 0.0) if kind == 'conv' else ('pool', None, None)
            for kind, conv, _ in self._plan]
    chain = None
    if getattr(self, 'fp8', False) and self.dtype == torch.bfloat16:
      # compute_dtype 'fp8' (BASELINE config 5): the 3 x 3 layers from conv2_2 on multiply fp8 operands, delayed scaling
      chain = getattr(self, '_fp8_chain', None)
      if chain is None or chain.amax.device != p_in.device:
        chain = self._fp8_chain = ops.Fp8Chain(plan, p_in.device)
    outs = ops.FrozenConvStackPair.apply(p_in, t_in, plan, tuple(self._taps),
                                         (self.dtype, 3) if complex_input else None, chain)
    n = len(self._taps)
    return list(outs[:n]), list(outs[n:])

  def forward(self, inp):
    """inp: [B,3,H,W] fp32 in (0,1) -> list of [B,C,h,w] fp32 maps (reference API)."""
    x = (inp - self.mean) / self.std
    feats = self.forward_nhwc(ops.ToNHWC.apply(x, self.dtype, 8))
    return [ops.ToNCHW.apply(f, f.shape[3]) for f in feats]
