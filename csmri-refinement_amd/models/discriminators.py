"""CNN (PatchGAN-style) discriminator, MI355X-native.

Drop-in for the configured path of reference models/discriminators.py:50-247:
[reflection pad -> conv k x k (stride 1|2) -> (BatchNorm) -> LeakyReLU ->
(Dropout2d)] per layer, final conv without padding, outputs prob/logits/features.
State-dict keys follow the reference's Sequential slots (convs.{1,4,8,12,17,22},
BN at convs.{5,9,13,18,23}, final_conv.0).  Options the hot-path config never
sets (fc layers, weight norm, instance norm, average pooling) raise.

features are NHWC device tensors (post-dropout, as the reference's in-place
Dropout2d makes them -- SURVEY A-7); ``feature_channels`` gives the real channel
count of each for the feature-matching loss."""
import contextlib

import torch
import torch.nn as nn

from csmri_hip import ops, lib
from models.utils import (ensure_pack_group, ConvParams, BNParams, same_padding, need_bias, default_compute_dtype,
                          COMPUTE_DTYPES)
from models.weight_inits import initialize_weights

REQUIRED_PARAMS = ['num_inputs', 'num_filters_per_layer', 'strides']
OPTIONAL_PARAMS = ['kernel_sizes', 'fc_layers', 'spatial_shape', 'act_fn', 'relu_leakiness',
                   'use_norm_layers', 'norm_layer', 'use_weightnorm', 'padding',
                   'final_conv_kernel_size', 'final_average_pooling', 'use_biases',
                   'compute_features', 'dropout_after', 'dropout_prob', 'compute_dtype']


def construct_model(conf, model_name, **kwargs):
  if model_name != 'CNNDiscriminator':
    raise ValueError('Unknown discriminator {}'.format(model_name))
  params = conf.to_param_dict(REQUIRED_PARAMS, OPTIONAL_PARAMS)
  model = CNNDiscriminator(**params)
  initialize_weights(model, conf.get_attr('weight_init', default={}))
  return model


class CNNDiscriminator(nn.Module):
  DEFAULT_RELU_LEAKINESS = 0.2

  def __init__(self, num_inputs, num_filters_per_layer, strides, kernel_sizes=None, fc_layers=(),
               spatial_shape=None, act_fn='lrelu', relu_leakiness=DEFAULT_RELU_LEAKINESS,
               use_norm_layers=True, norm_layer='batch', use_weightnorm=False, padding='zero',
               final_conv_kernel_size=1, use_biases=True, final_average_pooling=False,
               compute_features=False, dropout_after=(), dropout_prob=0.5, compute_dtype=None):
    super(CNNDiscriminator, self).__init__()
    if len(fc_layers) > 0 or use_weightnorm or final_average_pooling or norm_layer != 'batch' \
        or act_fn != 'lrelu':
      raise NotImplementedError('CNNDiscriminator option outside the hot path')
    if kernel_sizes is None:
      kernel_sizes = 3
    if isinstance(kernel_sizes, int):
      kernel_sizes = [kernel_sizes] * len(num_filters_per_layer)
    assert len(num_filters_per_layer) == len(strides) == len(kernel_sizes)
    dtype = COMPUTE_DTYPES.get(compute_dtype, compute_dtype) or default_compute_dtype()
    self.dtype = dtype
    self.compute_features = compute_features
    self.slope = relu_leakiness
    self.dropout_prob = dropout_prob
    self.num_inputs = num_inputs

    slots, self._layers = {}, []
    slot, cin = 0, num_inputs
    for li, (f, k, s) in enumerate(zip(num_filters_per_layer, kernel_sizes, strides)):
      has_bn = use_norm_layers != 'not-first' and bool(use_norm_layers)
      use_bias = use_biases and need_bias(use_norm_layers, norm_layer)
      if use_norm_layers == 'not-first':
        use_norm_layers = True
      conv = ConvParams(cin, f, k, bias=use_bias)
      conv.make_layer(s, same_padding(k, s), padding, dtype)
      slots[str(slot + 1)] = conv                       # slot: pad
      bn = None
      if has_bn:
        bn = BNParams(f)
        slots[str(slot + 2)] = bn
      slot += 4 if has_bn else 3                        # (pad, conv, [bn], act)
      drop = li in dropout_after
      if drop:
        slot += 1
      self._layers.append((conv, bn, drop, f))
      cin = f
    self.convs = nn.ModuleDict(slots)
    fin = ConvParams(cin, 1, final_conv_kernel_size, bias=use_biases)
    fin.make_layer(1, (0, 0, 0, 0), 'zero', dtype)
    self.final_conv = nn.ModuleDict({'0': fin})
    self.injected_dropout = None      # list of [B,C] masks consumed in order (parity tests)
    self.last_dropout_masks = []

  def weight_init_params(self, user_weight_init=None):
    return {'conv_weight': ('normal', 0.0, 0.02), 'batchnorm_weight': ('normal', 1.0, 0.02)}

  def set_wgrad(self, enabled):
    """Disable weight-gradient kernels (generator phase: D's .grad is discarded by the
    reference anyway, SURVEY A-5)."""
    for conv, _, _, _ in self._layers:
      conv.layer.train_weights = enabled
    self.final_conv['0'].layer.train_weights = enabled

  def _dropmask(self, b, c, device, groups=1, layer_idx=0, n_drop=1):
    if self.injected_dropout:
      # injected masks are listed per module call; a grouped call consumes `groups` calls' worth
      ms = [self.injected_dropout[g * n_drop + layer_idx] for g in range(groups)]
      m = torch.cat([x.to(device=device, dtype=torch.float32).reshape(b // groups, -1) for x in ms], 0)
      if layer_idx == n_drop - 1:
        del self.injected_dropout[:groups * n_drop]
    else:
      p = self.dropout_prob
      m = torch.bernoulli(torch.full((b, c), 1.0 - p, device=device)) / (1.0 - p)
    cp = ops.pad8(c)
    if cp != c:
      m = torch.nn.functional.pad(m, (0, cp - c))
    self.last_dropout_masks.append(m)
    return m.contiguous()

  def _draw_masks(self, b, device):
    """All Dropout2d masks of one forward in ONE launch (csmri_dropout2d_mask: Philox4x32-10 keyed by a per-module
    seed, the call counter in device memory so that a replayed hipGraph draws fresh masks); one contiguous [B,C]
    view per dropout layer, in layer order.  The seed comes from torch's CPU generator at first use
    (torch.manual_seed makes runs reproducible; with several ranks training.distributed.decorrelate_rng_streams has
    re-seeded that generator per rank, so every rank draws its own masks).  The (seed, call counter) pair is not part
    of state_dict(): like the reference, whose checkpoints hold no RNG state (utils/checkpoints.py:9-40), a resumed
    run starts a new mask sequence."""
    cs = [f for _, bn, drop, f in self._layers if bn is not None and drop]
    if self.injected_dropout or not cs or any(c % 8 for c in cs) or device.type != 'cuda':
      return None
    st = getattr(self, '_rng_state', None)
    if st is None or st.device != device:
      seed = int(torch.randint(0, 2 ** 62, (1,)).item())
      st = self._rng_state = torch.tensor([seed, 0, 0], dtype=torch.int64, device=device)   # {seed, call, tickets}
    n = b * sum(cs)
    flat = torch.empty(n, dtype=torch.float32, device=device)
    lib.call('csmri_dropout2d_mask', flat.data_ptr(), n, float(self.dropout_prob), st.data_ptr(), ops.stream())
    views, off = [], 0
    for c in cs:
      views.append(flat[off:off + b * c].view(b, c))
      off += b * c
    return views

  def forward(self, inp=None, nhwc=None, groups=1):
    """inp: [B,num_inputs,H,W] fp32 (reference API) -- or ``nhwc=`` an NHWC tensor
    [B,H,W,pad8(num_inputs)] already in the compute dtype (internal fast path).
    ``groups`` > 1: the batch is that many stacked module calls (e.g. [fake; real]); BatchNorm
    statistics, running-stat updates and dropout draws are per sub-batch, in order, so the
    outputs equal ``groups`` separate calls -- with half the kernel launches."""
    x = nhwc if nhwc is not None else ops.ToNHWC.apply(inp, self.dtype, ops.pad8(self.num_inputs))
    ensure_pack_group(self)
    self.last_dropout_masks = []      # the masks of THIS call only (a list that grew every step pinned their buffers)
    feats, chans = [], []
    drawn = self._draw_masks(x.shape[0], x.device) if self.training else None
    n_drop = sum(1 for _, bn_, d_, _ in self._layers if bn_ is not None and d_)
    di = 0
    for conv, bn, drop, f in self._layers:
      if bn is None:
        x = ops.ConvAct.apply(x, None, conv.weight, conv.bias, conv.layer, self.slope, None)
      else:
        mask = None
        if drop and self.training:
          if drawn is not None:
            mask = drawn.pop(0)
            self.last_dropout_masks.append(mask)
          else:
            mask = self._dropmask(x.shape[0], f, x.device, groups, di, n_drop)
          di += 1
        x = ops.ConvBnAct.apply(x, None, conv.weight, bn.weight, bn.bias, conv.layer,
                                bn.state(self.training, groups), self.slope, self.training, mask,
                                groups)
      feats.append(x)
      chans.append(f)
    fin = self.final_conv['0']
    lg = ops.ConvAct.apply(x, None, fin.weight, fin.bias, fin.layer, 1.0, torch.float32)
    logits = ops.ToNCHW.apply(lg, 1)
    out = {'prob': _Sigmoid.apply(logits), 'logits': logits}
    if self.compute_features:
      out['features'] = feats + [lg]
      out['feature_channels'] = chans + [1]
    return out


  def forward_grouped(self, nhwc, groups, subs):
    """ONE pass over ``groups`` stacked module calls (reference training/adversarial_runner.py:332,338,354: the three
    discriminator forwards of a training step read the same weights and do not depend on each other), returned as one
    output dict per entry of ``subs`` = [(lo, hi, x_rows, wgrad, stream)]: groups [lo, hi) as a differentiable function
    of ``x_rows`` (a tensor holding exactly those groups' rows of ``nhwc``; it may carry a gradient) whose backward
    runs on those rows only -- with (``wgrad``) or without the weight gradients.  BatchNorm statistics, running-statistics
    updates and Dropout2d draws are per group, in group order: the same numbers as ``groups`` separate calls."""
    assert self.training and nhwc.shape[0] % groups == 0
    ensure_pack_group(self)
    self.last_dropout_masks = []      # the masks of THIS pass only
    n = nhwc.shape[0] // groups
    drawn = self._draw_masks(nhwc.shape[0], nhwc.device)
    n_drop = sum(1 for _, bn_, d_, _ in self._layers if bn_ is not None and d_)
    recs, di, x = [], 0, nhwc
    with torch.no_grad():
      for conv, bn, drop, f in self._layers:
        if bn is None:
          y, _ = ops.conv_forward(conv.layer, x, None, True, self.slope, False, None)
          recs.append((y, None))
          x = y
        else:
          mask = None
          if drop:
            if drawn is not None:
              mask = drawn.pop(0)
              self.last_dropout_masks.append(mask)
            else:
              mask = self._dropmask(x.shape[0], f, x.device, groups, di, n_drop)
            di += 1
          rec = ops.ConvBnAct.run_forward(x, None, conv.layer, bn.state(True, groups), self.slope, True, mask, groups)
          recs.append((rec, mask))
          x = rec[1]
      fin = self.final_conv['0']
      lg_all, _ = ops.conv_forward(fin.layer, x, None, True, 1.0, False, torch.float32)
    outs = []
    for lo, hi, x_rows, wgrad, sub_stream in subs:
      # ``sub_stream``: the stream this sub-graph is built on (autograd runs a node's backward on the stream its
      # forward ran on); ``x_rows`` may be a callable that makes the input there
      ctx = contextlib.nullcontext()
      if sub_stream is not None:
        sub_stream.wait_stream(torch.cuda.current_stream())
        ctx = torch.cuda.stream(sub_stream)
      with ctx:
        if callable(x_rows):
          x_rows = x_rows()
        assert x_rows.shape[0] == (hi - lo) * n
        feats, chans, x = [], [], x_rows
        # with features handed out every layer output has two consumers (the next layer, the feature-matching loss):
        # the replay nodes return one alias for each and sum the two gradients inside their first backward kernel
        tap = bool(self.compute_features) and ops.FANIN_TAPS
        for (conv, bn, drop, f), (rec, mask) in zip(self._layers, recs):
          if bn is None:
            x = ops.ConvActReplay.apply(x, None, conv.weight, conv.bias, conv.layer, self.slope,
                                        [rec[lo * n:hi * n]], wgrad, tap)
          else:
            x = ops.ConvBnActReplay.apply(x, None, conv.weight, bn.weight, bn.bias, conv.layer,
                                          ops.BNState(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                                                      bn.momentum),
                                          self.slope, mask[lo * n:hi * n] if mask is not None else None, rec, n, lo, hi,
                                          wgrad, tap)
          x, feat = x if tap else (x, x)
          feats.append(feat)
          chans.append(f)
        if ops.FANIN_LOGITS:
          lg, logits = ops.ConvActReplay.apply(x, None, fin.weight, fin.bias, fin.layer, 1.0, [lg_all[lo * n:hi * n]],
                                               wgrad, False, 1)
        else:
          lg = ops.ConvActReplay.apply(x, None, fin.weight, fin.bias, fin.layer, 1.0, [lg_all[lo * n:hi * n]], wgrad)
          logits = ops.ToNCHW.apply(lg, 1)
        out = {'prob': _Sigmoid.apply(logits), 'logits': logits}
        if self.compute_features:
          out['features'] = feats + [lg]
          out['feature_channels'] = chans + [1]
      outs.append(out)
    return outs


class _Sigmoid(torch.autograd.Function):
  """prob = sigmoid(logits).  The GAN criterion consumes ``logits`` directly (fused
  BCE-on-logits kernel); prob is kept for the API and the accuracy metric."""

  @staticmethod
  def forward(ctx, logits):
    p = ops.sigmoid_prob(logits)
    ctx.save_for_backward(p)
    return p

  @staticmethod
  def backward(ctx, g):
    p, = ctx.saved_tensors
    return g * p * (1 - p)
