"""CPU oracle for the CS-MRI GAN-refinement training path.

TEST INFRASTRUCTURE ONLY.  This file is a plain PyTorch-CPU (fp32 / fp64)
restatement of the reference algorithm.  It is the checker for the HIP path:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product (``csmri-refinement_amd/``) never
does, and fails loudly when the HIP library is missing.

Parity status: PINNED against golden vectors generated in the build container
by importing the reference (``/root/reference``) under compatibility shims --
see ``tests/golden/make_golden.py`` (generator) and ``tests/test_oracle_golden.py``
(check).  Third-party arithmetic the reference reaches but does not vendor:
``pytorch-fft==0.14`` (cuFFT binding, environment.yml:154) -- pinned only by the
numpy known-answer relation of myfft.py:166-243, restated here as
``torch.fft.fft2(norm='ortho')``; torchvision VGG19 weights -- unpinned, both
sides use the same seeded synthetic weights.

Everything is functional: parameters live in flat ``dict[str, Tensor]`` keyed
with the reference's state-dict names so that reference
checkpoints/fixtures plug in directly.

Citations are ``path:line`` relative to the reference repository root.
"""
import math
import random as _pyrandom

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# padding rule                                              models/utils.py:58-85
# --------------------------------------------------------------------------


def same_padding(kernel_size, stride, dilation=1):
  """Returns (left, right, top, bottom) of the reference's SAME padding.

  total = ceil((k_eff - 1) / stride); even totals are split evenly, odd totals
  put the extra pixel right/bottom (models/utils.py:66-72,75-85).
  """
  k_eff = kernel_size + (kernel_size - 1) * (dilation - 1)
  total = int(math.ceil((k_eff - 1.0) / stride))
  lo = total // 2
  hi = lo if total % 2 == 0 else lo + 1
  return (lo, hi, lo, hi)


def pad2d(x, pads, mode):
  if all(p == 0 for p in pads):
    return x
  if mode == 'zero':
    return F.pad(x, pads)
  if mode == 'reflection':
    return F.pad(x, pads, mode='reflect')
  if mode == 'replication':
    return F.pad(x, pads, mode='replicate')
  raise ValueError(mode)


def conv_same(x, weight, bias, stride, mode):
  k = weight.shape[-1]
  return F.conv2d(pad2d(x, same_padding(k, stride), mode), weight, bias,
                  stride=stride)


# --------------------------------------------------------------------------
# data consistency                                  myfft.py:78-163 (a3,a4,a5)
# --------------------------------------------------------------------------


def dc_layer(x, k0, mask):
  """x_res = orthoIFFT2((1 - m) * orthoFFT2(x) + k0)      myfft.py:131-163.

  x, k0, mask: [B,2,H,W] (re, im planes).  Note the noiseless branch does not
  multiply k0 by the mask (myfft.py:141).
  """
  k = torch.fft.fft2(torch.complex(x[:, 0], x[:, 1]), norm='ortho')
  k = torch.stack((k.real, k.imag), dim=1)
  out = (1 - mask) * k + k0
  r = torch.fft.ifft2(torch.complex(out[:, 0], out[:, 1]), norm='ortho')
  return torch.stack((r.real, r.imag), dim=1)


def dc_adjoint(g, mask):
  """Adjoint of dc_layer w.r.t. x: orthoIFFT2((1-m) * orthoFFT2(g)).

  Follows from myfft.py:92-102,119-128 (backward of Fft2d/Ifft2d is the
  adjoint = inverse of the unitary transform).
  """
  return dc_layer(g, torch.zeros_like(g), mask)


# --------------------------------------------------------------------------
# RecNet                                              models/recnet.py:29-161
# --------------------------------------------------------------------------


def conv_block(P, prefix, x, num_convs=3, slope=0.01, padding='zero'):
  """ConvBlock: (pad, conv3x3, lrelu) x (n-1), pad, conv3x3.  recnet.py:29-62.

  Sequential indices: conv i sits at ``layers.{3*i+1}`` (pad, conv, act).
  """
  for i in range(num_convs):
    w = P['%s.layers.%d.weight' % (prefix, 3 * i + 1)]
    b = P['%s.layers.%d.bias' % (prefix, 3 * i + 1)]
    x = conv_same(x, w, b, 1, padding)
    if i < num_convs - 1:
      x = F.leaky_relu(x, slope)
  return x


def recnet_forward(P, inp, kspace, mask, num_blocks, num_convs=3, slope=0.01,
                   prefix='conv_blocks', skip_final_dc=False,
                   return_intermediate=False):
  """RecNet cascade x = DC(convblock(x))                recnet.py:139-161."""
  x = inp
  recs = []
  n_dc = num_blocks - 1 if skip_final_dc else num_blocks
  for b in range(num_blocks):
    x = conv_block(P, '%s.%d' % (prefix, b), x, num_convs, slope)
    if b < n_dc:
      x = dc_layer(x, kspace, mask)
      recs.append(x)
  if return_intermediate:
    return x, recs
  return x


def init_recnet(num_blocks, num_convs, num_filters, gen=None, prefix='conv_blocks',
                slope=0.01):
  """Weight init of RecNet                 recnet.py:54-59, weight_inits.py:5-114.

  conv weights kaiming_normal(a=slope, fan_in); biases 0 -- except the first
  conv of each block: weight xavier_uniform(gain 1) and its bias keeps torch's
  default U(+-1/sqrt(fan_in)).  RNG consumption order is NOT the
  reference's (module construction order differs); fixtures carry weights.
  """
  P = {}
  for b in range(num_blocks):
    cin = 2
    for i in range(num_convs):
      cout = num_filters if i < num_convs - 1 else 2
      w = torch.empty(cout, cin, 3, 3)
      bias = torch.zeros(cout)
      if i == 0:
        torch.nn.init.xavier_uniform_(w, gain=1.0, generator=gen)
        bound = 1.0 / math.sqrt(cin * 9)
        bias.uniform_(-bound, bound, generator=gen)
      else:
        torch.nn.init.kaiming_normal_(w, a=slope, generator=gen)
      P['%s.%d.layers.%d.weight' % (prefix, b, 3 * i + 1)] = w
      P['%s.%d.layers.%d.bias' % (prefix, b, 3 * i + 1)] = bias
      cin = cout
  return P


# --------------------------------------------------------------------------
# U-Net (configured path only)                            models/unet.py:27-290
# --------------------------------------------------------------------------

UNET_CONF = dict(num_inputs=2, num_outputs=1, num_layers_per_scale=2,
                 encode_filters=[32, 64, 128], decode_filters=[64, 32],
                 kernel_size=4, slope=0.1)


def _bn(P, S, key, x, training, momentum=0.1, eps=1e-5):
  """nn.BatchNorm2d; P holds weight/bias, S holds running buffers."""
  return F.batch_norm(x, S[key + '.running_mean'], S[key + '.running_var'],
                      P[key + '.weight'], P[key + '.bias'], training,
                      momentum, eps)


def _encode_unit(P, S, prefix, x, num_layers, k, slope, training):
  """ConvEncodeUnit.encode: [pad, conv(no bias), BN, lrelu] x n   unet.py:27-58.

  Sequential indices: conv at 4*i+1, BN at 4*i+2."""
  for i in range(num_layers):
    w = P['%s.%d.weight' % (prefix, 4 * i + 1)]
    x = conv_same(x, w, None, 1, 'reflection')
    x = _bn(P, S, '%s.%d' % (prefix, 4 * i + 2), x, training)
    x = F.leaky_relu(x, slope)
  return x


def unet_forward(P, S, x, training=True, conf=UNET_CONF, prefix=''):
  """UNET.forward as configured by configs/2-refinement.json:31-47.

  reflection-padded 4x4 convs, BN, LeakyReLU(0.1), MaxPool2, nn-resize-conv
  upsampling with BN+act on the upsampling path only, 1x1 head.
  unet.py:261-290 (forward), :100-139 (decode unit).
  """
  k = conf['kernel_size']
  slope = conf['slope']
  nl = conf['num_layers_per_scale']
  enc = conf['encode_filters']
  skips = []
  for s in range(len(enc)):
    x = _encode_unit(P, S, '%sencode_units.%d.encode' % (prefix, s), x, nl, k,
                     slope, training)
    if s != len(enc) - 1:
      skips.append(x)
      x = F.max_pool2d(x, 2, 2)
  for s in range(len(enc) - 1):
    pre = '%sconcat_decode_units.%d' % (prefix, s)
    # upsample = [Upsample(nearest x2), pad, conv, BN, lrelu]   unet.py:100-118
    x = F.interpolate(x, scale_factor=2, mode='nearest')
    x = conv_same(x, P[pre + '.upsample.2.weight'], None, 1, 'reflection')
    x = _bn(P, S, pre + '.upsample.3', x, training)
    x = F.leaky_relu(x, slope)
    x = torch.cat((skips[-(s + 1)], x), dim=1)            # unet.py:137
    x = _encode_unit(P, S, pre + '.decode.0.encode', x, nl, k, slope, training)
  return F.conv2d(x, P[prefix + 'head.0.weight'], P[prefix + 'head.0.bias'])


def init_unet(conf=UNET_CONF, gen=None, prefix=''):
  """conv: orthogonal(gain=sqrt 2); BN weight 1, bias 0; head conv default
  (orthogonal too -- classname match 'Conv2d'), head bias 0.
  weight_inits.py:90-114 with configs/2-refinement.json:43-46."""
  P, S = {}, {}
  gain = math.sqrt(2.0)

  def conv(key, cout, cin, k):
    w = torch.empty(cout, cin, k, k)
    torch.nn.init.orthogonal_(w, gain=gain, generator=gen)
    P[key + '.weight'] = w

  def bn(key, c):
    P[key + '.weight'] = torch.ones(c)
    P[key + '.bias'] = torch.zeros(c)
    S[key + '.running_mean'] = torch.zeros(c)
    S[key + '.running_var'] = torch.ones(c)

  k = conf['kernel_size']
  nl = conf['num_layers_per_scale']
  enc = conf['encode_filters']
  dec = conf['decode_filters']
  cin = conf['num_inputs']
  for s, f in enumerate(enc):
    for i in range(nl):
      conv('%sencode_units.%d.encode.%d' % (prefix, s, 4 * i + 1), f, cin, k)
      bn('%sencode_units.%d.encode.%d' % (prefix, s, 4 * i + 2), f)
      cin = f
  for s, f in enumerate(dec[:len(enc) - 1]):
    pre = '%sconcat_decode_units.%d' % (prefix, s)
    conv(pre + '.upsample.2', f, cin, k)
    bn(pre + '.upsample.3', f)
    cin = f + enc[-(s + 2)]
    for i in range(nl):
      conv(pre + '.decode.0.encode.%d' % (4 * i + 1), f, cin, k)
      bn(pre + '.decode.0.encode.%d' % (4 * i + 2), f)
      cin = f
  conv(prefix + 'head.0', conf['num_outputs'], cin, 1)
  P[prefix + 'head.0.bias'] = torch.zeros(conf['num_outputs'])
  return P, S


# --------------------------------------------------------------------------
# RefinementWrapper                         models/refinement_wrapper.py:51-220
# --------------------------------------------------------------------------


def scale_minmax(t):
  """_scale: per (b,c) min, then max of the shifted tensor.  :51-73."""
  b, c, h, w = t.shape
  out = t.reshape(b, c, h * w)
  mn = out.min(dim=2, keepdim=True)[0]
  out = out - mn
  mx = out.max(dim=2, keepdim=True)[0]
  out = out / mx
  out = out * 2 - 1
  return out.reshape(b, c, h, w), mn, mx


def unscale_minmax(t, mn, mx):
  """_unscale.  :76-92."""
  b, c, h, w = t.shape
  out = t.reshape(b, c, h * w)
  out = (out + 1) / 2
  out = out * mx + mn
  return out.reshape(b, c, h, w)


def refinement_forward(P, S, inp, kspace, mask, training=True, num_blocks=3,
                       num_convs=3):
  """RefinementWrapper._forward_reconstruction with mode 'real-penalty-add',
  input_mode 'output', frozen pretrained RecNet.  :169-220."""
  with torch.no_grad():
    pre = recnet_forward(P, inp, kspace, mask, num_blocks, num_convs,
                         prefix='pretrained_model.conv_blocks')
  pre = pre.detach()
  pre_real = pre[:, 0:1].contiguous()
  pre_imag = pre[:, 1:2].contiguous()
  real_scaled, mn, mx = scale_minmax(pre_real)
  u = unet_forward(P, S, pre, training, prefix='learnable_model.')
  u_scaled = P['scale'] * u
  refined = real_scaled + u_scaled
  out_real = unscale_minmax(refined, mn, mx)
  return {
      'pred': torch.cat((out_real, pre_imag), dim=1),
      'pretrained': pre,
      'prescaled_refinement': u,
      'scaled_refinement': u_scaled,
  }


# --------------------------------------------------------------------------
# CNN discriminator                              models/discriminators.py:50-247
# --------------------------------------------------------------------------

DISC_CONF = dict(num_inputs=1, filters=[64, 128, 256, 512, 1024, 1024],
                 strides=[2, 2, 2, 2, 2, 1], kernel_size=4, final_kernel=4,
                 slope=0.2, dropout_after=[3, 4, 5], dropout_prob=0.5)


def disc_layer_indices(conf=DISC_CONF):
  """Sequential indices of conv / BN inside ``convs`` (discriminators.py:129-155):
  layer 0 = [pad, conv, act]; later layers [pad, conv, BN, act(, dropout)]."""
  idx = 0
  out = []
  for li in range(len(conf['filters'])):
    conv_i = idx + 1
    has_bn = li > 0          # use_norm_layers == 'not-first'
    bn_i = idx + 2 if has_bn else None
    idx += 4 if has_bn else 3
    if li in conf['dropout_after']:
      idx += 1
    out.append((conv_i, bn_i))
  return out


def disc_forward(P, S, x, training=True, conf=DISC_CONF, dropout_masks=None,
                 gen=None):
  """CNNDiscriminator.forward with compute_features (discriminators.py:211-234).

  features = 6 post-activation maps (post-DROPOUT for layers with in-place
  Dropout2d, SURVEY A-7) + logits.  dropout_masks: optional list of [B,C,1,1]
  tensors in {0, 1/(1-p)} (one per dropout layer, in order); drawn from ``gen``
  when None and training.
  """
  feats = []
  used_masks = []
  k = conf['kernel_size']
  di = 0
  for li, (ci, bi) in enumerate(disc_layer_indices(conf)):
    w = P['convs.%d.weight' % ci]
    b = P.get('convs.%d.bias' % ci)
    x = conv_same(x, w, b, conf['strides'][li], 'reflection')
    if bi is not None:
      x = _bn(P, S, 'convs.%d' % bi, x, training)
    x = F.leaky_relu(x, conf['slope'])
    if li in conf['dropout_after'] and training:
      p = conf['dropout_prob']
      if dropout_masks is not None:
        m = dropout_masks[di]
      else:
        keep = torch.bernoulli(torch.full((x.shape[0], x.shape[1], 1, 1),
                                          1 - p), generator=gen)
        m = keep / (1 - p)
      used_masks.append(m)
      x = x * m
      di += 1
    feats.append(x)
  logits = F.conv2d(x, P['final_conv.0.weight'], P['final_conv.0.bias'])
  feats.append(logits)
  return {'prob': torch.sigmoid(logits), 'logits': logits, 'features': feats,
          'dropout_masks': used_masks}


def init_disc(conf=DISC_CONF, gen=None):
  """conv N(0,0.02); BN weight N(1,0.02), bias 0; biases 0.
  discriminators.py:189-209, weight_inits.py:5-14."""
  P, S = {}, {}
  cin = conf['num_inputs']
  k = conf['kernel_size']
  for li, (ci, bi) in enumerate(disc_layer_indices(conf)):
    f = conf['filters'][li]
    P['convs.%d.weight' % ci] = torch.empty(f, cin, k, k).normal_(0.0, 0.02, generator=gen)
    if bi is None:
      P['convs.%d.bias' % ci] = torch.zeros(f)
    else:
      P['convs.%d.weight' % bi] = torch.empty(f).normal_(1.0, 0.02, generator=gen)
      P['convs.%d.bias' % bi] = torch.zeros(f)
      S['convs.%d.running_mean' % bi] = torch.zeros(f)
      S['convs.%d.running_var' % bi] = torch.ones(f)
    cin = f
  fk = conf['final_kernel']
  P['final_conv.0.weight'] = torch.empty(1, cin, fk, fk).normal_(0.0, 0.02, generator=gen)
  P['final_conv.0.bias'] = torch.zeros(1)
  return P, S


# --------------------------------------------------------------------------
# VGG19 perceptual loss                 models/vgg.py:8-80, vgg_loss.py:13-65
# --------------------------------------------------------------------------

VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M',
             512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']
VGG_MEAN = (0.485, 0.456, 0.406)
VGG_STD = (0.229, 0.224, 0.225)


def vgg_conv_indices(cfg=VGG19_CFG):
  """torchvision ``features`` indices of the conv layers (conv, relu, [pool])."""
  idx, out = 0, []
  for v in cfg:
    if v == 'M':
      idx += 1
    else:
      out.append(idx)
      idx += 2
  return out


def init_vgg(gen=None, cfg=VGG19_CFG):
  """Synthetic VGG weights: torchvision's non-pretrained init
  kaiming_normal_(fan_out, relu), bias 0 (SURVEY A-9).  Keys follow the
  reference module: blocks.{b}.{features_idx}.weight (vgg.py:36-44)."""
  P = {}
  cin, idx, block = 3, 0, 0
  for v in cfg:
    if v == 'M':
      idx += 1
      block += 1
      continue
    w = torch.empty(v, cin, 3, 3)
    torch.nn.init.kaiming_normal_(w, mode='fan_out', nonlinearity='relu',
                                  generator=gen)
    P['blocks.%d.%d.weight' % (block, idx)] = w
    P['blocks.%d.%d.bias' % (block, idx)] = torch.zeros(v)
    cin = v
    idx += 2
  return P


def vgg_features(P, x, last_block=4, cfg=VGG19_CFG):
  """VGG19.forward up to the end of block ``last_block`` (relu5_4 for 4).
  Block b starts with the b-th max-pool (vgg.py:36-44,58-80)."""
  mean = torch.tensor(VGG_MEAN, dtype=x.dtype).view(1, 3, 1, 1)
  std = torch.tensor(VGG_STD, dtype=x.dtype).view(1, 3, 1, 1)
  x = (x - mean) / std
  idx, block = 0, 0
  for v in cfg:
    if v == 'M':
      if block == last_block:
        break
      block += 1
      x = F.max_pool2d(x, 2, 2)
      idx += 1
      continue
    x = F.relu(F.conv2d(x, P['blocks.%d.%d.weight' % (block, idx)],
                        P['blocks.%d.%d.bias' % (block, idx)], padding=1))
    idx += 2
  return x


def complex_abs(t):
  """sqrt(re^2+im^2), [B,2,H,W]->[B,1,H,W]       utils/tensor_transforms.py:62-75."""
  return ((t[:, 0] ** 2 + t[:, 1] ** 2) ** 0.5).unsqueeze(1)


def vgg_loss(PV, pred, target):
  """VGGLoss.forward for complex inputs, MSE on relu5_4.  vgg_loss.py:43-65 with
  criteria.py:15-28 (default criterion 'MSE', no range normalisation)."""
  p = complex_abs(pred)
  p = torch.cat((p, p, p), dim=1)
  t = complex_abs(target.detach())
  t = torch.cat((t, t, t), dim=1)
  return F.mse_loss(vgg_features(PV, p), vgg_features(PV, t).detach())


# --------------------------------------------------------------------------
# adversarial losses                          models/adversarial_loss.py:27-160
# --------------------------------------------------------------------------


def gan_loss_disc(out_fake, out_real, label_smoothing=0.1):
  """BCE(prob_fake, 0) + BCE(prob_real, 1 - smoothing).  :71-98."""
  pf, pr = out_fake['prob'], out_real['prob']
  return (F.binary_cross_entropy(pf, torch.zeros_like(pf)) +
          F.binary_cross_entropy(pr, torch.full_like(pr, 1.0 - label_smoothing)))


def gan_loss_gen(out_fake):
  """BCE(prob_fake, 1).  :87-98."""
  pf = out_fake['prob']
  return F.binary_cross_entropy(pf, torch.ones_like(pf))


def feature_matching_loss(out_fake, out_real):
  """mean_i L1(f_fake_i, f_real_i.detach()).  :152-160."""
  ff, fr = out_fake['features'], out_real['features']
  loss = 0
  for a, b in zip(ff, fr):
    loss = loss + F.l1_loss(a, b.detach())
  return loss / len(ff)


def feature_penalty(out_gen):
  """L1(prescaled_refinement, 0).  criteria.py:31-47,86-109."""
  u = out_gen['prescaled_refinement']
  return F.l1_loss(u, torch.zeros_like(u))


# --------------------------------------------------------------------------
# metric                 metrics/image_metrics.py:7-19, rec_transforms.py:79-85
# --------------------------------------------------------------------------


def psnr_batch(pred, target):
  """MetricFunction(psnr) on a batch: per image clamp(|.|,0,1), 10 log10(1/mse),
  mean over images (metrics/__init__.py:38-72)."""
  p = torch.clamp(complex_abs(pred), 0.0, 1.0)
  t = torch.clamp(complex_abs(target), 0.0, 1.0)
  vals = []
  for i in range(p.shape[0]):
    mse = F.mse_loss(p[i:i + 1], t[i:i + 1]).item()
    vals.append(10.0 * np.log10(1.0 / mse))
  return float(np.mean(vals))


def ssim_window(window_size=11, sigma=1.5):
  """Normalised 1-D gaussian and its outer product (metrics/pytorch_ssim/__init__.py:12-20)."""
  g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2))
                    for x in range(window_size)], dtype=torch.float32)
  g = g / g.sum()
  return g, g.unsqueeze(1).mm(g.unsqueeze(0)).float()


def ssim_images(pred, target, window_size=11):
  """MetricFunction(ssim) per image: clamp(|.|,0,1) (rec_transforms.py:79-85), then
  pytorch_ssim.ssim (metrics/pytorch_ssim/__init__.py:22-42, image_metrics.py:22-42): gaussian
  11x11 window, zero padding, C1 = 0.01^2, C2 = 0.03^2, mean of the SSIM map.  Returns the
  list of per-image values (their mean is the batch metric, metrics/__init__.py:38-72)."""
  p = torch.clamp(complex_abs(pred), 0.0, 1.0)
  t = torch.clamp(complex_abs(target), 0.0, 1.0)
  _, w2 = ssim_window(window_size)
  w = w2[None, None]
  pad = window_size // 2
  vals = []
  for i in range(p.shape[0]):
    a, b = p[i:i + 1], t[i:i + 1]
    mu1, mu2 = F.conv2d(a, w, padding=pad), F.conv2d(b, w, padding=pad)
    mu1_sq, mu2_sq, mu12 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(a * a, w, padding=pad) - mu1_sq
    s2 = F.conv2d(b * b, w, padding=pad) - mu2_sq
    s12 = F.conv2d(a * b, w, padding=pad) - mu12
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu12 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))
    vals.append(float(m.mean()))
  return vals


def binary_accuracy_fake(prob_fake):
  """disc metric 'binary_accuracy' = accuracy on the fake batch:
  per-image mean prob, class = prob > .5, target 0.  scalar_metrics.py:11-53."""
  p = prob_fake.reshape(prob_fake.shape[0], -1).mean(dim=1)
  return float(((p > 0.5) == torch.zeros_like(p, dtype=torch.bool)).float().mean())


# --------------------------------------------------------------------------
# image pool                                           utils/image_pool.py:8-60
# --------------------------------------------------------------------------


class ImagePool(object):
  """History of generated images; python ``random`` drives the decisions.
  ``decisions`` (list of (use_pool: bool, idx: int)) can be injected."""

  def __init__(self, pool_size, p=0.5):
    self.pool_size, self.p, self.images = pool_size, p, []

  def query(self, batch, decisions=None):
    if self.pool_size == 0:
      return batch
    out = []
    for i in range(batch.shape[0]):
      img = batch[i:i + 1].detach()
      if len(self.images) < self.pool_size:
        self.images.append(img)
        out.append(img)
      else:
        if decisions is not None:
          use, idx = decisions[i]
        else:
          use = _pyrandom.uniform(0, 1) < self.p
          idx = _pyrandom.randint(0, self.pool_size - 1) if use else -1
        if use:
          out.append(self.images[idx].clone())
          self.images[idx] = img
        else:
          out.append(img)
    return torch.cat(out, 0)


# --------------------------------------------------------------------------
# training steps     training/runner.py:154-178, adversarial_runner.py:314-389
# --------------------------------------------------------------------------


def philox4x32_10(counter, key):
  """Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the generator behind
  torch's / curand's Philox streams): counter uint32 [n,4], key uint32 [2] -> uint32 [n,4].  Restated in numpy for
  the product's Dropout2d mask kernel (csmri_dropout2d_mask, include/csmri_hip.h), which the reference draws with
  torch's RNG in nn.Dropout2d (models/discriminators.py:150-152)."""
  import numpy as np
  c = np.array(counter, dtype=np.uint64).reshape(-1, 4).copy()
  k0, k1 = np.uint64(int(key[0])), np.uint64(int(key[1]))
  M0, M1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
  for _ in range(10):
    p0, p1 = M0 * c[:, 0], M1 * c[:, 2]
    hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
    c = np.stack([hi1 ^ c[:, 1] ^ k0, lo1, hi0 ^ c[:, 3] ^ k1, lo0], 1)
    k0, k1 = (k0 + np.uint64(0x9E3779B9)) & MASK, (k1 + np.uint64(0xBB67AE85)) & MASK
  return c.astype(np.uint32)


def dropout2d_mask(seed, call, n, p):
  """The n mask values csmri_dropout2d_mask writes for state (seed, call): keep / (1 - p) with
  keep = [(philox >> 8) * 2^-24 < 1 - p], element i from counter (i // 4, 0, call_lo, call_hi), lane i % 4."""
  import numpy as np
  g = (n + 3) // 4
  idx = np.arange(g, dtype=np.uint64)
  ctr = np.stack([idx & np.uint64(0xFFFFFFFF), idx >> np.uint64(32),
                  np.full(g, call & 0xFFFFFFFF, dtype=np.uint64), np.full(g, call >> 32, dtype=np.uint64)], 1)
  r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, seed >> 32)).reshape(-1)[:n]
  u = (r >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
  return torch.from_numpy(np.where(u < np.float32(1.0 - p), np.float32(1.0 / (1.0 - p)), np.float32(0.0)))


def make_adam(params, lr=2e-4, beta1=0.9, beta2=0.999):
  """training/optimizers.py:19-22 -> torch.optim.Adam (eps 1e-8)."""
  return torch.optim.Adam(list(params), lr, betas=(beta1, beta2))


def recnet_mse_step(P, opt, batch, num_blocks, num_convs=3):
  """Runner._train_step: zero_grad, forward, MSE, backward, Adam.
  runner.py:154-178.  P tensors must be leaf tensors with requires_grad."""
  opt.zero_grad()
  pred = recnet_forward(P, batch['inp'], batch['kspace'], batch['mask'],
                        num_blocks, num_convs)
  loss = F.mse_loss(pred, batch['target'])
  loss.backward()
  opt.step()
  return {'loss_MSE': loss.item(), 'loss': loss.item(),
          'psnr': psnr_batch(pred.detach(), batch['target'])}, pred.detach()


GEN_LOSS_WEIGHTS = dict(gan=0.5, FeatureMatching=1.0, VGG19=10.0,
                        FeaturePenalty=2.0)


def _step_keep_versions(opt):
  """optimizer.step() with torch-0.3.1 semantics: the in-place parameter update
  is invisible to autograd's version check (SURVEY A-4, App. B-6)."""
  ps = [p for g in opt.param_groups for p in g['params']]
  vs = [p._version for p in ps]
  opt.step()
  torch._C._autograd._unsafe_set_version_counter(ps, vs)


def gan_train_step(PG, SG, PD, SD, PV, gen_opt, disc_opt, batch, pool=None,
                   dropout_masks=None, pool_decisions=None, faithful=True,
                   weights=GEN_LOSS_WEIGHTS, label_smoothing=0.1, gen=None):
  """AdversarialRunner._train_single_step.  adversarial_runner.py:322-389.

  PG/PD: trainable leaf tensors (+ frozen pretrained_model.* in PG);
  SG/SD: BN running buffers; PV: VGG weights.
  dropout_masks: optional [3 passes][3 layers] injected masks.
  faithful=True reproduces the reference ordering: the generator backward runs
  through the discriminator AFTER its Adam step (post-update weights, pre-update
  saved activations/statistics, SURVEY A-4 semantics (A)); False defers the D
  step (semantics (B)).
  """
  dm = dropout_masks or [None, None, None]
  out_gen = refinement_forward(PG, SG, batch['inp'], batch['kspace'],
                               batch['mask'], True)
  # -- discriminator phase ------------------------------------------- :331-347
  fake_in = complex_abs(out_gen['pred']).detach()
  if pool is not None:
    fake_in = pool.query(fake_in, pool_decisions)
  out_fake_d = disc_forward(PD, SD, fake_in, True, dropout_masks=dm[0], gen=gen)
  real_in = complex_abs(batch['target']).detach()
  out_real = disc_forward(PD, SD, real_in, True, dropout_masks=dm[1], gen=gen)
  l_disc_gan = gan_loss_disc(out_fake_d, out_real, label_smoothing)
  # -- generator phase ----------------------------------------------- :349-370
  out_fake = disc_forward(PD, SD, complex_abs(out_gen['pred']), True,
                          dropout_masks=dm[2], gen=gen)
  l_gan = gan_loss_gen(out_fake)
  l_fm = feature_matching_loss(out_fake, out_real)
  l_vgg = vgg_loss(PV, out_gen['pred'], batch['target'])
  l_fp = feature_penalty(out_gen)
  total_disc = 1.0 * l_disc_gan
  total_gen = (weights['gan'] * l_gan + weights['FeatureMatching'] * l_fm +
               weights['VGG19'] * l_vgg + weights['FeaturePenalty'] * l_fp)
  # -- updates ------------------------------------------------------- :372-383
  disc_opt.zero_grad()
  total_disc.backward(retain_graph=False)
  if faithful:
    _step_keep_versions(disc_opt)
  gen_opt.zero_grad()
  d_params = [p for g in disc_opt.param_groups for p in g['params']]
  saved = [p.grad.clone() if p.grad is not None else None for p in d_params]
  total_gen.backward()
  # the generator backward also deposits (unused) grads in D (A-5): drop them
  for p, g in zip(d_params, saved):
    p.grad = g
  if not faithful:
    disc_opt.step()
  gen_opt.step()
  losses = {
      'disc_loss_gan': float(l_disc_gan), 'gen_loss_gan': float(l_gan),
      'gen_loss_FeatureMatching': float(l_fm), 'gen_loss_VGG19': float(l_vgg),
      'gen_loss_FeaturePenalty': float(l_fp), 'disc_loss': float(total_disc),
      'gen_loss': float(total_gen),
  }
  metrics = {
      'gen_psnr': psnr_batch(out_gen['pred'].detach(), batch['target']),
      'disc_binary_accuracy': binary_accuracy_fake(out_fake.get('prob').detach()),
  }
  return losses, metrics, out_gen


def data_parallel_gan_step(replicas, PV, shards, dropout_masks=None):
  """SURVEY 8(e) multi-GPU parity oracle: N model replicas in lock step, each running
  gan_train_step (the reference's AdversarialRunner step) on ITS shard of the global batch with its own
  BatchNorm statistics, dropout draws and image pool; at every optimizer.step() the replicas' gradients are
  first replaced by their mean -- what the gradient all-reduce / N does between the backward and Adam.
  (The reference itself only has single-process nn.DataParallel, utils/custom_data_parallel.py:26-35,
  whose replicas likewise keep per-replica BatchNorm batch statistics.)

  replicas: list of dicts with keys PG, SG, PD, SD, gen_opt, disc_opt, pool (one per rank, same initial
  weights); shards: per-rank batch dicts; dropout_masks: per-rank [3 passes][3 layers] or None.
  Returns the per-rank (losses, metrics) and the averaged gradients {'G': {...}, 'D': {...}}."""
  import threading
  n = len(replicas)
  bar = threading.Barrier(n)
  avg = {}

  def wrap(key, tag, names_of):
    opts = [r[key] for r in replicas]
    for rank, opt in enumerate(opts):
      orig = opt.step

      def step(orig=orig, rank=rank, opts=opts, tag=tag):
        bar.wait()
        if rank == 0:
          plist = [[p for g in o.param_groups for p in g['params']] for o in opts]
          mean = []
          for group in zip(*plist):
            m = sum(p.grad for p in group) / n
            for p in group:
              p.grad = m.clone()
            mean.append(m)
          avg[tag] = {k: m for k, m in zip(names_of(replicas[0]), mean)}
        bar.wait()
        orig()
      opt.step = step
  wrap('disc_opt', 'D', lambda r: list(r['PD'].keys()))
  wrap('gen_opt', 'G', lambda r: [k for k, v in r['PG'].items() if v.requires_grad])
  out, err = [None] * n, []

  def run(rank):
    try:
      r = replicas[rank]
      dm = dropout_masks[rank] if dropout_masks is not None else None
      l, m, _ = gan_train_step(r['PG'], r['SG'], r['PD'], r['SD'], PV, r['gen_opt'], r['disc_opt'],
                               shards[rank], pool=r.get('pool'), dropout_masks=dm)
      out[rank] = (l, m)
    except BaseException as e:       # a failed replica must not leave the others at the barrier
      err.append(e)
      bar.abort()
  threads = [threading.Thread(target=run, args=(i,)) for i in range(n)]
  for t in threads:
    t.start()
  for t in threads:
    t.join()
  if err:
    raise err[0]
  return out, avg


def lr_multistep(base_lr, milestones, gamma, epoch):
  """MultiStepLR as built by training/lr_schedulers.py:26-31 (``multistep``): the value in force while
  the scheduler's epoch counter is ``epoch`` (the runner steps it once per epoch_beginning, so training
  epoch e runs at epoch index e; index 0 is the construction-time value)."""
  return base_lr * gamma ** sum(1 for m in milestones if m <= epoch)


def lr_polynomial(base_lr, end_lr, decay_epochs, epoch, from_epoch=0, power=1.0):
  """LambdaLR with _get_polynomial_decay, training/lr_schedulers.py:4-15,32-42 (``linear`` is
  power 1)."""
  if epoch < from_epoch:
    return base_lr
  end_epoch = float(from_epoch + decay_epochs)
  e = min(epoch, end_epoch)
  return (base_lr - end_lr) * (1.0 - e / end_epoch) ** power + end_lr


def pretraining_flags(epoch, gen_schedule=None, disc_schedule=None):
  """(discriminator_enabled, generator_enabled) for an epoch: AdversarialRunner.epoch_beginning,
  adversarial_runner.py:273-298, with _get_pretraining_schedule (:195-209): an int n means epochs
  1..n, a pair [a, b) is used as is, None never applies.  The generator schedule is evaluated first
  and the discriminator schedule may re-enable the discriminator it switched off."""
  def sched(e):
    if e is None:
      return (-1, -1)
    if isinstance(e, int):
      return (1, e + 1)
    return tuple(e)
  disc_en = gen_en = True
  start, end = sched(gen_schedule)
  if start <= epoch < end:
    disc_en, gen_en = False, True
  else:
    disc_en = True
  start, end = sched(disc_schedule)
  if start <= epoch < end:
    disc_en, gen_en = True, False
  else:
    gen_en = True
  return disc_en, gen_en


def gan_train_multi_step(PG, SG, PD, SD, PV, gen_opt, disc_opt, batches, disc_updates=1,
                         gen_updates=1, disc_enabled=True, gen_enabled=True, pool=None,
                         dropout_masks=None, weights=GEN_LOSS_WEIGHTS, label_smoothing=0.1):
  """AdversarialRunner._train_multiple_steps, adversarial_runner.py:391-525.

  ``batches``: the max(disc_updates, gen_updates) batches drawn up front.  The discriminator is
  updated on batches[:disc_updates] (generator forward, D(fake.detach via pool), D(real), loss,
  backward, step), then the generator on batches[:gen_updates] (generator forward, D(fake) through the
  UPDATED discriminator, D(real) again only for FeatureMatching, all generator losses, backward,
  step).  dropout_masks: one [3 layers] list per discriminator forward, in call order.  Returns the
  per-name averages of the loss values over the updates (reference :509-511) and the last out_gen."""
  masks = list(dropout_masks) if dropout_masks is not None else None

  def next_masks():
    return masks.pop(0) if masks is not None else None
  sums, counts = {}, {}

  def acc(name, v):
    sums[name] = sums.get(name, 0.0) + float(v.detach() if torch.is_tensor(v) else v)
    counts[name] = counts.get(name, 0) + 1
  out_gen = None
  d_params = [p for g in disc_opt.param_groups for p in g['params']]
  if disc_enabled:
    for batch in batches[:disc_updates]:
      out_gen = refinement_forward(PG, SG, batch['inp'], batch['kspace'], batch['mask'], True)
      fake_in = complex_abs(out_gen['pred']).detach()
      if pool is not None:
        fake_in = pool.query(fake_in, None)
      out_fake = disc_forward(PD, SD, fake_in, True, dropout_masks=next_masks())
      out_real = disc_forward(PD, SD, complex_abs(batch['target']).detach(), True, dropout_masks=next_masks())
      l = gan_loss_disc(out_fake, out_real, label_smoothing)
      acc('disc_loss_gan', l)
      disc_opt.zero_grad()
      (1.0 * l).backward()
      disc_opt.step()
      acc('disc_loss', l)
  if gen_enabled:
    for batch in batches[:gen_updates]:
      out_gen = refinement_forward(PG, SG, batch['inp'], batch['kspace'], batch['mask'], True)
      total = 0.0
      if disc_enabled:
        out_fake = disc_forward(PD, SD, complex_abs(out_gen['pred']), True, dropout_masks=next_masks())
        out_real = disc_forward(PD, SD, complex_abs(batch['target']).detach(), True, dropout_masks=next_masks())
        l_gan, l_fm = gan_loss_gen(out_fake), feature_matching_loss(out_fake, out_real)
        acc('gen_loss_gan', l_gan)
        acc('gen_loss_FeatureMatching', l_fm)
        total = weights['gan'] * l_gan + weights['FeatureMatching'] * l_fm
      l_vgg, l_fp = vgg_loss(PV, out_gen['pred'], batch['target']), feature_penalty(out_gen)
      acc('gen_loss_VGG19', l_vgg)
      acc('gen_loss_FeaturePenalty', l_fp)
      total = total + weights['VGG19'] * l_vgg + weights['FeaturePenalty'] * l_fp
      gen_opt.zero_grad()
      saved = [p.grad.clone() if p.grad is not None else None for p in d_params]
      total.backward()
      for p, g in zip(d_params, saved):      # gradients deposited in D by the generator backward are unused (A-5)
        p.grad = g
      gen_opt.step()
      acc('gen_loss', total)
  return {k: sums[k] / counts[k] for k in sums}, out_gen


# --------------------------------------------------------------------------
# synthetic inputs      compressed_sensing.py:82-123,460-512; dnn_io.py:4-61;
#                       myImageTransformations.py:1196-1238; rec_transforms.py:47
# --------------------------------------------------------------------------


def _normal_pdf(length, sensitivity):
  """compressed_sensing.py:13-14."""
  return np.exp(-sensitivity * (np.arange(length) - length / 2) ** 2)


def cartesian_mask_rows(nx, acc, sample_n=8, rng=None):
  """Row indices (un-shifted, i.e. centred) selected by cartesian_mask for ONE
  slice, and the pdf, restating compressed_sensing.py:82-123.  Returns the 0/1
  row vector of length nx BEFORE ifftshift."""
  rng = np.random if rng is None else rng
  pdf_x = _normal_pdf(nx, 0.5 / (nx / 10.) ** 2)
  lmda = nx / (2. * acc)
  n_lines = nx // acc
  pdf_x += lmda * 1. / nx
  if sample_n:
    pdf_x[nx // 2 - sample_n // 2:nx // 2 + sample_n // 2] = 0
    pdf_x /= np.sum(pdf_x)
    n_lines -= sample_n
  rows = np.zeros(nx)
  idx = rng.choice(nx, int(n_lines), False, pdf_x)
  rows[idx] = 1
  if sample_n:
    rows[nx // 2 - sample_n // 2:nx // 2 + sample_n // 2] = 1
  return rows


def cartesian_mask(shape, acc, sample_n=8, rng=None):
  """cartesian_mask(shape=(N,nx,ny), centred=False): constant along ny,
  ifftshift-ed along both axes.  compressed_sensing.py:82-123."""
  n, nx, ny = shape
  m = np.zeros((n, nx, ny))
  for i in range(n):
    m[i] = cartesian_mask_rows(nx, acc, sample_n, rng)[:, None]
  return np.fft.ifftshift(m, axes=(-1, -2))


def radial_mask(shape, n_lines, angle_begin=0.0, rand=False, golden_angle=False, centred=False, rng=None):
  """radial_sampling (compressed_sensing.py:568-647) for square slices: spokes gridded to the
  nearest Cartesian sample, un-centred by default (ifftshift) as the training transform uses it
  (BASELINE config 5: `acceleration_factor` = number of spokes, myImageTransformations.py:63-70).
  Integer 0/1 mask of ``shape`` = (..., nx, ny)."""
  rng = np.random if rng is None else rng
  golden = np.pi / ((1 + np.sqrt(5)) / 2)
  n, nx, ny = int(np.prod(shape[:-2])), shape[-2], shape[-1]
  assert nx == ny, 'square slices only (the reference pads otherwise)'
  mask = np.zeros((n, nx, ny), dtype=int)
  if rand:
    angle_begin = np.pi * rng.random()
  y = np.arange(-nx / 2, nx / 2, 1)
  x = np.arange(-ny / 2, ny / 2, 1)
  if golden_angle:
    angles = [angle_begin + i * golden for i in range(n_lines * n)]
  else:
    angles = np.tile(np.arange(0, np.pi, np.pi / n_lines), n)
    angles = angles + np.repeat(rng.random(n) * np.pi / n_lines, n_lines)
  kloc = np.outer(y, np.cos(angles)) + 1j * np.outer(x, np.sin(angles))
  k1 = np.round(kloc + (0.5 + 0.5j)) + ((nx / 2) + (ny / 2) * 1j)
  re, im = np.real(k1), np.imag(k1)
  re = re - nx * (re > nx)
  im = im - ny * (im > ny)
  re = re + nx * (re < 1)
  im = im + ny * (im < 1)
  t = np.repeat(np.arange(n), n_lines * nx)
  xi = (re.transpose().reshape(-1) - 1).astype(int)
  yi = (im.transpose().reshape(-1) - 1).astype(int)
  mask[t, xi, yi] = 1
  if not centred:
    mask = np.fft.ifftshift(mask, axes=(-2, -1))
  return mask.reshape(shape)


def phantom(h, w, seed):
  """Seeded band-limited random phantom in [0,1] with a few ellipses, divided
  by its max (mirrors x/np.max(np.abs(x)), rec_transforms.py:47).  Strictly
  positive so |.| has no exact zeros (SURVEY A-6)."""
  rs = np.random.RandomState(seed)
  f = np.fft.fft2(rs.rand(h, w))
  fy = np.fft.fftfreq(h)[:, None]
  fx = np.fft.fftfreq(w)[None, :]
  img = np.real(np.fft.ifft2(f * np.exp(-(fy ** 2 + fx ** 2) * (h * 0.35) ** 2)))
  img = (img - img.min()) / (img.max() - img.min() + 1e-12)
  yy, xx = np.mgrid[0:h, 0:w]
  for _ in range(4):
    cy, cx = rs.uniform(0.25, 0.75) * h, rs.uniform(0.25, 0.75) * w
    ry, rx = rs.uniform(0.05, 0.25) * h, rs.uniform(0.05, 0.25) * w
    img = img + rs.uniform(0.2, 0.8) * (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1)
  img = img + 0.02
  return img / np.max(np.abs(img))


def synth_sample(h, w, acc, seed, sample_n=8):
  """One sample: (inp, kspace, mask, target) float32 [2,H,W] each, following
  Undersample (myImageTransformations.py:1196-1238) + undersample
  (compressed_sensing.py:460-512) + complex2real/mask packing (dnn_io.py:4-61)."""
  img = phantom(h, w, seed)
  rng = np.random.RandomState(seed + 7919)
  mask = cartesian_mask((1, h, w), acc, sample_n, rng)[0]
  k_full = np.fft.fft2(img.astype(np.complex128), norm='ortho')
  k_u = mask * k_full
  x_u = np.fft.ifft2(k_u, norm='ortho')
  c2r = lambda z: np.stack((np.real(z), np.imag(z))).astype(np.float32)
  return (c2r(x_u), c2r(k_u), c2r(mask * (1 + 1j)),
          c2r(img.astype(np.complex128)))


def synth_batch(b, h, w, acc=4, seed=0, sample_n=8):
  """Batch dict with the reference's keys (scar_segmentation.py:212-218)."""
  parts = [synth_sample(h, w, acc, seed + 1000 + i, sample_n) for i in range(b)]
  keys = ('inp', 'kspace', 'mask', 'target')
  return {k: torch.from_numpy(np.stack([p[j] for p in parts]))
          for j, k in enumerate(keys)}
