#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference).  It imports the
reference's own modules (torch-0.3.1-era code) under small compatibility shims
 and records inputs and outputs as arrays.  Nothing of
the reference's source is stored -- only data.

  python tests/golden/make_golden.py            # writes tests/golden/F*.npz

Fixtures: F1 synthetic batch + mask rows, F2 DC fwd/adjoint,
F3 RecNet fwd/loss/grads/Adam steps, F4 RefinementWrapper, F5 discriminator,
F6 VGG loss, F7 full GAN train steps, F8 PSNR, F9 SSIM, F10 radial masks, F11 multi-update steps + pretraining schedule + LR schedulers, F12 weight initialisation digests,
F13 a checkpoint written by the reference's save_checkpoint + what the reference computes after it.
"""
import collections
import collections.abc
import os
import sys
import types
import warnings

warnings.filterwarnings('ignore')
sys.dont_write_bytecode = True
REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REF)
sys.path.insert(1, os.path.join(ROOT, 'oracle'))

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

collections.Iterable = collections.abc.Iterable

# ---------------------------------------------------------------- shims ----


def _install_shims():
  # pytorch_fft (CUDA-only, not vendored): split-plane FFTs via torch.fft.
  pf = types.ModuleType('pytorch_fft')
  pff = types.ModuleType('pytorch_fft.fft')

  def mk(fn, nd):
    def f(re, im):
      out = fn(torch.complex(re, im), dim=tuple(range(-nd, 0)))
      return out.real.contiguous(), out.imag.contiguous()
    return f

  for name, fn, nd in (('fft', torch.fft.fftn, 1), ('ifft', torch.fft.ifftn, 1),
                       ('fft2', torch.fft.fftn, 2), ('ifft2', torch.fft.ifftn, 2),
                       ('fft3', torch.fft.fftn, 3), ('ifft3', torch.fft.ifftn, 3)):
    setattr(pff, name, mk(fn, nd))
  for name in ('rfft', 'irfft', 'rfft2', 'irfft2', 'rfft3', 'irfft3'):
    setattr(pff, name, None)
  pf.fft = pff
  sys.modules['pytorch_fft'] = pf
  sys.modules['pytorch_fft.fft'] = pff

  # torchvision: only the VGG19 cfg-E feature stack + trivial transforms.
  tv = types.ModuleType('torchvision')
  tvm = types.ModuleType('torchvision.models')
  tvt = types.ModuleType('torchvision.transforms')
  tvu = types.ModuleType('torchvision.utils')
  cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M',
         512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']

  def vgg19(pretrained=False):
    layers, c = [], 3
    for v in cfg:
      if v == 'M':
        layers.append(nn.MaxPool2d(2, 2))
      else:
        layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU(inplace=True)]
        c = v
    m = nn.Module()
    m.features = nn.Sequential(*layers)
    return m

  tvm.vgg19 = vgg19

  class Compose(object):
    def __init__(self, t):
      self.transforms = t

    def __call__(self, x):
      for t in self.transforms:
        x = t(x)
      return x

  class Lambda(object):
    def __init__(self, f):
      self.f = f

    def __call__(self, x):
      return self.f(x)

  tvt.Compose, tvt.Lambda = Compose, Lambda
  tvu.make_grid = tvu.save_image = None
  tv.models, tv.transforms, tv.utils = tvm, tvt, tvu
  for n, m in (('torchvision', tv), ('torchvision.models', tvm),
               ('torchvision.transforms', tvt), ('torchvision.utils', tvu)):
    sys.modules[n] = m
  cv2 = types.ModuleType('cv2')
  cv2.BORDER_CONSTANT = 0
  sys.modules['cv2'] = cv2
  sys.modules['nibabel'] = types.ModuleType('nibabel')

  # legacy instance-style autograd Functions -> differentiable callables
  import data.reconstruction.deep_med_lib.my_pytorch.myfft as myfft

  class _F2(object):
    def __init__(self, norm, inv):
      self.norm, self.inv = norm, inv

    def __call__(self, re, im):
      f = torch.fft.ifft2 if self.inv else torch.fft.fft2
      o = f(torch.complex(re, im),
            norm='ortho' if self.norm == 'ortho' else 'backward')
      return o.real, o.imag

  myfft.Fft2d = lambda norm=None: _F2(norm, False)
  myfft.Ifft2d = lambda norm=None: _F2(norm, True)


_install_shims()

import utils as ref_utils                                      # noqa: E402
from utils.config import Configuration                         # noqa: E402
import models.criteria as ref_criteria                         # noqa: E402
import metrics as ref_metrics                                  # noqa: E402
import metrics.image_metrics as ref_image_metrics              # noqa: E402
import training.adversarial_runner as ref_AR                   # noqa: E402
import training.runner as ref_RR                               # noqa: E402
import data.reconstruction.deep_med_lib.utils.compressed_sensing as ref_cs  # noqa
import data.reconstruction.deep_med_lib.my_pytorch.myfft as ref_myfft       # noqa
from models import construct_model                             # noqa: E402

import csmri_oracle as O                                       # noqa: E402

# torch-0.3.1 API rot (SURVEY App. B 5,6): losses are shape (1,), .data[0]
_orig_get_criterion = ref_criteria.get_criterion


class _Shape1(nn.Module):
  def __init__(self, c):
    super(_Shape1, self).__init__()
    self.c = c

  def forward(self, *a, **k):
    return self.c(*a, **k).reshape(1)


def _get_criterion(*a, **k):
  return _Shape1(_orig_get_criterion(*a, **k))


ref_criteria.get_criterion = _get_criterion
ref_AR.get_criterion = _get_criterion
ref_RR.get_criterion = _get_criterion
_glm = ref_metrics.get_loss_metric
ref_AR.get_loss_metric = lambda v: _glm(float(v))
ref_RR.get_loss_metric = ref_AR.get_loss_metric


def _psnr(p, t):
  mse = F.mse_loss(p, t).item()
  return 10. * np.log10(1. / mse)


ref_image_metrics.compute_psnr = _psnr


import metrics.scalar_metrics as ref_scalar_metrics            # noqa: E402
_orig_binary_accuracy = ref_scalar_metrics.binary_accuracy
ref_scalar_metrics.binary_accuracy = lambda p, t: float(_orig_binary_accuracy(p, t))


class _TorchSum1(object):
  """`torch` as seen by training/runner.py: torch.sum(...) keeps shape (1,) so
  that the 0.3.1 idiom `total_loss.data[0]` (runner.py:175) still works."""

  def __getattr__(self, name):
    return getattr(torch, name)

  @staticmethod
  def sum(*a, **k):
    return torch.sum(*a, **k).reshape(1)


ref_RR.torch = _TorchSum1()


def _update_step_keep_versions(optimizer, losses, weights):
  total = torch.sum(torch.cat(losses) * weights)
  optimizer.zero_grad()
  total.backward()
  ps = [p for g in optimizer.param_groups for p in g['params']]
  vs = [p._version for p in ps]
  optimizer.step()
  torch._C._autograd._unsafe_set_version_counter(ps, vs)
  return total.reshape(1)


ref_AR.AdversarialRunner._update_step = staticmethod(_update_step_keep_versions)


class _Loader(list):
  batch_size = 2


def npd(d):
  return {k: v.detach().cpu().numpy().copy() for k, v in d.items()}


def save(name, **arrs):
  path = os.path.join(HERE, name + '.npz')
  np.savez_compressed(path, **arrs)
  print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024.))


# ------------------------------------------------------------------ F1 ----


def f1():
  """Synthetic batch: the reference's numpy forward model on our phantom."""
  out = {}
  for tag, (h, acc) in {'64': (64, 4), '256': (256, 4)}.items():
    rng = np.random.RandomState(1234)
    m = ref_cs.cartesian_mask((2, h, h), acc, sample_n=8, centred=False, rng=rng)
    out['mask_' + tag] = m[:, :, 0].astype(np.uint8)   # constant along ny
    out['rows_' + tag] = np.stack([np.nonzero(m[i, :, 0])[0] for i in range(2)])
    img = np.stack([O.phantom(h, h, 5 + i) for i in range(2)])
    xu, xfu = ref_cs.undersample(img, m, centred=False, norm='ortho')
    # phantom is regenerated from its seed by the test; keep outputs small
    if h == 64:
      out['xu_' + tag] = xu
      out['kfu_' + tag] = xfu
    else:
      out['xu_' + tag] = xu[:, ::8, ::8]
      out['kfu_' + tag] = xfu[:, ::8, ::8]
  save('F1_synth', **out)


# ------------------------------------------------------------------ F2 ----


def f2():
  out = {}
  dc = ref_myfft.DataConsistencyInKspace(norm='ortho')
  for tag, (b, h) in {'64': (2, 64), '256': (1, 256), '128x64': (2, None)}.items():
    hh, ww = (128, 64) if h is None else (h, h)
    g = torch.Generator().manual_seed(hh * 7 + ww)
    x = torch.randn(b, 2, hh, ww, generator=g, dtype=torch.float64)
    m2 = (torch.rand(b, 1, hh, ww, generator=g) < 0.3).double().expand(b, 2, hh, ww).contiguous()
    k0 = torch.randn(b, 2, hh, ww, generator=g, dtype=torch.float64) * m2
    gy = torch.randn(b, 2, hh, ww, generator=g, dtype=torch.float64)
    xv = x.clone().requires_grad_(True)
    y = dc.perform(xv, k0, m2)
    (y * gy).sum().backward()
    # numpy restatement inside the reference (compressed_sensing.py:515-529)
    xc = x[:, 0].numpy() + 1j * x[:, 1].numpy()
    yc = ref_cs.data_consistency(xc, k0[:, 0].numpy() + 1j * k0[:, 1].numpy(),
                                 m2[:, 0].numpy())
    assert np.allclose(yc.real, y[:, 0].detach().numpy(), atol=1e-10)
    # inputs are regenerated by the test from the same torch.Generator seed
    sub = (slice(None), slice(None), slice(None, None, 4), slice(None, None, 4)) if hh == 256 else Ellipsis
    out.update({'shape_' + tag: np.array([b, hh, ww]),
                'x_probe_' + tag: x.numpy()[..., :2, :2],
                'y_' + tag: y.detach().numpy()[sub], 'gx_' + tag: xv.grad.numpy()[sub]})
  save('F2_dc', **out)


# ------------------------------------------------------------------ F3 ----


def f3():
  out = {}
  for tag, nb in (('b1', 1), ('b5', 5)):
    conf = Configuration.from_json(os.path.join(REF, 'configs/1-recnet.json'))
    conf.model['num_blocks'] = nb
    conf.batch_size = 2
    ref_utils.set_random_seeds(conf.seed)
    runner = ref_RR.build_runner(conf, '', 'train')
    P0 = npd(runner.model.state_dict())
    batch = O.synth_batch(2, 64, 64, acc=4, seed=3)
    # forward / loss / grads (no update)
    pred = runner.model(batch['inp'], batch['kspace'], batch['mask'])
    loss = F.mse_loss(pred, batch['target'])
    runner.model.zero_grad()
    loss.backward()
    grads = {'grad.' + k: p.grad.numpy().copy()
             for k, p in runner.model.named_parameters()}
    runner.model.zero_grad()
    out.update({tag + '.P0.' + k: v for k, v in P0.items()})
    out.update({tag + '.' + k: v for k, v in grads.items()})
    out[tag + '.pred'] = pred.detach().numpy()
    out[tag + '.loss'] = np.float64(loss.item())
    # 3 Adam steps through the reference's own Runner.train_epoch
    losses = []
    for step in range(3):
      l, m = runner.train_epoch(_Loader([dict(batch)]), 1)
      losses.append([l['loss_MSE'].value, m['psnr'].value])
      if step in (0, 2):
        out.update({'%s.P%d.%s' % (tag, step + 1, k): v
                    for k, v in npd(runner.model.state_dict()).items()})
    out[tag + '.step_losses'] = np.array(losses)
  save('F3_recnet', **out)


# ------------------------------------------------------------------ F4 ----

SMALL_GEN = dict(rec_filters=8, enc=[8, 16, 32], dec=[16, 8])
SMALL_DISC = [8, 16, 32, 64, 64, 64]


def _gan_conf(small=True):
  conf = Configuration.from_json(os.path.join(REF, 'configs/2-refinement.json'))
  conf.generator_model['pretrained_model']['pretrained_weights'] = None
  conf.discriminator_model['name'] = 'CNNDiscriminator'
  conf.batch_size = 2
  if small:
    conf.generator_model['pretrained_model']['num_filters'] = SMALL_GEN['rec_filters']
    conf.generator_model['learnable_model']['encode_filters'] = SMALL_GEN['enc']
    conf.generator_model['learnable_model']['decode_filters'] = SMALL_GEN['dec']
    conf.discriminator_model['num_filters_per_layer'] = SMALL_DISC
  return conf


def f4():
  conf = _gan_conf()
  gc = Configuration.from_dict(conf.generator_model, conf)
  ref_utils.set_random_seeds(conf.seed)
  gen = construct_model(gc, gc.name, cuda='')
  gen.train()
  with torch.no_grad():
    gen.scale.fill_(0.37)
  state0 = npd(gen.state_dict())
  batch = O.synth_batch(2, 128, 128, acc=4, seed=11)
  o = gen(batch['inp'], batch['kspace'], batch['mask'])
  g = torch.Generator().manual_seed(5)
  gp = torch.randn(o['pred'].shape, generator=g)
  gu = torch.randn(o['prescaled_refinement'].shape, generator=g)
  ((o['pred'] * gp).sum() + (o['prescaled_refinement'] * gu).sum()).backward()
  out = {'P.' + k: v for k, v in state0.items()}
  out.update({'S1.' + k: v for k, v in npd(gen.state_dict()).items()
              if 'running' in k})
  out.update({'out.' + k: v.detach().numpy() for k, v in o.items()})
  out.update({'grad.' + k: p.grad.numpy() for k, p in gen.named_parameters()
              if p.grad is not None})
  out['gp'], out['gu'] = gp.numpy(), gu.numpy()
  save('F4_refinement', **out)


# ------------------------------------------------------------------ F5 ----


class _InjectedDropout(object):
  """Replaces nn.Dropout2d.forward by multiplication with queued masks and
  records them.  torch 2.x cannot backprop through the reference's in-place
  Dropout2d after in-place LeakyReLU, so dropout runs out of place and the
  discriminator's forward is wrapped to put the POST-dropout tensors into
  ``features`` -- which is what the in-place op produced in torch 0.3.1
  (the appended feature IS the tensor dropout mutates; SURVEY A-7)."""

  def __init__(self):
    self.queue = []
    self.used = []
    self._outs = []

  def install(self, model, gen):
    me = self

    def fwd(mod, x):
      if not mod.training:
        return x
      if me.queue:
        m = me.queue.pop(0)
      else:
        keep = torch.bernoulli(torch.full((x.shape[0], x.shape[1], 1, 1),
                                          1 - mod.p), generator=gen)
        m = keep / (1 - mod.p)
      me.used.append(m)
      y = x * m
      me._outs.append((x, y))
      return y

    for m in model.modules():
      if isinstance(m, nn.Dropout2d):
        m.inplace = False
        m.forward = types.MethodType(fwd, m)

    orig_forward = model.forward

    def wrapped(inp):
      me._outs = []
      out = orig_forward(inp)
      if 'features' in out:
        for pre, post in me._outs:
          for i, ft in enumerate(out['features']):
            if ft is pre:
              out['features'][i] = post
      return out

    model.forward = wrapped


def f5():
  conf = _gan_conf()
  dconf = Configuration.from_dict(conf.discriminator_model, conf)
  ref_utils.set_random_seeds(7)
  disc = construct_model(dconf, 'CNNDiscriminator', cuda='')
  disc.train()
  inj = _InjectedDropout()
  g = torch.Generator().manual_seed(3)
  inj.install(disc, g)
  state0 = npd(disc.state_dict())
  x_fake = torch.rand(2, 1, 128, 128, generator=g)
  x_real = torch.rand(2, 1, 128, 128, generator=g)
  xf = x_fake.clone().requires_grad_(True)
  of = disc(xf)
  orr = disc(x_real)
  crit_d = _orig_get_criterion(conf, 'gan', '', loss_type='disc')
  crit_g = _orig_get_criterion(conf, 'gan', '', loss_type='gen')
  crit_fm = _orig_get_criterion(conf, 'FeatureMatching', '', loss_type='gen')
  ld = crit_d(of, orr)
  lg = crit_g(of, orr)
  lfm = crit_fm(of, orr)
  (ld + 0.5 * lg + lfm).backward()
  out = {'P.' + k: v for k, v in state0.items()}
  out.update({'S1.' + k: v for k, v in npd(disc.state_dict()).items() if 'running' in k})
  out['x_fake'], out['x_real'] = x_fake.numpy(), x_real.numpy()
  for i, m in enumerate(inj.used):
    out['mask%d' % i] = m.numpy()
  out['prob_fake'] = of['prob'].detach().numpy()
  out['logits_fake'] = of['logits'].detach().numpy()
  out['logits_real'] = orr['logits'].detach().numpy()
  for i, f in enumerate(of['features']):
    out['feat_fake%d' % i] = f.detach().numpy()
  out['loss_disc'], out['loss_gen'], out['loss_fm'] = (np.float64(ld.item()),
                                                       np.float64(lg.item()),
                                                       np.float64(lfm.item()))
  out['grad_x'] = xf.grad.numpy()
  out.update({'grad.' + k: p.grad.numpy() for k, p in disc.named_parameters()})
  save('F5_disc', **out)


# ------------------------------------------------------------------ F6 ----


def _load_vgg_weights(vgg_module, seed):
  PV = O.init_vgg(gen=torch.Generator().manual_seed(seed))
  sd = vgg_module.state_dict()
  for k, v in PV.items():
    assert k in sd and sd[k].shape == v.shape, k
    sd[k].copy_(v)
  return PV


def f6():
  conf = _gan_conf()
  crit = _orig_get_criterion(conf, 'VGG19', '')
  _load_vgg_weights(crit.criterion.vgg, seed=19)
  batch = O.synth_batch(2, 64, 64, acc=4, seed=21)
  g = torch.Generator().manual_seed(2)
  pred = (batch['target'] + 0.05 * torch.randn(batch['target'].shape, generator=g)).requires_grad_(True)
  loss = crit({'pred': pred}, batch)
  loss.backward()
  from utils.tensor_transforms import complex_abs
  p = complex_abs(pred.detach())
  feat = crit.criterion.vgg(torch.cat((p, p, p), 1))[0]
  save('F6_vgg', vgg_seed=np.int64(19), pred=pred.detach().numpy(),
       target=batch['target'].numpy(), loss=np.float64(loss.item()),
       grad_pred=pred.grad.numpy(), feat_sum=np.float64(feat.double().sum().item()),
       feat_abs_sum=np.float64(feat.double().abs().sum().item()),
       feat_slice=feat[:, :4].detach().numpy())


# ------------------------------------------------------------------ F7 ----


def f7():
  """Two full GAN steps of the reference's AdversarialRunner.train_epoch at 128^2,
  B=2, reduced widths, injected dropout masks, torch-0.3.1 step semantics (A)."""
  conf = _gan_conf()
  ref_utils.set_random_seeds(conf.seed)
  runner = ref_AR.build_runner(conf, '', 'train')
  vgg_crit = runner.gen_criteria['VGG19'].c
  _load_vgg_weights(vgg_crit.criterion.vgg, seed=19)
  with torch.no_grad():
    runner.gen.scale.fill_(0.25)
  inj = _InjectedDropout()
  g = torch.Generator().manual_seed(99)
  inj.install(runner.disc, g)
  out = {'G0.' + k: v for k, v in npd(runner.gen.state_dict()).items()}
  out.update({'D0.' + k: v for k, v in npd(runner.disc.state_dict()).items()})
  out['loss_weights_gen'] = runner.gen_loss_weights.numpy()
  out['loss_weights_disc'] = runner.disc_loss_weights.numpy()
  out['loss_order_gen'] = np.array(list(runner.gen_adv_criteria) + list(runner.gen_criteria))
  names = None
  for step in range(2):
    batch = O.synth_batch(2, 128, 128, acc=4, seed=40 + step)
    n_before = len(inj.used)
    l, m = runner.train_epoch(_Loader([batch]), 1)
    for j, mk in enumerate(inj.used[n_before:]):
      out['step%d.mask%d' % (step, j)] = mk.numpy()
    names = sorted(l.keys())
    out['step%d.losses' % step] = np.array([l[k].value for k in names], dtype=np.float64)
    out['step%d.metrics' % step] = np.array([m['gen_psnr'].value,
                                             float(m['disc_binary_accuracy'].value)], dtype=np.float64)
    out.update({'G%d.%s' % (step + 1, k): v for k, v in npd(runner.gen.state_dict()).items()
                if not k.startswith('pretrained_model')})
    out.update({'D%d.%s' % (step + 1, k): v for k, v in npd(runner.disc.state_dict()).items()})
  out['loss_names'] = np.array(names)
  out['vgg_seed'] = np.int64(19)
  save('F7_gan_step', **out)


# ------------------------------------------------------------------ F8 ----


def f8():
  conf = _gan_conf()
  fn = ref_metrics.get_metric_fn(conf, 'psnr', '', 'train')
  g = torch.Generator().manual_seed(8)
  target = torch.rand(3, 2, 32, 32, generator=g) * 1.2
  pred = target + 0.1 * torch.randn(3, 2, 32, 32, generator=g)
  val = fn({'pred': pred}, {'target': target}).value
  save('F8_psnr', pred=pred.numpy(), target=target.numpy(), psnr=np.float64(val))


# ------------------------------------------------------------------ F9 ----
# SSIM validation metric: the reference's MetricFunction('ssim') on a batch.
# metrics/image_metrics.py:41 reads `.data[0]` of a 0-dim tensor (torch-0.3 idiom): only that
# read is replaced; the arithmetic is the reference's own metrics/pytorch_ssim module.


def f9():
  from metrics import pytorch_ssim as ref_ssim
  ref_image_metrics.compute_ssim = \
      lambda p, t, window_size=11: float(ref_ssim.ssim(p, t, window_size=window_size).item())
  conf = _gan_conf()
  fn = ref_metrics.get_metric_fn(conf, 'ssim', '', 'test')
  g = torch.Generator().manual_seed(9)
  yy, xx = torch.meshgrid(torch.linspace(-1, 1, 48), torch.linspace(-1, 1, 48), indexing='ij')
  base = torch.exp(-3.0 * (xx ** 2 + yy ** 2))[None, None] * torch.rand(3, 1, 1, 1, generator=g)
  target = torch.cat([base + 0.05 * torch.rand(3, 1, 48, 48, generator=g),
                      0.02 * torch.randn(3, 1, 48, 48, generator=g)], 1) * 1.1
  pred = target + 0.05 * torch.randn(3, 2, 48, 48, generator=g)
  per_image = [fn({'pred': pred[i:i + 1]}, {'target': target[i:i + 1]}).value for i in range(3)]
  val = fn({'pred': pred}, {'target': target}).value
  save('F9_ssim', pred=pred.numpy(), target=target.numpy(), ssim=np.float64(val),
       ssim_per_image=np.array(per_image, dtype=np.float64))


# ----------------------------------------------------------------- F10 ----
# radial undersampling masks (BASELINE config 5 data format): the reference's radial_sampling,
# golden-angle with random start and uniform spokes; stored as sample indices (bit-exact test).


def f10():
  out = {}
  for tag, (n, nx, lines, golden) in {'g512': (2, 512, 70, True), 'u128': (2, 128, 24, False),
                                      'g64': (1, 64, 8, True)}.items():
    rng = np.random.RandomState(4321)
    m = ref_cs.radial_sampling((n, nx, nx), lines, rand=True, golden_angle=golden, centred=False, rng=rng)
    out['idx_' + tag] = np.flatnonzero(m).astype(np.int64)
    out['shape_' + tag] = np.array(m.shape, dtype=np.int64)
    out['args_' + tag] = np.array([n, nx, lines, int(golden)], dtype=np.int64)
  save('F10_radial', **out)


# ----------------------------------------------------------------- F11 ----
# several updates per step, a discriminator-pretraining schedule and both learning-rate
# schedulers, through the reference's own AdversarialRunner (epoch_beginning / train_epoch /
# epoch_finished as train.py:263-276 calls them).  Epoch 1: generator disabled (discriminator
# pretraining), epochs 2-3: two discriminator updates + one generator update per step.


def f11():
  conf = _gan_conf()
  conf.discriminator_optimizer = dict(conf.discriminator_optimizer, updates_per_step=2, lr_scheduler='linear',
                                      end_learning_rate=2e-5, decay_steps=4)
  conf.generator_optimizer = dict(conf.generator_optimizer, lr_scheduler='multistep', decay_steps=[2],
                                  decay_factor=0.5)
  conf.pretrain_discriminator_epochs = 1
  ref_utils.set_random_seeds(conf.seed)
  runner = ref_AR.build_runner(conf, '', 'train')
  vgg_crit = runner.gen_criteria['VGG19'].c
  _load_vgg_weights(vgg_crit.criterion.vgg, seed=19)
  with torch.no_grad():
    runner.gen.scale.fill_(0.25)
  inj = _InjectedDropout()
  g = torch.Generator().manual_seed(77)
  inj.install(runner.disc, g)
  out = {'G0.' + k: v for k, v in npd(runner.gen.state_dict()).items()}
  out.update({'D0.' + k: v for k, v in npd(runner.disc.state_dict()).items()})
  out['conf'] = np.array(['disc updates_per_step=2 lr_scheduler=linear end_learning_rate=2e-5 decay_steps=4; '
                          'gen lr_scheduler=multistep decay_steps=[2] decay_factor=0.5; '
                          'pretrain_discriminator_epochs=1'])
  for epoch in (1, 2, 3):
    runner.epoch_beginning(epoch)
    out['ep%d.flags' % epoch] = np.array([int(runner.discriminator_enabled), int(runner.generator_enabled)])
    out['ep%d.lr' % epoch] = np.array([runner.gen_optimizer.param_groups[0]['lr'],
                                       runner.disc_optimizer.param_groups[0]['lr']], dtype=np.float64)
    batches = [O.synth_batch(2, 128, 128, acc=4, seed=300 + 10 * epoch + i) for i in range(2)]
    n_before = len(inj.used)
    l, m = runner.train_epoch(_Loader(batches), epoch)
    runner.epoch_finished(epoch)
    used = inj.used[n_before:]
    out['ep%d.num_masks' % epoch] = np.int64(len(used))
    for j, mk in enumerate(used):
      out['ep%d.mask%d' % (epoch, j)] = mk.numpy()
    names = sorted(l.keys())
    out['ep%d.loss_names' % epoch] = np.array(names)
    out['ep%d.losses' % epoch] = np.array([l[k].value for k in names], dtype=np.float64)
    out['ep%d.gen_psnr' % epoch] = np.float64(m['gen_psnr'].value)
    out.update({'G%d.%s' % (epoch, k): v for k, v in npd(runner.gen.state_dict()).items()
                if not k.startswith('pretrained_model')})
    out.update({'D%d.%s' % (epoch, k): v for k, v in npd(runner.disc.state_dict()).items()})
  out['vgg_seed'] = np.int64(19)
  save('F11_schedules', **out)


# ----------------------------------------------------------------- F12 ----
# the reference's weight initialisation (models/weight_inits.py:5-114 with the per-model
# weight_init_params: recnet.py:54-59, unet.py:253-259, discriminators.py:189-209) under fixed torch seeds, at the
# FULL widths of configs/2-refinement.json and for the 1-recnet.json RecNet.  torch's CPU generator gives the
# same stream on every machine for one torch version, so a construction that draws in the same order with the
# same initialisers reproduces the tensors bit for bit: the fixture stores per tensor its shape, first 8
# values, float64 sum, sum of squares and an order-sensitive checksum (dot with cos(0.37 i)).


def _digest(t):
  a = t.detach().double().reshape(-1).numpy()
  i = np.arange(a.size, dtype=np.float64)
  return np.array([a.sum(), (a * a).sum(), (a * np.cos(0.37 * i)).sum(), a.min(), a.max()], dtype=np.float64)


def f12():
  out = {}
  conf = _gan_conf(small=False)
  ref_utils.set_random_seeds(7)
  gc = Configuration.from_dict(conf.generator_model, conf)
  gen = construct_model(gc, gc.name, cuda='')
  ref_utils.set_random_seeds(8)
  dc = Configuration.from_dict(conf.discriminator_model, conf)
  disc = construct_model(dc, dc.name, cuda='')
  rconf = Configuration.from_json(os.path.join(REF, 'configs/1-recnet.json'))
  ref_utils.set_random_seeds(9)
  rc = Configuration.from_dict(rconf.model, rconf)
  rec = construct_model(rc, rc.name, cuda='')
  for tag, model in (('G', gen), ('D', disc), ('R', rec)):
    for k, v in model.state_dict().items():
      if 'num_batches' in k:
        continue
      out['%s.%s.digest' % (tag, k)] = _digest(v)
      out['%s.%s.head' % (tag, k)] = v.detach().reshape(-1)[:8].numpy().copy()
      out['%s.%s.shape' % (tag, k)] = np.array(v.shape, dtype=np.int64)
  out['seeds'] = np.array([7, 8, 9])
  save('F12_weight_init', **out)


# ----------------------------------------------------------------- F13 ----
# a checkpoint written by the reference's OWN save_checkpoint (utils/checkpoints.py:9-16) from its
# AdversarialRunner after one training step (so the Adam states are populated), and what the reference then
# does with it: the generator's eval-mode prediction on a batch and the losses of the NEXT training step.


def f13():
  import utils.checkpoints as ref_ck
  conf = _gan_conf()
  ref_utils.set_random_seeds(conf.seed)
  runner = ref_AR.build_runner(conf, '', 'train')
  vgg_crit = runner.gen_criteria['VGG19'].c
  _load_vgg_weights(vgg_crit.criterion.vgg, seed=19)
  with torch.no_grad():
    runner.gen.scale.fill_(0.25)
  inj = _InjectedDropout()
  g = torch.Generator().manual_seed(55)
  inj.install(runner.disc, g)
  out = {}
  batch0 = O.synth_batch(2, 128, 128, acc=4, seed=900)
  runner.train_epoch(_Loader([batch0]), 1)
  for j, mk in enumerate(inj.used):
    out['step0.mask%d' % j] = mk.numpy()
  path = os.path.join(HERE, 'F13_reference_checkpoint.pth')
  conf._src_file = None                      # the pickled Configuration must not point into the reference tree
  ref_ck.save_checkpoint(path, conf, runner, 4, {'psnr': 31.5})
  print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024.0))
  runner._set_test()
  with torch.no_grad():
    vb = O.synth_batch(2, 128, 128, acc=4, seed=901)
    pred = runner.gen(vb['inp'], vb['kspace'], vb['mask'])['pred']
  out['eval_pred'] = pred.numpy()
  n0 = len(inj.used)
  batch1 = O.synth_batch(2, 128, 128, acc=4, seed=902)
  l, m = runner.train_epoch(_Loader([batch1]), 2)
  for j, mk in enumerate(inj.used[n0:]):
    out['step1.mask%d' % j] = mk.numpy()
  names = sorted(l.keys())
  out['loss_names'] = np.array(names)
  out['step1.losses'] = np.array([l[k].value for k in names], dtype=np.float64)
  out['step1.gen_psnr'] = np.float64(m['gen_psnr'].value)
  out.update({'G2.' + k: v for k, v in npd(runner.gen.state_dict()).items() if not k.startswith('pretrained_model')})
  save('F13_checkpoint_expected', **out)


if __name__ == '__main__':
  which = sys.argv[1:] or ['f1', 'f2', 'f3', 'f4', 'f5', 'f6', 'f7', 'f8', 'f9', 'f10', 'f11', 'f12', 'f13']
  for name in which:
    print('==', name)
    globals()[name]()
  np.savez(os.path.join(HERE, 'META.npz'), torch_version=np.array(torch.__version__),
           numpy_version=np.array(np.__version__))
