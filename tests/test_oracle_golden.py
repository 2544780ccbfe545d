"""Pins the CPU oracle (oracle/csmri_oracle.py) to golden vectors produced by
running the reference itself (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

import csmri_oracle as O

from conftest import GOLDEN


def load(name):
  return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def T(a):
  return torch.from_numpy(np.asarray(a))


def sub(d, prefix):
  return {k[len(prefix):]: T(v) for k, v in d.items() if k.startswith(prefix)}


# ------------------------------------------------------------------ F1 ----


@pytest.mark.parametrize('h', [64, 256])
def test_f1_mask_and_undersample(h):
  f = load('F1_synth')
  rng = np.random.RandomState(1234)
  m = O.cartesian_mask((2, h, h), 4, 8, rng)
  # integer mask indexing is bit-exact
  assert np.array_equal(m[:, :, 0].astype(np.uint8), f['mask_%d' % h])
  assert np.array_equal(np.stack([np.nonzero(m[i, :, 0])[0] for i in range(2)]),
                        f['rows_%d' % h])
  assert np.all(m == m[:, :, :1])
  assert f['rows_%d' % h].shape[1] == h // 4
  img = np.stack([O.phantom(h, h, 5 + i) for i in range(2)])
  k = m * np.fft.fft2(img, norm='ortho')
  xu = np.fft.ifft2(k, norm='ortho')
  s = 1 if h == 64 else 8
  assert np.allclose(xu[:, ::s, ::s], f['xu_%d' % h], atol=1e-12)
  assert np.allclose(k[:, ::s, ::s], f['kfu_%d' % h], atol=1e-12)


def test_synth_batch_layout():
  b = O.synth_batch(2, 64, 64, acc=4, seed=0)
  for k in ('inp', 'kspace', 'mask', 'target'):
    assert b[k].shape == (2, 2, 64, 64) and b[k].dtype == torch.float32
  assert torch.equal(b['mask'][:, 0], b['mask'][:, 1])
  assert set(np.unique(b['mask'].numpy())) == {0.0, 1.0}
  assert float(b['target'][:, 1].abs().max()) == 0.0
  # k-space is zero off the mask (data_consistency relies on it, myfft.py:141)
  assert float((b['kspace'] * (1 - b['mask'])).abs().max()) == 0.0
  assert int(b['mask'][0, 0, :, 0].sum()) == 16


# ------------------------------------------------------------------ F2 ----


def _f2_inputs(b, hh, ww):
  g = torch.Generator().manual_seed(hh * 7 + ww)
  x = torch.randn(b, 2, hh, ww, generator=g, dtype=torch.float64)
  m2 = (torch.rand(b, 1, hh, ww, generator=g) < 0.3).double().expand(b, 2, hh, ww).contiguous()
  k0 = torch.randn(b, 2, hh, ww, generator=g, dtype=torch.float64) * m2
  gy = torch.randn(b, 2, hh, ww, generator=g, dtype=torch.float64)
  return x, m2, k0, gy


@pytest.mark.parametrize('tag', ['64', '256', '128x64'])
def test_f2_dc(tag):
  f = load('F2_dc')
  b, hh, ww = [int(v) for v in f['shape_' + tag]]
  x, m2, k0, gy = _f2_inputs(b, hh, ww)
  assert np.array_equal(x.numpy()[..., :2, :2], f['x_probe_' + tag])
  y = O.dc_layer(x, k0, m2)
  gx = O.dc_adjoint(gy, m2)
  s = (slice(None), slice(None), slice(None, None, 4), slice(None, None, 4)) if hh == 256 else Ellipsis
  assert np.allclose(y.numpy()[s], f['y_' + tag], atol=1e-12)
  assert np.allclose(gx.numpy()[s], f['gx_' + tag], atol=1e-12)


# ------------------------------------------------------------------ F3 ----


@pytest.mark.parametrize('tag,nb', [('b1', 1), ('b5', 5)])
def test_f3_recnet(tag, nb):
  f = load('F3_recnet')
  P = {k: v.clone().requires_grad_(True) for k, v in sub(f, tag + '.P0.').items()}
  batch = O.synth_batch(2, 64, 64, acc=4, seed=3)
  pred = O.recnet_forward(P, batch['inp'], batch['kspace'], batch['mask'], nb)
  assert torch.allclose(pred, T(f[tag + '.pred']), atol=2e-6, rtol=1e-5)
  loss = torch.nn.functional.mse_loss(pred, batch['target'])
  assert abs(loss.item() - float(f[tag + '.loss'])) < 1e-7
  loss.backward()
  for k, g in sub(f, tag + '.grad.').items():
    assert torch.allclose(P[k].grad, g, atol=1e-7, rtol=1e-4), k
  for p in P.values():
    p.grad = None
  opt = O.make_adam(P.values(), 2e-4, 0.9, 0.999)
  for step in range(3):
    l, _ = O.recnet_mse_step(P, opt, batch, nb)
    assert abs(l['loss_MSE'] - f[tag + '.step_losses'][step, 0]) < 1e-6
    assert abs(l['psnr'] - f[tag + '.step_losses'][step, 1]) < 1e-4
    if step in (0, 2):
      for k, v in sub(f, '%s.P%d.' % (tag, step + 1)).items():
        assert torch.allclose(P[k].detach(), v, atol=1e-6, rtol=1e-5), (step, k)


# ------------------------------------------------------------------ F4 ----

SMALL_UNET = dict(O.UNET_CONF, encode_filters=[8, 16, 32], decode_filters=[16, 8])


def _split_state(d):
  P = {k: v for k, v in d.items() if 'running' not in k and 'num_batches' not in k}
  S = {k: v.clone() for k, v in d.items() if 'running' in k}
  return P, S


def test_f4_refinement():
  f = load('F4_refinement')
  P, S = _split_state(sub(f, 'P.'))
  P = {k: (v.clone().requires_grad_(True) if not k.startswith('pretrained_model') else v)
       for k, v in P.items()}
  batch = O.synth_batch(2, 128, 128, acc=4, seed=11)
  Sg = S
  o = _refine(P, Sg, batch)
  for k in ('pred', 'pretrained', 'prescaled_refinement', 'scaled_refinement'):
    assert torch.allclose(o[k], T(f['out.' + k]), atol=5e-6, rtol=1e-5), k
  ((o['pred'] * T(f['gp'])).sum() + (o['prescaled_refinement'] * T(f['gu'])).sum()).backward()
  for k, g in sub(f, 'grad.').items():
    assert torch.allclose(P[k].grad, g, atol=2e-5, rtol=2e-4), k
  for k, v in sub(f, 'S1.').items():
    assert torch.allclose(Sg[k], v, atol=1e-6, rtol=1e-5), k


def _refine(P, S, batch):
  # UNET_CONF is consulted through unet_forward's default arg -> pass explicitly
  import torch as t
  with t.no_grad():
    pre = O.recnet_forward(P, batch['inp'], batch['kspace'], batch['mask'], 3, 3,
                           prefix='pretrained_model.conv_blocks')
  real_scaled, mn, mx = O.scale_minmax(pre[:, 0:1].contiguous())
  u = O.unet_forward(P, S, pre, True, conf=SMALL_UNET, prefix='learnable_model.')
  us = P['scale'] * u
  out_real = O.unscale_minmax(real_scaled + us, mn, mx)
  return {'pred': t.cat((out_real, pre[:, 1:2]), 1), 'pretrained': pre,
          'prescaled_refinement': u, 'scaled_refinement': us}


# ------------------------------------------------------------------ F5 ----

SMALL_DISC = dict(O.DISC_CONF, filters=[8, 16, 32, 64, 64, 64])


def test_f5_disc():
  f = load('F5_disc')
  P, S = _split_state(sub(f, 'P.'))
  P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
  masks = [T(f['mask%d' % i]) for i in range(6)]
  xf = T(f['x_fake']).clone().requires_grad_(True)
  of = O.disc_forward(P, S, xf, True, SMALL_DISC, dropout_masks=masks[:3])
  orr = O.disc_forward(P, S, T(f['x_real']), True, SMALL_DISC, dropout_masks=masks[3:])
  assert torch.allclose(of['logits'], T(f['logits_fake']), atol=1e-5, rtol=1e-4)
  assert torch.allclose(orr['logits'], T(f['logits_real']), atol=1e-5, rtol=1e-4)
  for i, ft in enumerate(of['features']):
    assert torch.allclose(ft, T(f['feat_fake%d' % i]), atol=1e-5, rtol=1e-4), i
  ld = O.gan_loss_disc(of, orr, 0.1)
  lg = O.gan_loss_gen(of)
  lfm = O.feature_matching_loss(of, orr)
  assert abs(ld.item() - float(f['loss_disc'])) < 1e-6
  assert abs(lg.item() - float(f['loss_gen'])) < 1e-6
  assert abs(lfm.item() - float(f['loss_fm'])) < 1e-6
  (ld + 0.5 * lg + lfm).backward()
  assert torch.allclose(xf.grad, T(f['grad_x']), atol=1e-7, rtol=1e-3)
  for k, g in sub(f, 'grad.').items():
    assert torch.allclose(P[k].grad, g, atol=1e-6, rtol=1e-3), k
  for k, v in sub(f, 'S1.').items():
    assert torch.allclose(S[k], v, atol=1e-6, rtol=1e-5), k


def test_disc_layer_indices_full_width():
  # convs.{1,4,8,12,17,22}, BN at convs.{5,9,13,18,23}
  idx = O.disc_layer_indices(O.DISC_CONF)
  assert [c for c, _ in idx] == [1, 4, 8, 12, 17, 22]
  assert [b for _, b in idx] == [None, 5, 9, 13, 18, 23]


# ------------------------------------------------------------------ F6 ----


def test_f6_vgg():
  f = load('F6_vgg')
  PV = O.init_vgg(gen=torch.Generator().manual_seed(int(f['vgg_seed'])))
  pred = T(f['pred']).clone().requires_grad_(True)
  loss = O.vgg_loss(PV, pred, T(f['target']))
  assert abs(loss.item() - float(f['loss'])) < 1e-5 * max(1.0, abs(float(f['loss'])))
  loss.backward()
  assert torch.allclose(pred.grad, T(f['grad_pred']), atol=1e-6, rtol=1e-3)
  p = O.complex_abs(pred.detach())
  feat = O.vgg_features(PV, torch.cat((p, p, p), 1))
  assert torch.allclose(feat[:, :4], T(f['feat_slice']), atol=1e-4, rtol=1e-4)
  assert abs(feat.double().sum().item() - float(f['feat_sum'])) < 1e-4 * float(f['feat_abs_sum'])


# ------------------------------------------------------------------ F7 ----


def test_f7_gan_steps():
  f = load('F7_gan_step')
  PG, SG = _split_state(sub(f, 'G0.'))
  PD, SD = _split_state(sub(f, 'D0.'))
  PG = {k: (v.clone().requires_grad_(True) if not k.startswith('pretrained_model') else v)
        for k, v in PG.items()}
  PD = {k: v.clone().requires_grad_(True) for k, v in PD.items()}
  PV = O.init_vgg(gen=torch.Generator().manual_seed(int(f['vgg_seed'])))
  assert list(f['loss_order_gen']) == ['gan', 'FeatureMatching', 'VGG19', 'FeaturePenalty']
  assert np.allclose(f['loss_weights_gen'], [0.5, 1.0, 10.0, 2.0])
  gopt = O.make_adam([v for k, v in PG.items() if v.requires_grad], 2e-4, 0.5, 0.999)
  dopt = O.make_adam(PD.values(), 2e-4, 0.5, 0.999)
  names = [str(n) for n in f['loss_names']]
  orig_unet, orig_disc = O.UNET_CONF, O.DISC_CONF
  O.UNET_CONF, O.DISC_CONF = SMALL_UNET, SMALL_DISC
  try:
    _patch_defaults()
    pool = O.ImagePool(80)
    for step in range(2):
      batch = O.synth_batch(2, 128, 128, acc=4, seed=40 + step)
      masks = [T(f['step%d.mask%d' % (step, j)]) for j in range(9)]
      dm = [masks[0:3], masks[3:6], masks[6:9]]
      losses, metrics, _ = O.gan_train_step(PG, SG, PD, SD, PV, gopt, dopt, batch,
                                            pool=pool, dropout_masks=dm, faithful=True)
      ref = dict(zip(names, f['step%d.losses' % step]))
      for k in names:
        assert abs(losses[k] - ref[k]) < 2e-5 * max(1.0, abs(ref[k])), (step, k, losses[k], ref[k])
      assert abs(metrics['gen_psnr'] - f['step%d.metrics' % step][0]) < 1e-3
      assert abs(metrics['disc_binary_accuracy'] - f['step%d.metrics' % step][1]) < 1e-6
      for k, v in sub(f, 'G%d.' % (step + 1)).items():
        cur = PG[k] if k in PG else SG.get(k)
        if cur is None:
          continue
        assert torch.allclose(cur.detach(), v, atol=2e-5, rtol=1e-3), (step, 'G', k)
      for k, v in sub(f, 'D%d.' % (step + 1)).items():
        cur = PD[k] if k in PD else SD.get(k)
        if cur is None:
          continue
        assert torch.allclose(cur.detach(), v, atol=2e-5, rtol=1e-3), (step, 'D', k)
  finally:
    O.UNET_CONF, O.DISC_CONF = orig_unet, orig_disc
    _patch_defaults()


def _patch_defaults():
  """unet_forward/disc_forward bind their conf default at def time; rebind."""
  O.unet_forward.__defaults__ = tuple(O.UNET_CONF if isinstance(d, dict) and 'encode_filters' in d else d
                                      for d in O.unet_forward.__defaults__)
  O.disc_forward.__defaults__ = tuple(O.DISC_CONF if isinstance(d, dict) and 'filters' in d else d
                                      for d in O.disc_forward.__defaults__)


# ------------------------------------------------------------------ F8 ----


def test_f8_psnr():
  f = load('F8_psnr')
  v = O.psnr_batch(T(f['pred']), T(f['target']))
  assert abs(v - float(f['psnr'])) < 1e-6


def test_f9_ssim():
  """SSIM validation metric against the reference's own pytorch_ssim."""
  f = load('F9_ssim')
  vals = O.ssim_images(T(f['pred']), T(f['target']))
  assert np.allclose(vals, f['ssim_per_image'], rtol=0, atol=2e-6), (vals, f['ssim_per_image'])
  assert abs(float(np.mean(vals)) - float(f['ssim'])) < 2e-6


def test_f10_radial_masks_bit_exact():
  """Radial undersampling (BASELINE config 5 data format): sample indices equal the reference's."""
  f = load('F10_radial')
  for tag in ('g512', 'u128', 'g64'):
    n, nx, lines, golden = (int(v) for v in f['args_' + tag])
    m = O.radial_mask((n, nx, nx), lines, rand=True, golden_angle=bool(golden), centred=False,
                      rng=np.random.RandomState(4321))
    assert tuple(m.shape) == tuple(int(v) for v in f['shape_' + tag])
    assert np.array_equal(np.flatnonzero(m), f['idx_' + tag]), tag


# ----------------------------------------------------------------- F11 ----


def test_f11_multi_update_steps_schedules_and_lr_schedulers():
  """the reference's _train_multiple_steps (2 discriminator updates + 1 generator update
  per step), discriminator pretraining in epoch 1 and both LR schedulers over three epochs."""
  f = load('F11_schedules')
  PG, SG = _split_state(sub(f, 'G0.'))
  PD, SD = _split_state(sub(f, 'D0.'))
  PG = {k: (v.clone().requires_grad_(True) if not k.startswith('pretrained_model') else v)
        for k, v in PG.items()}
  PD = {k: v.clone().requires_grad_(True) for k, v in PD.items()}
  PV = O.init_vgg(gen=torch.Generator().manual_seed(int(f['vgg_seed'])))
  gopt = O.make_adam([v for k, v in PG.items() if v.requires_grad], 2e-4, 0.5, 0.999)
  dopt = O.make_adam(PD.values(), 2e-4, 0.5, 0.999)
  orig_unet, orig_disc = O.UNET_CONF, O.DISC_CONF
  O.UNET_CONF, O.DISC_CONF = SMALL_UNET, SMALL_DISC
  try:
    _patch_defaults()
    pool = O.ImagePool(80)
    for epoch in (1, 2, 3):
      disc_en, gen_en = O.pretraining_flags(epoch, None, 1)
      assert [int(disc_en), int(gen_en)] == list(f['ep%d.flags' % epoch])
      lr_g = O.lr_multistep(2e-4, [2], 0.5, epoch)
      lr_d = O.lr_polynomial(2e-4, 2e-5, 4, epoch)
      assert np.allclose([lr_g, lr_d], f['ep%d.lr' % epoch], rtol=1e-12)
      gopt.param_groups[0]['lr'], dopt.param_groups[0]['lr'] = lr_g, lr_d
      batches = [O.synth_batch(2, 128, 128, acc=4, seed=300 + 10 * epoch + i) for i in range(2)]
      n = int(f['ep%d.num_masks' % epoch])
      masks = [T(f['ep%d.mask%d' % (epoch, j)]) for j in range(n)]
      dm = [masks[i:i + 3] for i in range(0, n, 3)]
      losses, out_gen = O.gan_train_multi_step(PG, SG, PD, SD, PV, gopt, dopt, batches, disc_updates=2,
                                               gen_updates=1, disc_enabled=disc_en, gen_enabled=gen_en,
                                               pool=pool, dropout_masks=dm)
      names = [str(s) for s in f['ep%d.loss_names' % epoch]]
      assert sorted(losses) == names
      for k, v in zip(names, f['ep%d.losses' % epoch]):
        assert abs(losses[k] - v) < 5e-5 * max(1.0, abs(v)), (epoch, k, losses[k], v)
      psnr = O.psnr_batch(out_gen['pred'].detach(), batches[1 if not gen_en else 0]['target'])
      assert abs(psnr - float(f['ep%d.gen_psnr' % epoch])) < 2e-3
      for tag, P, S in (('G', PG, SG), ('D', PD, SD)):
        for k, v in sub(f, '%s%d.' % (tag, epoch)).items():
          cur = P[k] if k in P else S.get(k)
          if cur is None:
            continue
          assert torch.allclose(cur.detach().float(), v.float(), atol=5e-5, rtol=2e-3), (epoch, tag, k)
  finally:
    O.UNET_CONF, O.DISC_CONF = orig_unet, orig_disc
    _patch_defaults()


def test_philox4x32_10_known_answers_and_dropout_mask():
  """The oracle's numpy Philox4x32-10 against the published known-answer vectors of the generator (Random123
  kat_vectors: `philox4x32 10`), and the Dropout2d mask derived from it (what csmri_dropout2d_mask must write)."""
  import numpy as np
  kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
         ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
         ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
          (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
  for ctr, key, want in kat:
    got = O.philox4x32_10([list(ctr)], key)[0]
    assert tuple(int(x) for x in got) == want, (ctr, key, [hex(int(x)) for x in got])
  m = O.dropout2d_mask(0x1234567890abcdef, 3, 40961, 0.5)
  assert m.shape == (40961,) and set(m.unique().tolist()) == {0.0, 2.0} and abs(float(m.mean()) - 1.0) < 0.02
  m2 = O.dropout2d_mask(0x1234567890abcdef, 4, 40961, 0.5)
  assert not torch.equal(m, m2)                                  # the call counter decorrelates successive passes
  m3 = O.dropout2d_mask(7, 0, 1000, 0.2)
  assert set(np.round(m3.unique().numpy(), 5).tolist()) == {0.0, 1.25} and abs(float((m3 > 0).float().mean()) - 0.8) < 0.05
