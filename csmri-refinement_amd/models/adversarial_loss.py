"""Adversarial criteria (reference models/adversarial_loss.py:7-160).

GAN: BCE on the discriminator's sigmoid output with label smoothing; evaluated as
one fused BCE-on-logits kernel (same value as F.binary_cross_entropy(sigmoid(l))
including torch's clamp of log at -100).  FeatureMatching: mean over the feature
list of L1(f_fake, f_real.detach()).  LSGAN/WGAN are not used by the hot-path
configs and raise."""
import torch.nn as nn

from csmri_hip import ops


def get_adversarial_loss(conf, loss_name, cuda, loss_type):
  assert loss_type in ('disc', 'gen')
  smoothing = conf.get_attr('discriminator_label_smoothing', default=0.0)
  if loss_name.upper() == 'GAN':
    return GANLoss(loss_type, cuda, smoothing)
  if loss_name in ('FeatureMatching', 'feature-matching'):
    distance_fn = conf.get_attr('feature_matching_loss_distance_function', default='L1')
    return FeatureMatchingLoss(loss_type, distance_fn)
  if loss_name.upper() in ('LSGAN', 'WGAN'):
    raise NotImplementedError('{} is outside the hot path (not used by the shipped configs)'.format(loss_name))
  raise ValueError('Unknown loss {}'.format(loss_name))


class GANLoss(nn.Module):
  def __init__(self, loss_type, cuda, disc_label_smoothing=0.0):
    super(GANLoss, self).__init__()
    assert 0.0 <= disc_label_smoothing < 1.0
    self.loss_type = loss_type
    self.gen_label = 1.0
    self.disc_real_label = 1.0 - disc_label_smoothing
    self.disc_fake_label = 0.0

  def forward(self, out_disc_fake, out_disc_real):
    if self.loss_type == 'gen':
      return ops.BCELogits.apply(out_disc_fake['logits'], self.gen_label)
    pair = out_disc_fake.get('_pair_logits')
    if pair is not None and pair is out_disc_real.get('_pair_logits') and pair.is_cuda and \
        {out_disc_fake.get('_pair_half'), out_disc_real.get('_pair_half')} == {0, 1}:
      # the two halves of ONE batched pass: one launch on the un-split logits (no slice nodes in the autograd graph:
      # their backward was two fills, two copies and an add).  Each half gets the label of the dict it came from --
      # whichever way round the two were stacked
      labels = [None, None]
      labels[out_disc_fake['_pair_half']] = self.disc_fake_label
      labels[out_disc_real['_pair_half']] = self.disc_real_label
      return ops.BCELogitsPair.apply(pair, labels[0], labels[1])
    return (ops.BCELogits.apply(out_disc_fake['logits'], self.disc_fake_label) +
            ops.BCELogits.apply(out_disc_real['logits'], self.disc_real_label))


class FeatureMatchingLoss(nn.Module):
  def __init__(self, loss_type, distance_fn):
    super(FeatureMatchingLoss, self).__init__()
    assert distance_fn in ('MSE', 'L1'), 'Unknown distance function {}'.format(distance_fn)
    self.kind = 0 if distance_fn == 'L1' else 1
    self.sign = 1.0 if loss_type == 'gen' else -1.0

  def forward(self, out_disc_fake, out_disc_real):
    ff, fr = out_disc_fake['features'], out_disc_real['features']
    chans = out_disc_fake['feature_channels']
    n = len(ff)
    if n <= 16 and all(a.dtype == b.dtype for a, b in zip(ff, fr)):
      # every layer's distance in one launch pair (and one backward launch)
      return ops.MultiMeanLoss.apply(self.kind, [self.sign / n] * n, list(chans), *ff,
                                     *[b.detach() for b in fr])
    loss = 0
    for a, b, c in zip(ff, fr, chans):
      loss = loss + ops.MeanLoss.apply(a, b.detach(), self.kind, c)
    return self.sign * loss / n
