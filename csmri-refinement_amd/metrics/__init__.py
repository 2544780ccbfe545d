"""Metrics on the hot path: PSNR (the parity metric) and the discriminator's
binary accuracy (reference metrics/__init__.py:38-72, metrics/image_metrics.py:7-19,
metrics/scalar_metrics.py:11-53, rec_transforms.py:79-85, metrics/metric.py).

Values stay on the device until somebody reads ``.value`` -- one batched readback
per logging interval instead of the reference's blocking ``loss.data[0]`` per loss."""
import math

import torch

from csmri_hip import ops


class SharedVec(object):
  """A device vector that several metrics index into (all loss scalars of one training step).
  The sum of two shared vectors is computed once, however many of their elements are added:
  accumulating the per-step losses over an epoch costs one launch per step, not one per loss."""

  def __init__(self, t):
    self.t = t
    self._sum = None

  def plus(self, other):
    if self._sum is None or self._sum[0] is not other:
      self._sum = (other, SharedVec(self.t + other.t))
    return self._sum[1]

  def scaled(self, f):
    if self._sum is None or self._sum[0] != ('scale', f):
      self._sum = (('scale', f), SharedVec(self.t * f))
    return self._sum[1]


class VecRef(object):
  """Element ``idx`` of a SharedVec, usable where a 0-dim tensor or a float is."""

  def __init__(self, shared, idx):
    self.shared, self.idx = shared, idx

  def tensor(self):
    return self.shared.t[self.idx]

  def item(self):
    return self.tensor().item()

  def __float__(self):
    return float(self.item())

  def __add__(self, other):
    if isinstance(other, VecRef):
      if other.idx == self.idx:
        return VecRef(self.shared.plus(other.shared), self.idx)
      return self.tensor() + other.tensor()
    return self.tensor() + other

  __radd__ = __add__

  def __truediv__(self, n):
    return VecRef(self.shared.scaled(1.0 / n), self.idx)


class Metric(object):
  """Running mean; ``values`` may be a float, a 0-dim device tensor or an iterable."""
  higher_is_better = True

  def __init__(self, values):
    if isinstance(values, torch.Tensor) and values.dim() > 0:
      values = list(values.reshape(-1))
    if isinstance(values, (list, tuple)) or hasattr(values, '__next__'):
      vals = list(values)
      self._value = None
      self.sum_values = sum(vals) if vals else 0.0
      self.num_updates = len(vals)
    else:
      self._value = values
      self.sum_values = values
      self.num_updates = 1

  @staticmethod
  def _f(v):
    return float(v.item()) if isinstance(v, (torch.Tensor, VecRef)) else float(v)

  @property
  def value(self):
    if self._value is None:
      return self._f(self.sum_values) / max(self.num_updates, 1)
    return self._f(self._value)

  def __str__(self):
    a = abs(self.value)
    if a >= 1e-4:
      return '{:.4f}'.format(self.value)
    return '{:.8f}'.format(self.value) if a >= 1e-8 else '{:.12f}'.format(self.value)

  def accumulate(self, metric):
    self._value = metric._value
    self.sum_values = self.sum_values + metric.sum_values
    self.num_updates += metric.num_updates

  def average(self):
    return type(self)(self.sum_values / max(self.num_updates, 1))

  def __gt__(self, other):
    return self.value > other.value if self.higher_is_better else self.value < other.value


class MaxMetric(Metric):
  higher_is_better = True

  @property
  def worst_value(self):
    return MaxMetric(-float('inf'))


class MinMetric(Metric):
  higher_is_better = False

  @property
  def worst_value(self):
    return MinMetric(float('inf'))


def get_loss_metric(value):
  return MinMetric(value)


def accumulate_metric(dictionary, name, metric):
  if name in dictionary:
    dictionary[name].accumulate(metric)
  else:
    dictionary[name] = type(metric)(metric.sum_values) if metric._value is None else metric
    if metric._value is None:
      dictionary[name].num_updates = metric.num_updates


def _complex_nhwc(t):
  fast = getattr(t, '_nhwc', None)         # RecNet.forward leaves its device-layout result on the NCHW output
  if fast is not None and fast.dim() == 4 and fast.shape[-1] == 2 and fast.shape[0] == t.shape[0]:
    return fast.contiguous()
  if t.dim() == 4 and t.shape[-1] == 2 and t.shape[1] != 2:
    return t.detach().contiguous()
  return ops.nchw_to_nhwc(t.detach(), torch.float32, 2)


class PSNRMetric(object):
  """MetricFunction('psnr') of the reference on complex images: per image
  p = clamp(|pred|,0,1), t = clamp(|target|,0,1), 10 log10(1/mse(p,t)); mean over
  the batch.  One kernel produces the per-image MSEs."""

  def __call__(self, prediction, target, transform=True):
    if isinstance(prediction, dict):
      fast = prediction.get('_nhwc')
      prediction = fast['pred'] if fast is not None else prediction['pred']
    if isinstance(target, dict):
      target = target['target']
    mse = ops.psnr_mse(_complex_nhwc(prediction), _complex_nhwc(target))
    return MaxMetric(ops.psnr_mean(mse))             # mean_b 10 log10(1 / mse_b), one launch (double arithmetic)


class SSIMMetric(object):
  """MetricFunction('ssim') of the reference (metrics/__init__.py:139, image_metrics.py:22-42,
  pytorch_ssim/__init__.py:22-42): per image on clamp(|.|,0,1), gaussian 11x11 window; mean
  over the batch.  One tile kernel + one per-image sum."""

  def __call__(self, prediction, target, transform=True):
    if isinstance(prediction, dict):
      fast = prediction.get('_nhwc')
      prediction = fast['pred'] if fast is not None else prediction['pred']
    if isinstance(target, dict):
      target = target['target']
    vals = ops.ssim(_complex_nhwc(prediction), _complex_nhwc(target))
    return MaxMetric(vals.double().mean())


class DiscAccuracyMetric(object):
  """disc_accuracy (scalar_metrics.py:26-53): per-image mean probability, class =
  p > 0.5, accuracy against label 0 (fake) and/or 1 (real)."""

  def __init__(self, fake, real):
    self.fake, self.real = fake, real

  def __call__(self, prob_fake, prob_real, transform=False):
    ref = prob_fake if self.fake else prob_real
    if ref.is_cuda and ref.shape[0] <= 64 and (not (self.fake and self.real) or prob_fake.shape == prob_real.shape):
      return MaxMetric(ops.disc_accuracy(prob_fake if self.fake else None, prob_real if self.real else None))
    parts = []
    if self.fake:
      p = prob_fake.detach().reshape(prob_fake.shape[0], -1).mean(dim=1)
      parts.append((p > 0.5) == torch.zeros_like(p, dtype=torch.bool))
    if self.real:
      p = prob_real.detach().reshape(prob_real.shape[0], -1).mean(dim=1)
      parts.append((p > 0.5) == torch.ones_like(p, dtype=torch.bool))
    return MaxMetric(torch.cat(parts).float().mean())


def get_metric_fn(conf, metric_name, cuda, mode, pred_key='pred', target_key='target'):
  assert mode in ('train', 'test')
  if metric_name == 'psnr':
    return PSNRMetric()
  if metric_name == 'ssim':
    return SSIMMetric()
  if metric_name in ('binary_accuracy', 'accuracy_fake'):
    return DiscAccuracyMetric(True, False)
  if metric_name == 'accuracy_real':
    return DiscAccuracyMetric(False, True)
  if metric_name == 'accuracy':
    return DiscAccuracyMetric(True, True)
  raise NotImplementedError("metric '%s' is outside the hot path (SURVEY 8f: hfen / segmentation scores)"
                            % metric_name)
