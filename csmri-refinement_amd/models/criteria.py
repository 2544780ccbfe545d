"""Criterion registry (reference models/criteria.py:8-128).

Ordinary criteria are called ``criterion(out_gen, batch)``, adversarial ones
``criterion(out_disc_fake, out_disc_real)``.  Every reduction is a csmri_loss
kernel call.  Criteria outside the hot-path configs (SmoothL1, CrossEntropy, NLL)
raise NotImplementedError."""
import torch
import torch.nn as nn

from csmri_hip import ops


class _MeanCriterion(nn.Module):
  """nn.MSELoss / nn.L1Loss equivalent.  ``nhwc=False``: [B,C,H,W] fp32 tensors
  (reference layout, converted on device); ``nhwc=True``: internal NHWC tensors,
  ``c_real`` real channels."""

  def __init__(self, kind):
    super(_MeanCriterion, self).__init__()
    self.kind = kind

  def forward(self, prediction, target, nhwc=False, c_real=None):
    if not nhwc and target is not None and prediction.is_cuda and prediction.shape == target.shape and \
        prediction.dtype == target.dtype == torch.float32 and prediction.is_contiguous() and \
        target.is_contiguous() and prediction.numel() % 4 == 0:
      # a mean over ALL elements does not care about the layout: both tensors as flat runs of 4 floats, no
      # conversion to the padded NHWC layout and back (RecNet MSE step: 4 conversion passes over the batch less)
      return ops.MeanLoss.apply(prediction.reshape(1, 1, -1, 4), target.detach().reshape(1, 1, -1, 4), self.kind, 4)
    if not nhwc:
      c_real = prediction.shape[1]
      prediction = ops.ToNHWC.apply(prediction, torch.float32, ops.pad8(c_real))
      if target is not None:
        target = ops.nchw_to_nhwc(target.detach(), torch.float32, ops.pad8(c_real))
    elif target is not None and target.shape != prediction.shape:
      target = ops.nchw_to_nhwc(target.detach(), prediction.dtype, prediction.shape[3])
    return ops.MeanLoss.apply(prediction, target, self.kind, c_real or prediction.shape[3])


def _mse():
  return _MeanCriterion(1)


def _l1():
  return _MeanCriterion(0)


def _unsupported(name):
  def ctor(*a, **k):
    raise NotImplementedError('criterion {} is outside the hot path'.format(name))
  return ctor


def _get_adv_criterion(conf, loss_name, cuda, target_key, loss_type):
  from models.adversarial_loss import get_adversarial_loss
  return get_adversarial_loss(conf, loss_name, cuda, loss_type)


def _get_vgg_criterion(conf, loss_name, cuda, target_key):
  from models.vgg_loss import VGGLoss
  vconf = conf.vgg_loss if conf.has_attr('vgg_loss') else {}
  vgg_loss = VGGLoss(loss_name, cuda, vconf.get('blocks', -1), vconf.get('criterion', 'MSE'),
                     vconf.get('weights'), seed=vconf.get('seed', 0),
                     weights_path=vconf.get('weights_path'), allow_random=vconf.get('allow_random'))
  return CriterionWrapper(vgg_loss, target_key, tap='vgg')


def _get_feature_penalty_criterion(conf, loss_name, cuda, target_key):
  assert conf.has_attr('feature_penalty'), \
      'Feature penalty loss needs additional config under key "feature_penalty"'
  assert 'input_key' in conf.feature_penalty, 'Feature penalty loss needs "input_key"'
  criterion = conf.feature_penalty.get('criterion', 'MSE')
  assert criterion in ('MSE', 'L1'), 'Unknown criterion {} for feature penalty loss'.format(criterion)
  return CriterionWrapperWithScalarTarget(_CRITERIA[criterion](), cuda, scalar_target=0.0,
                                          input_key=conf.feature_penalty['input_key'])


_CRITERIA = {
    'MSE': _mse, 'L1': _l1,
    'SmoothL1Loss': _unsupported('SmoothL1Loss'), 'CrossEntropy': _unsupported('CrossEntropy'),
    'NLLLoss': _unsupported('NLLLoss'),
    'GAN': _get_adv_criterion, 'LSGAN': _get_adv_criterion, 'WGAN': _get_adv_criterion,
    'FeatureMatching': _get_adv_criterion,
    'VGG19': _get_vgg_criterion, 'FeaturePenalty': _get_feature_penalty_criterion,
    'gan': _get_adv_criterion, 'lsgan': _get_adv_criterion, 'feature-matching': _get_adv_criterion,
}
_DIRECT = ('MSE', 'L1', 'SmoothL1Loss', 'CrossEntropy', 'NLLLoss')


def _select(out_gen, key, tap=None):
  """(tensor, is_internal_nhwc): prefer the device-layout tensor the model provides -- and among those the alias
  reserved for this consumer (``key@tap``) when the model hands one out."""
  if isinstance(out_gen, dict):
    fast = out_gen.get('_nhwc')
    if fast is not None and key in fast:
      return fast.get('%s@%s' % (key, tap), fast[key]), True
    return out_gen[key], False
  return out_gen, False


_NHWC_REAL_CHANNELS = {'pred': 2, 'prescaled_refinement': 1}


class CriterionWrapper(nn.Module):
  def __init__(self, criterion, target_key='target', input_key='pred', tap=None):
    super(CriterionWrapper, self).__init__()
    self.criterion, self.target_key, self.input_key, self.tap = criterion, target_key, input_key, tap

  def forward(self, out_gen, batch):
    pred, nhwc = _select(out_gen, self.input_key, self.tap)
    target = batch[self.target_key]
    if isinstance(self.criterion, _MeanCriterion):
      return self.criterion(pred, target, nhwc, _NHWC_REAL_CHANNELS.get(self.input_key))
    return self.criterion(pred, target)


class CriterionWrapperWithScalarTarget(CriterionWrapper):
  """Target is a constant; 0.0 is handled inside the kernel (b = NULL)."""

  def __init__(self, criterion, cuda, scalar_target, input_key='pred'):
    super(CriterionWrapperWithScalarTarget, self).__init__(criterion, input_key=input_key)
    if scalar_target != 0.0:
      raise NotImplementedError('only a zero scalar target is on the hot path')

  def forward(self, out_gen, batch):
    pred, nhwc = _select(out_gen, self.input_key)
    return self.criterion(pred, None, nhwc, _NHWC_REAL_CHANNELS.get(self.input_key))


def get_criterion(conf, loss_name, cuda, target_key=None, input_key=None, **kwargs):
  assert loss_name in _CRITERIA, 'Unknown loss {}'.format(loss_name)
  ctor = _CRITERIA[loss_name]
  if input_key is None:
    input_key = 'pred'
  if target_key is None:
    target_key = conf.get_attr('loss_target_keys', default={}).get(loss_name, 'target')
  if loss_name in _DIRECT:
    return CriterionWrapper(ctor(), target_key, input_key)
  return ctor(conf, loss_name, cuda, target_key, **kwargs)
