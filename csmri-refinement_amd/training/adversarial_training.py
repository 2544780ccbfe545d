"""Discriminator input function (reference training/adversarial_training.py:21-135).

Configured path: input_method 'simple-magnitude' (|complex image|), image pool of
generated images queried only for detached fake inputs.  The magnitude is computed
by one kernel straight into the discriminator's NHWC input layout."""
import torch

from csmri_hip import ops
from utils.image_pool import ImagePool

DEFAULT_INPUT_METHOD = 'simple'


def _as_complex_nhwc(t, detach):
  """[B,2,H,W] fp32 or internal [B,H,W,2] -> interleaved complex [B,H,W,2]."""
  if t.shape[-1] == 2 and t.dim() == 4 and t.shape[1] != 2:
    return t.detach() if detach else t
  if detach:
    return ops.nchw_to_nhwc(t.detach(), torch.float32, 2)
  return ops.ToNHWC.apply(t, torch.float32, 2)


def _build_input_fn(method, dtype_fn, image_pool=None, pool_label_swapping=False):
  if method not in ('simple-magnitude',):
    raise NotImplementedError("discriminator input_method '%s' is outside the hot path" % method)

  def _pred_of(prediction_or_target):
    if isinstance(prediction_or_target, dict):
      fast = prediction_or_target.get('_nhwc')
      return fast['pred'] if fast is not None else prediction_or_target['pred']
    return prediction_or_target

  def input_wrapper(prediction_or_target, inp, out_gen, is_real_input, detach=False,
                    pool_decisions=None, out=None, mag=None):
    """``out`` (detached inputs only): a dense [B,H,W,8] tensor of the compute dtype that receives the result --
    the runner hands the groups of one stacked batch to the fake / real calls instead of concatenating their
    results.  ``mag`` (detached inputs only): the magnitude image if the caller already has it (the same |pred|
    feeds the history pool and the generator-phase pass)."""
    pred = _pred_of(prediction_or_target)
    xc = _as_complex_nhwc(pred, detach)
    if detach and xc.is_cuda:
      pooled = image_pool is not None and (not is_real_input or pool_label_swapping) and image_pool.pool_size > 0
      if mag is None:
        mag = ops.complex_abs_raw(xc.contiguous(), dtype_fn(), 0, None if pooled else out)
      elif not pooled and out is not None:
        mag = out.copy_(mag)
      return image_pool.query(mag, pool_decisions, out) if pooled else mag
    assert out is None and mag is None
    mag = ops.ComplexAbs.apply(xc.contiguous(), dtype_fn(), 0)       # [B,H,W,8], channel 0
    if detach:
      mag = mag.detach()
      if image_pool is not None and (not is_real_input or pool_label_swapping):
        mag = image_pool.query(mag, pool_decisions)
    return mag

  def magnitude(prediction_or_target, out):
    """|image| into ``out`` ([B,H,W,8] of the compute dtype): no autograd, no history pool."""
    xc = _as_complex_nhwc(_pred_of(prediction_or_target), True)
    return ops.complex_abs_raw(xc.contiguous(), dtype_fn(), 0, out)

  def link(prediction_or_target, filled):
    """The differentiable (is_real_input=False, detach=False) input whose values ``magnitude`` has ALREADY written
    to ``filled``: launches nothing, returns ``filled`` as a function of the prediction (backward: the magnitude's
    derivative)."""
    xc = _as_complex_nhwc(_pred_of(prediction_or_target), False)
    return ops.ComplexAbs.apply(xc.contiguous(), dtype_fn(), 0, [filled], True)

  input_wrapper.magnitude, input_wrapper.link = magnitude, link
  input_wrapper.out_dtype = dtype_fn

  return input_wrapper


def get_discriminator_input_fn(conf, disc_conf, no_pool=False, dtype_fn=None):
  from models.utils import default_compute_dtype
  image_pool = None
  if disc_conf.get_attr('use_image_pool', default=False) and not no_pool:
    pool_size = disc_conf.get_attr('image_pool_size', default=5 * conf.batch_size)
    image_pool = ImagePool(pool_size, disc_conf.get_attr('image_pool_sample_prob', default=0.5))
  for key in ('normalize_input', 'scale_input_zero_one', 'strip_bg_class'):
    if disc_conf.get_attr(key, default=False):
      raise NotImplementedError("discriminator option '%s' is outside the hot path" % key)
  fn = _build_input_fn(disc_conf.get_attr('input_method', default=DEFAULT_INPUT_METHOD),
                       dtype_fn or default_compute_dtype, image_pool,
                       disc_conf.get_attr('image_pool_label_swapping', default=False))
  fn.image_pool = image_pool
  return fn
