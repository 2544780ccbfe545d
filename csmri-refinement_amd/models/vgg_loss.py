"""VGG perceptual loss for complex images (reference models/vgg_loss.py:13-65).

|pred| and |target| -> 3 equal channels -> ImageNet normalisation -> VGG19
features -> criterion (MSE by default through criteria.py:15-28) between the
block features; the target branch carries no gradient.  Magnitude, replication
and normalisation are one kernel; the feature loss is a deterministic two-stage
reduction."""
import torch
import torch.nn as nn

from csmri_hip import ops

_KIND = {'L1': 0, 'MSE': 1}


_VGG_MULTI = __import__('os').environ.get('CSMRI_VGG_MULTI', '1') != '0'     # A/B knob

class VGGLoss(nn.Module):
  def __init__(self, loss_name, cuda, blocks=-1, criterion='L1', weights=None, seed=0,
               weights_path=None, allow_random=None):
    super(VGGLoss, self).__init__()
    if loss_name != 'VGG19':
      raise ValueError('Unknown VGG loss {}'.format(loss_name))
    from models.vgg import VGG19
    if blocks == -1:
      blocks = [VGG19.LAST_FEATURE_MAP]
    elif not isinstance(blocks, list):
      blocks = [blocks]
    self.vgg = VGG19(blocks, requires_grad=False, seed=seed)
    # The reference uses torchvision's ImageNet weights (models/vgg.py:35, vgg19(pretrained=True)),
    # which need a download.  ``vgg_loss.weights_path``: a torchvision-format state_dict file
    # (keys features.{i}.weight/bias, or this module's own blocks.{b}.{i}.*).  Without it the
    # extractor keeps seeded kaiming weights -- a different perceptual loss than the reference's,
    # hence the warning (or an error when the config says vgg_loss.allow_random = false).
    if weights_path:
      self.vgg.load_pretrained(weights_path)
    elif allow_random is False:
      raise RuntimeError('vgg_loss.weights_path is not set and vgg_loss.allow_random is false')
    else:
      import warnings
      warnings.warn('VGG19 perceptual loss runs on seeded random weights (seed %d): set '
                    'vgg_loss.weights_path to a torchvision vgg19 state_dict to reproduce the '
                    "reference's ImageNet-pretrained loss" % seed, stacklevel=2)
    self.kind = _KIND[criterion]
    self.weights = list(weights) if weights is not None else [1.] * len(blocks)
    assert len(self.weights) == len(blocks)

  def forward(self, prediction, target):
    """prediction / target: interleaved complex fp32 [B,H,W,2] (internal layout) or
    [B,2,H,W] fp32 (reference layout)."""
    if prediction.shape[-1] != 2:
      assert prediction.shape[1] == 2, 'only complex (2-channel) inputs are on the hot path'
      prediction = ops.ToNHWC.apply(prediction, torch.float32, 2)
    if target.shape[-1] != 2:
      target = ops.nchw_to_nhwc(target.detach(), torch.float32, 2)
    dt = self.vgg.dtype
    frozen = all(not p.requires_grad for p in self.vgg.parameters())
    if frozen and prediction.is_cuda and prediction.is_contiguous():
      # frozen extractor: one batched pass over [pred; target], the magnitudes written into its two halves
      p_feats, t_feats = self.vgg.features_pair(prediction, target.detach(), complex_input=True)
    elif frozen:
      p_in = ops.ComplexAbs.apply(prediction, dt, 3)
      with torch.no_grad():
        t_in = ops.ComplexAbs.apply(target.detach().contiguous(), dt, 3)
      p_feats, t_feats = self.vgg.features_pair(p_in, t_in)
    else:
      p_in = ops.ComplexAbs.apply(prediction, dt, 3)
      with torch.no_grad():
        t_in = ops.ComplexAbs.apply(target.detach().contiguous(), dt, 3)
      with torch.no_grad():
        t_feats = self.vgg.forward_nhwc(t_in)
      p_feats = self.vgg.forward_nhwc(p_in)
    n = len(p_feats)
    if _VGG_MULTI and n <= 16 and len({f.dtype for f in p_feats}) == 1:
      # every block's distance in one launch pair (and one backward launch) instead of three launches per block
      # in a chain between the VGG branch and the generator backward (same kernel FeatureMatchingLoss uses)
      return ops.MultiMeanLoss.apply(self.kind, [float(w) for w in self.weights], [f.shape[3] for f in p_feats],
                                     *p_feats, *[t.detach() for t in t_feats])
    loss = 0
    for wgt, pf, tf in zip(self.weights, p_feats, t_feats):
      loss = loss + wgt * ops.MeanLoss.apply(pf, tf.detach(), self.kind, pf.shape[3])
    return loss
