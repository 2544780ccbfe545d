"""RefinementWrapper: frozen pretrained RecNet + learnable U-Net whose output is
added to the min/max-scaled real channel (mode 'real-penalty-add').

Drop-in for reference models/refinement_wrapper.py (constructor arguments,
forward(inp,kspace,mask) -> dict pred/pretrained/prescaled_refinement/
scaled_refinement, ``parameters()`` yielding only trainable tensors, state-dict
keys scale / pretrained_model.* / learnable_model.*).  The scale / add / unscale
chain of refinement_wrapper.py:51-92,169-194 is one fused kernel (plus a
per-sample min/max reduction)."""
import inspect

import torch
import torch.nn as nn

from csmri_hip import ops
from models import construct_model as build_model
from models.utils import freeze
from utils.config import Configuration

REQUIRED_PARAMS = ['pretrained_model', 'learnable_model']
OPTIONAL_PARAMS = ['mode', 'input_mode', 'freeze_pretrained_model']
KEY_RENAMES = {'pretrained_model': 'pretrained_model_conf',
               'learnable_model': 'learnable_model_conf'}


def construct_model(conf, model_name, **kwargs):
  params = conf.to_param_dict(REQUIRED_PARAMS, OPTIONAL_PARAMS, KEY_RENAMES)
  for key in ('pretrained_model_conf', 'learnable_model_conf'):
    params[key] = Configuration.from_dict(params[key], conf)
  model = RefinementWrapper(**params)
  from utils.checkpoints import initialize_pretrained_model
  initialize_pretrained_model(params['pretrained_model_conf'], model.pretrained_model,
                              kwargs.get('cuda'), conf.file)
  if params.get('freeze_pretrained_model', True):
    freeze(model.pretrained_model)
  return model


class RefinementWrapper(nn.Module):
  def __init__(self, pretrained_model_conf, learnable_model_conf, mode='add', input_mode='input',
               mse_path_model_conf=None, freeze_pretrained_model=True,
               disable_strict_loading=False):
    super(RefinementWrapper, self).__init__()
    if mode != 'real-penalty-add':
      raise NotImplementedError("RefinementWrapper mode '%s' is outside the hot path "
                                "(configs/2-refinement.json uses 'real-penalty-add')" % mode)
    if input_mode != 'output':
      raise NotImplementedError("input_mode '%s' is outside the hot path" % input_mode)
    self.mode, self.input_mode = mode, input_mode
    self.freeze_pretrained_model = freeze_pretrained_model
    self.pretrained_model = build_model(pretrained_model_conf, pretrained_model_conf.name)
    self.learnable_model = build_model(learnable_model_conf, learnable_model_conf.name)
    self.scale = nn.Parameter(torch.zeros(1))       # refinement_wrapper.py:111: init 0
    sig = list(inspect.signature(self.pretrained_model.forward).parameters)
    if sig != ['inp', 'kspace', 'mask']:
      raise RuntimeError('Could not find fitting forward function with params {}'.format(sig))

  def parameters(self, recurse=True):
    """Only trainable parameters (frozen RecNet excluded) -- refinement_wrapper.py:146-162."""
    params = super(RefinementWrapper, self).parameters(recurse)
    if not self.freeze_pretrained_model:
      return params
    return filter(lambda p: p.requires_grad, params)

  def precompute(self, inp, kspace, mask):
    """The frozen pretrained reconstruction alone (no autograd): it depends only on the batch, so
    a runner may issue it for the NEXT batch while the current step trains (forward_with_pre)."""
    assert self.freeze_pretrained_model
    with torch.no_grad():
      pre = self.pretrained_model(inp.detach(), kspace.detach(), mask.detach())
    return pre.detach()

  def forward(self, inp, kspace, mask):
    if self.freeze_pretrained_model:
      pre = self.precompute(inp, kspace, mask)
    else:
      pre = self.pretrained_model(inp, kspace, mask)
    return self.forward_with_pre(inp, kspace, mask, pre)

  def forward_with_pre(self, inp, kspace, mask, pre):
    """forward() given the pretrained model's output for this batch."""
    unet = self.learnable_model
    x = ops.ToNHWC.apply(pre, unet.dtype, 8)
    u = unet.forward_nhwc(x)                                      # [B,H,W,8], channel 0
    pre_c = ops.nchw_to_nhwc(pre.detach(), torch.float32, 2)      # interleaved complex
    pred_c, scaled, pred_c2, u2 = ops.RefineCombine.apply(pre_c, u, self.scale)
    if not ops.FANIN_REFINE:
      pred_c2, u2 = pred_c, u
    return {
        'pred': ops.ToNCHW.apply(pred_c, 2),
        'pretrained': pre,
        'prescaled_refinement': ops.ToNCHW.apply(u, 1),
        'scaled_refinement': scaled.unsqueeze(1),
        # internal device-layout views for the fused criteria (not part of the API).  'pred@vgg' and
        # 'prescaled_refinement' are aliases with their own autograd edge into RefineCombine (models/criteria._select):
        # the gradients of the prediction's two consumers and of the refinement's two are summed inside its backward
        # kernel, not by add launches in front of it
        '_nhwc': {'pred': pred_c, 'pred@vgg': pred_c2, 'prescaled_refinement': u2},
    }
