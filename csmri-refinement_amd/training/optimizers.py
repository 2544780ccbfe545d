"""Optimizer factory (reference training/optimizers.py:5-24 -> torch.optim.Adam).

FlatAdam keeps all trainable parameters of a model in ONE contiguous fp32 buffer
(the Parameters become views), their gradients in a second one, and runs Adam as a
single fused kernel over the flat buffers; the same flat gradient buffer is the
RCCL all-reduce bucket.  Update rule and state_dict layout follow torch.optim.Adam
(eps 1e-8, no weight decay, no amsgrad)."""
import torch
import torch.nn as nn

from csmri_hip import ops
from training.distributed import GradBucket


class FlatAdam(object):
  def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8):
    self.params = [p for p in params if p.requires_grad]
    assert len(self.params) > 0, 'optimizer got no trainable parameters'
    dev = self.params[0].device
    if dev.type != 'cuda':
      raise RuntimeError('FlatAdam runs on the GPU only (construct it after moving the model)')
    self.lr, self.betas, self.eps = lr, tuple(betas), eps
    self.offsets, total = [], 0
    for p in self.params:
      self.offsets.append(total)
      total += (p.numel() + 3) // 4 * 4            # keep every view 16-byte aligned
    self.numel = total
    self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
    self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
    self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
    self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
    for p, off in zip(self.params, self.offsets):
      n = p.numel()
      self.flat_p[off:off + n].copy_(p.data.reshape(-1))
      p.data = self.flat_p[off:off + n].view_as(p)
      p.grad = self.flat_g[off:off + n].view_as(p)
    self.step_count = 0                       # host mirror of step_dev
    self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    # the learning rate as the kernel reads it (csmri_adam_dev_lr): param_groups[0]['lr'] is what schedulers move;
    # sync_lr() copies it here whenever it has changed -- before an eager step and before a graph replay
    self.lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=dev)
    self._lr_mirror = float(lr)
    self._scale = 1.0
    # sub-buckets for the gradient exchange, cut at conv weights from the END of the buffer (the backward
    # finishes the last layers first) whenever a tail of >= 4 M elements has accumulated: the
    # discriminator becomes [convs.22.. | convs.17.. | the rest], small models stay one bucket
    cuts, acc = [], 0
    for i in range(len(self.params) - 1, 0, -1):
      acc += self.params[i].numel()
      if acc >= (4 << 20) and self.params[i].dim() == 4 and self.offsets[i] > 0:
        cuts.append(i)
        acc = 0
    bounds = [0] + [self.offsets[i] for i in reversed(cuts)] + [total]
    splits = list(reversed([(bounds[j], bounds[j + 1]) for j in range(len(bounds) - 1)]))
    self.bucket = GradBucket(self.flat_g, splits)
    self._first_param_of_split = {id(self.params[i]): len(cuts) - 1 - k for k, i in enumerate(reversed(cuts))}
    self.param_groups = [{'lr': lr, 'betas': self.betas, 'eps': eps, 'weight_decay': 0,
                          'amsgrad': False, 'params': list(range(len(self.params)))}]
    # callable -> PackGroups holding packed copies of exactly these weights: a step then invalidates
    # only those (without it: the global epoch, i.e. every trainable network's packs)
    self.pack_groups = None
    # lazy_zero: zero_grad() only MARKS the gradients stale; the first weight-gradient launch of a parameter then
    # overwrites instead of accumulating (csmri_wgrad / csmri_bn_bwd_apply `accumulate = 0`), later ones accumulate
    # as usual, and whatever is still marked at apply() is zeroed there.  Same values as fill + accumulate; saves
    # the fill of the flat buffer (112 MB for the discriminator) and one read of it.  Applies to the parameters
    # the library's kernels have written before (they set `_kernel_grad`); the others are zeroed as usual.
    self.lazy_zero = False
    ops.bump_weight_epoch()

  def zero_grad(self):
    for p, off in zip(self.params, self.offsets):   # re-attach if something replaced .grad
      if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
        p.grad = self.flat_g[off:off + p.numel()].view_as(p)
    if self.lazy_zero:
      # parameters whose gradients the library's kernels write (they honour the mark: ops.conv_wgrad, the BatchNorm
      # backward) are only marked; whatever torch autograd accumulates into (e.g. the refinement wrapper's scale,
      # and everything before the first backward has shown who writes what) is zeroed now, in one launch
      rest = []
      for p in self.params:
        if getattr(p, '_kernel_grad', False):
          p._grad_fresh = True
        else:
          rest.append(p.grad)
      if len(rest) == len(self.params):
        self.flat_g.zero_()
      elif rest:
        torch._foreach_zero_(rest)
    else:
      self.flat_g.zero_()

  def start_allreduce(self):
    self.bucket.start()

  def grad_ready(self, layer):
    """ops.GRAD_READY_HOOK target: the weight-gradient launch of `layer` has been issued (on the weight-gradient
    side stream when one is in use).  When that layer's weight opens a sub-bucket, everything behind it in
    the buffer is final: start that sub-bucket's exchange now, under the rest of the backward."""
    j = self._first_param_of_split.get(id(layer.weight))
    if j is None or torch.cuda.is_current_stream_capturing():
      return
    from training.distributed import exchange_active
    if not exchange_active():
      return
    side = ops._WGRAD['stream']
    if side is not None:
      with torch.cuda.stream(side):
        self.bucket.start(j)
    else:
      self.bucket.start(j)

  def wait_allreduce(self):
    self._scale = self.bucket.wait()

  def sync_lr(self):
    """Bring the device-side learning rate up to date with param_groups[0]['lr'] (a fill on the current stream; nothing
    when the rate has not moved).  Not callable while the stream is capturing: the rate must not become a graph node."""
    lr = float(self.param_groups[0]['lr'])
    if lr != self._lr_mirror:
      assert not torch.cuda.is_current_stream_capturing(), 'learning rate changed inside a stream capture'
      self.lr_dev.fill_(lr)
      self._lr_mirror = lr

  def apply(self):
    """The Adam kernel itself (step counter and learning rate on the device: hipGraph-capturable)."""
    self.sync_lr()
    if self.lazy_zero:
      for p in self.params:                         # parameters no kernel wrote since zero_grad(): zero gradient
        if getattr(p, '_grad_fresh', False):
          p.grad.zero_()
          p._grad_fresh = False
    ops.adam_step_dev(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq,
                      self.lr_dev, self.betas[0], self.betas[1], self.eps,
                      self.step_dev, self._scale)
    self.step_count += 1
    groups = self.pack_groups() if self.pack_groups is not None else None
    if groups is None:
      ops.bump_weight_epoch()
    else:
      for g in groups:
        g.bump()

  def step(self):
    self.wait_allreduce()
    self.apply()

  # -- torch.optim.Adam compatible state ---------------------------------------
  def state_dict(self):
    state = {}
    if self.step_count > 0:
      for i, (p, off) in enumerate(zip(self.params, self.offsets)):
        n = p.numel()
        state[i] = {'step': torch.tensor(float(self.step_count)),
                    'exp_avg': self.exp_avg[off:off + n].view_as(p).clone(),
                    'exp_avg_sq': self.exp_avg_sq[off:off + n].view_as(p).clone()}
    return {'state': state, 'param_groups': [dict(g) for g in self.param_groups]}

  def load_state_dict(self, sd):
    for i, st in sd['state'].items():
      i = int(i)
      off, n = self.offsets[i], self.params[i].numel()
      self.exp_avg[off:off + n].copy_(st['exp_avg'].reshape(-1))
      self.exp_avg_sq[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
      self.step_count = int(st['step'])
    self.step_dev.fill_(self.step_count)
    if sd.get('param_groups'):
      self.param_groups[0]['lr'] = sd['param_groups'][0].get('lr', self.lr)


def get_optimizer(conf, optimizer_name, variables_or_model):
  if isinstance(variables_or_model, nn.Module):
    variables = variables_or_model.parameters()
    if isinstance(variables, dict):
      assert conf.has_attr('parameter_key'), 'Parameter key unspecfied, but model requires one.'
      variables = variables[conf.parameter_key]
  else:
    variables = variables_or_model
  if optimizer_name == 'Adam':
    return FlatAdam(variables, conf.learning_rate,
                    betas=(conf.get_attr('beta1', default=0.9), conf.get_attr('beta2', default=0.999)))
  if optimizer_name == 'RMSProp':
    raise NotImplementedError('RMSProp is outside the hot path (configs use Adam)')
  raise ValueError('Unknown optimizer {}'.format(optimizer_name))
