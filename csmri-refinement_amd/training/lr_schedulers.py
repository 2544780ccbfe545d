"""Learning-rate schedulers behind the reference's config keys (reference training/lr_schedulers.py:4-44:
``lr_scheduler: multistep | linear | polynomial`` with decay_steps / decay_factor / end_learning_rate /
start_decay / decay_power on the optimizer config).

The reference hands torch.optim.lr_scheduler.MultiStepLR / LambdaLR a torch optimizer; here the
optimizer is FlatAdam (one fused kernel over flat buffers), so the two schedules are restated in
closed form over ``optimizer.param_groups[*]['lr']`` with the same epoch bookkeeping: constructed with
``initial_epoch = -1`` the scheduler applies epoch 0 immediately, and the runner's epoch_beginning()
calls step() once per epoch, so training epoch e (1-based) runs at the epoch-e value."""
from bisect import bisect_right


class _Scheduler(object):
  def __init__(self, optimizer, last_epoch=-1):
    self.optimizer = optimizer
    if last_epoch == -1:
      for g in optimizer.param_groups:
        g.setdefault('initial_lr', g['lr'])
    else:
      for i, g in enumerate(optimizer.param_groups):
        if 'initial_lr' not in g:
          raise KeyError("param 'initial_lr' is not specified in param_groups[{}] when resuming".format(i))
    self.base_lrs = [g['initial_lr'] for g in optimizer.param_groups]
    self.last_epoch = last_epoch
    self.step()

  def get_lr(self):
    raise NotImplementedError

  def step(self, epoch=None):
    self.last_epoch = self.last_epoch + 1 if epoch is None else epoch
    for g, lr in zip(self.optimizer.param_groups, self.get_lr()):
      g['lr'] = lr

  def state_dict(self):
    return {k: v for k, v in self.__dict__.items() if k not in ('optimizer', 'lr_lambda')}

  def load_state_dict(self, sd):
    self.__dict__.update(sd)


class MultiStepLR(_Scheduler):
  """lr = base * gamma ** (number of milestones <= epoch)."""

  def __init__(self, optimizer, milestones, gamma=0.1, last_epoch=-1):
    if list(milestones) != sorted(milestones):
      raise ValueError('Milestones should be a list of increasing integers. Got {}'.format(milestones))
    self.milestones, self.gamma = list(milestones), gamma
    super(MultiStepLR, self).__init__(optimizer, last_epoch)

  def get_lr(self):
    return [b * self.gamma ** bisect_right(self.milestones, self.last_epoch) for b in self.base_lrs]


class LambdaLR(_Scheduler):
  """lr = base * lr_lambda(epoch)."""

  def __init__(self, optimizer, lr_lambda, last_epoch=-1):
    self.lr_lambda = lr_lambda
    super(LambdaLR, self).__init__(optimizer, last_epoch)

  def get_lr(self):
    return [b * self.lr_lambda(self.last_epoch) for b in self.base_lrs]


class ReduceLROnPlateau(object):
  """Only its TYPE is on the path (pre- vs post-epoch stepping, reference lr_schedulers.py:18-24); no
  shipped config constructs one."""


def _get_polynomial_decay(lr, end_lr, decay_epochs, from_epoch=0, power=1.0):
  # reference lr_schedulers.py:4-15 (epochs are zero indexed)
  end_epoch = float(from_epoch + decay_epochs)

  def lr_lambda(epoch):
    if epoch < from_epoch:
      return 1.0
    epoch = min(epoch, end_epoch)
    new_lr = ((lr - end_lr) * (1. - epoch / end_epoch) ** power + end_lr)
    return new_lr / lr
  return lr_lambda


def is_pre_epoch_scheduler(scheduler):
  return scheduler is not None and not isinstance(scheduler, ReduceLROnPlateau)


def is_post_epoch_scheduler(scheduler):
  return isinstance(scheduler, ReduceLROnPlateau)


def get_lr_scheduler(optimizer_conf, scheduler_name, optimizer, initial_epoch=-1):
  if scheduler_name == 'multistep':
    return MultiStepLR(optimizer, optimizer_conf.decay_steps, optimizer_conf.decay_factor, initial_epoch)
  elif scheduler_name == 'linear' or scheduler_name == 'polynomial':
    power = 1.0 if scheduler_name == 'linear' else optimizer_conf.decay_power
    lr_lambda = _get_polynomial_decay(optimizer_conf.learning_rate, optimizer_conf.end_learning_rate,
                                      optimizer_conf.decay_steps,
                                      optimizer_conf.get_attr('start_decay', default=0), power)
    return LambdaLR(optimizer, lr_lambda, initial_epoch)
  else:
    raise ValueError('Unknown learning rate scheduler {}'.format(scheduler_name))
