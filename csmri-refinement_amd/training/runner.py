"""Standard (single model, e.g. RecNet + MSE) runner -- reference training/runner.py.

_train_step: zero_grad -> forward -> criteria -> weighted sum -> backward (which runs
the adjoint of every data-consistency layer) -> all-reduce -> fused Adam."""

import torch

from metrics import get_metric_fn, get_loss_metric
from models import construct_model
from models.criteria import get_criterion
from training.base_runner import BaseRunner
from training.optimizers import get_optimizer
from training.lr_schedulers import get_lr_scheduler, is_pre_epoch_scheduler, is_post_epoch_scheduler
from training import distributed as dist_utils
from utils.checkpoints import initialize_pretrained_model
from utils.config import Configuration
import utils


def build_runner(conf, cuda, mode='train'):
  model_conf = Configuration.from_dict(conf.model, conf)
  model = construct_model(model_conf, model_conf.name, cuda)
  val_metric_fns = {name: get_metric_fn(conf, name, cuda, 'test')
                    for name in conf.get_attr('validation_metrics', default=[])}
  model = utils.cudaify(model, cuda)
  if mode != 'train':
    return Runner(model, cuda=cuda, val_metric_fns=val_metric_fns)
  criteria = {}
  if conf.has_attr('loss_name'):
    criteria[conf.loss_name] = get_criterion(conf, conf.loss_name, cuda)
  else:
    for loss_name in conf.losses:
      criteria[loss_name] = get_criterion(conf, loss_name, cuda)
  assert len(criteria) > 0, 'Need at least one loss to optimize something!'
  if model_conf.has_attr('pretrained_weights'):
    initialize_pretrained_model(model_conf, model, cuda, conf.file)
  dist_utils.broadcast_module(model)
  opt_conf = Configuration.from_dict(conf.optimizer, conf)
  optimizer = get_optimizer(opt_conf, opt_conf.name, model.parameters())
  lr_scheduler = None
  if opt_conf.has_attr('lr_scheduler'):          # reference training/runner.py:48-52
    lr_scheduler = get_lr_scheduler(opt_conf, opt_conf.lr_scheduler, optimizer)
  train_metric_fns = {name: get_metric_fn(conf, name, cuda, 'train')
                      for name in conf.get_attr('train_metrics', default=[])}
  return Runner(model, criteria, conf.get_attr('loss_weights', {}), optimizer, lr_scheduler, cuda,
                train_metric_fns, val_metric_fns)


class Runner(BaseRunner):
  def __init__(self, model, criteria=None, loss_weights=None, optimizer=None, lr_scheduler=None,
               cuda='', train_metric_fns=None, val_metric_fns=None, output_transform=None,
               train_input_batch_transform=None, test_input_batch_transform=None):
    super(Runner, self).__init__(cuda)
    self.model = model
    self.criteria = criteria or {}
    self.loss_weights = self._get_loss_weights(loss_weights or {}, self.criteria)
    self.optimizer = optimizer
    self.lr_scheduler = lr_scheduler
    # data parallelism: a sub-bucket of gradients leaves as soon as the backward has issued its last layer.  The hook
    # is installed for the duration of this runner's own backward only (_step_body), never process-wide
    self._grad_hook = optimizer.grad_ready if optimizer is not None and dist_utils.exchange_active() else None
    self.train_metric_fns = train_metric_fns or {}
    self.val_metric_fns = val_metric_fns or {}
    self.train_model_input_fn = self._get_model_input_fn(model, train_input_batch_transform)
    self.test_model_input_fn = self._get_model_input_fn(model, test_input_batch_transform)
    # loss weights are configuration constants: read back once, here (never under a graph capture)
    self._host_weights = [float(w) for w in self.loss_weights.detach().cpu()] if self.criteria else None

  def get_named_outputs(self, data):
    batch, out = data[0], data[1]
    pred = out['pred'] if isinstance(out, dict) else out
    return {'input': batch['inp'], 'prediction': pred, 'target': batch['target']}

  def get_named_models(self):
    return {'model': self.model}

  def state_dict(self):
    return {'model': self.model.state_dict(), 'optimizer': self.optimizer.state_dict()}

  def load_state_dict(self, state_dict):
    self.model.load_state_dict(state_dict['model'])
    from csmri_hip import ops
    ops.bump_weight_epoch()
    if self.optimizer is not None:
      assert 'optimizer' in state_dict, 'Incompatible checkpoint'
      self.optimizer.load_state_dict(state_dict['optimizer'])

  def __str__(self):
    return 'Model:\n' + str(self.model)

  def epoch_beginning(self, epoch):               # reference training/runner.py:140-142
    lr = self._lr()
    if is_pre_epoch_scheduler(self.lr_scheduler):
      self.lr_scheduler.step()
    self._after_lr_change(lr)

  def epoch_finished(self, epoch):                # reference training/runner.py:144-146
    lr = self._lr()
    if is_post_epoch_scheduler(self.lr_scheduler):
      self.lr_scheduler.step()
    self._after_lr_change(lr)

  def _lr(self):
    return self.optimizer.param_groups[0]['lr'] if self.optimizer is not None else None

  def _after_lr_change(self, old_lr):
    """Nothing to re-capture: the captured Adam kernel reads the learning rate from device memory
    (FlatAdam.lr_dev / sync_lr, csmri_adam_dev_lr); see AdversarialRunner._after_lr_change."""
    return None

  def predict(self, batch):
    return self.model(*self.train_model_input_fn(batch, use_batch_transform=False))

  def _step_body(self, batch):
    """zero_grad -> forward -> criteria -> weighted sum -> backward; returns (loss tensors, total, out)."""
    self.optimizer.zero_grad()
    out = self.model(*self.train_model_input_fn(batch))
    from csmri_hip import ops
    names, losses = [], []
    for name, criterion in self.criteria.items():
      names.append(name)
      losses.append(criterion(out, batch))
    if losses[0].is_cuda and len(losses) <= 16:
      if getattr(self, '_host_weights', None) is None:
        self._host_weights = [float(w) for w in self.loss_weights.detach().cpu()]
      total = ops.weighted_sum(losses, self._host_weights)
    else:
      total = torch.sum(torch.stack(losses) * self.loss_weights)
    # weight gradients stay on the main stream here (a side stream as in the adversarial runner measured 10.37 vs
    # 8.5 ms on C2: at batch 64 every kernel fills the chip and is HBM-bound, co-running only adds contention)
    ops.enable_wgrad_stream(False)
    ops.GRAD_READY_HOOK = self._grad_hook
    try:
      ops.backward_scalar(total)
      ops.join_wgrad_stream()
    finally:
      ops.GRAD_READY_HOOK = None
    return names, [l.detach() for l in losses], total.detach(), out

  def enable_graphs(self, example_batch, warmup=2):
    """Capture the step (reference training/runner.py:154-178) as hipGraphs for this batch shape and replay them per
    step: the RecNet step is ~150 short launches and eager issue leaves the GPU idle between them.  One rank: ONE graph
    (zero_grad, forward, backward, Adam).  Data parallelism: TWO graphs -- [zero_grad, forward, backward] and [Adam] --
    with the gradient exchange (RCCL collectives are not captured) issued eagerly between them; 1/world is a launch
    argument of the captured Adam kernel.
    ``warmup`` eager steps run first (allocator / pack caches); they are REAL optimizer updates on the example
    batch -- pass warmup=0 (after at least one eager step elsewhere) when that matters."""
    static = {k: v.detach().clone() for k, v in example_batch.items()}
    self._set_train()
    from csmri_hip import ops as _ops
    side = _ops.named_stream('warmup')
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      for _ in range(warmup):
        self._step_body(static)
        self.optimizer.start_allreduce()
        self.optimizer.step()
    torch.cuda.current_stream().wait_stream(side)
    from models.utils import prepare_packs_for_capture
    prepare_packs_for_capture(self.model)
    torch.cuda.synchronize()
    import gc
    gc.collect()
    cap = _ops.named_stream('capture')
    self.optimizer.sync_lr()                         # (the rate is a device scalar the captured kernel reads: never a node)
    split = dist_utils.exchange_active()
    g = torch.cuda.CUDAGraph()
    g2 = None
    with torch.cuda.graph(g, stream=cap, capture_error_mode='thread_local'):
      names, losses, total, out = self._step_body(static)
      if not split:
        self.optimizer.step()
    if split:
      self.optimizer._scale = 1.0 / dist_utils.world_size()       # (what wait_allreduce() returns on every step)
      g2 = torch.cuda.CUDAGraph()
      with torch.cuda.graph(g2, pool=g.pool(), stream=cap, capture_error_mode='thread_local'):
        self.optimizer.apply()
    self.optimizer.step_count -= 1                   # the capture pass executed nothing
    self._graph = {'graph': g, 'graph_adam': g2, 'static': static, 'names': names, 'losses': losses, 'total': total,
                   'out': out}
    return self

  def disable_graphs(self):
    if getattr(self, '_graph', None) is not None:
      from training.adversarial_runner import retire_graphs
      retire_graphs(self._graph)
    self._graph = None

  def __del__(self):
    try:
      if getattr(self, '_graph', None) is not None:
        from training.adversarial_runner import retire_graphs
        retire_graphs(self._graph)
    except Exception:       # interpreter shutdown
      pass

  def _train_step(self, loader):
    batch = self._request_data(loader)
    if batch is None:
      return 0, None, None
    G = getattr(self, '_graph', None)
    if G is not None:
      torch._foreach_copy_(list(G['static'].values()), [batch[k] for k in G['static']])
      self.optimizer.sync_lr()
      G['graph'].replay()
      if G['graph_adam'] is not None:
        self.optimizer.start_allreduce()
        self.optimizer.wait_allreduce()
        G['graph_adam'].replay()
      self.optimizer.step_count += 1
      names, losses, total, out, batch = G['names'], [l.clone() for l in G['losses']], G['total'].clone(), G['out'], G['static']
    else:
      names, losses, total, out = self._step_body(batch)
      self.optimizer.start_allreduce()
      self.optimizer.step()
    loss_metrics = {'loss_' + n: get_loss_metric(l) for n, l in zip(names, losses)}
    loss_metrics['loss'] = get_loss_metric(total)
    return 1, loss_metrics, (batch, out)

  def _val_step(self, loader, compute_metrics=True):
    batch = self._request_data(loader, volatile=True)
    if batch is None:
      return None, None
    out = self.model(*self.test_model_input_fn(batch))
    loss_metrics = {}
    if compute_metrics:
      for name, criterion in self.criteria.items():
        loss_metrics['loss_' + name] = get_loss_metric(criterion(out, batch).detach())
    return loss_metrics, (batch, out)

  def _compute_train_metrics(self, data):
    return {name: fn(data[1], data[0]) for name, fn in self.train_metric_fns.items()}

  def _compute_test_metrics(self, data):
    return {name: fn(data[1], data[0]) for name, fn in self.val_metric_fns.items()}

  def _set_train(self):
    self.model.train()

  def _set_test(self):
    self.model.eval()
