"""Epoch loop shared by the runners (reference training/base_runner.py:19-147).

Same contract: train_epoch(loader, epoch, summary_writer, steps_per_train_summary,
verbose) -> (losses, metrics) dicts of Metric; validate(loader, n) -> (data, losses,
metrics); the batch dict keys handed to a model are selected by the parameter
names of its forward().  Batches move host->device asynchronously (pinned memory
when the loader provides it) and, under data parallelism, each rank takes its
contiguous shard of the global batch."""
import inspect
import logging
from itertools import chain

import numpy as np
import torch

import utils
from metrics import accumulate_metric
from training import distributed as dist_utils


class BaseRunner(object):
  def __init__(self, cuda=''):
    self.cuda = cuda
    self.epoch = 0
    self.data_iter = None
    self.device = utils.device_for(cuda)

  def _get_loss_weights(self, weights_by_criterion, *args):
    """Weights in criteria order, adversarial first (base_runner.py:19-27)."""
    weights = [weights_by_criterion.get(name, 1.0) for criteria in args for name in criteria]
    if len(weights) == 0:
      return None
    return torch.from_numpy(np.array(weights, dtype=np.float32)).to(self.device)

  def _request_data(self, loader, volatile=False):
    try:
      batch = next(self.data_iter)
    except StopIteration:
      self.data_iter = None
      return None
    batch = dist_utils.shard_batch(batch)
    return {k: v.to(self.device, non_blocking=True) for k, v in batch.items()}

  def _get_model_input_fn(self, model, batch_transform=None):
    params = list(inspect.signature(model.forward).parameters)

    def input_fn(batch, use_batch_transform=True):
      if use_batch_transform and batch_transform is not None:
        batch = batch_transform(self, dict(batch))
      return [batch[name] for name in params]

    return input_fn

  def train_epoch(self, loader, epoch, summary_writer=None, steps_per_train_summary=1,
                  verbose=False):
    self.epoch = epoch
    n_batches = len(loader)
    epoch_losses, epoch_metrics = {}, {}
    self._set_train()
    self.data_iter = iter(loader)
    current = 0
    while current < n_batches:
      num, loss_metrics, data = self._train_step(loader)
      if num == 0:
        break
      current += num
      metrics = self._compute_train_metrics(data)
      del data
      for name, m in loss_metrics.items():
        accumulate_metric(epoch_losses, name, m)
      for name, m in metrics.items():
        accumulate_metric(epoch_metrics, name, m)
      if current % steps_per_train_summary == 0:
        s = '===> Epoch[{}]({}/{}): '.format(epoch, current, n_batches)
        s += ', '.join('{}: {}'.format(k, v) for k, v in loss_metrics.items())
        if verbose:
          s += '\n' + '\n'.join('     {}: {}'.format(k, v) for k, v in metrics.items())
        logging.info(s)
        if summary_writer is not None:
          step = n_batches * (epoch - 1) + current
          for name, m in chain(loss_metrics.items(), metrics.items()):
            summary_writer.add_scalar('train/{}'.format(name), m.value, step)
    return ({k: v.average() for k, v in epoch_losses.items()},
            {k: v.average() for k, v in epoch_metrics.items()})

  def validate(self, loader, num_batches_to_return=0):
    epoch_data, epoch_losses, epoch_metrics = [], {}, {}
    self._set_test()
    self.data_iter = iter(loader)
    with torch.no_grad():
      for _ in range(len(loader)):
        loss_metrics, data = self._val_step(loader)
        if data is None:
          break
        if len(epoch_data) < num_batches_to_return:
          epoch_data.append(utils.cpuify(_strip_internal(data)))
        metrics = self._compute_test_metrics(data)
        del data
        for name, m in loss_metrics.items():
          accumulate_metric(epoch_losses, name, m)
        for name, m in metrics.items():
          accumulate_metric(epoch_metrics, name, m)
    return (epoch_data, {k: v.average() for k, v in epoch_losses.items()},
            {k: v.average() for k, v in epoch_metrics.items()})

  def infer(self, loader):
    epoch_data = []
    self._set_test()
    self.data_iter = iter(loader)
    with torch.no_grad():
      for _ in range(len(loader)):
        _, data = self._val_step(loader, compute_metrics=False)
        if data is None:
          break
        epoch_data.append(utils.cpuify(_strip_internal(data)))
    return epoch_data

  # -- subclass hooks ----------------------------------------------------------
  def get_named_outputs(self, data):
    raise NotImplementedError('Subclasses must override get_named_outputs')

  def get_named_models(self):
    raise NotImplementedError('Subclasses must override get_named_models')

  def state_dict(self):
    raise NotImplementedError('Subclasses must override state_dict')

  def load_state_dict(self, state_dict):
    raise NotImplementedError('Subclasses must override load_state_dict')

  def epoch_beginning(self, epoch):
    pass

  def epoch_finished(self, epoch):
    pass


def _strip_internal(data):
  """Drop the internal device-layout views before handing results to callers."""
  def clean(o):
    if isinstance(o, dict):
      return {k: clean(v) for k, v in o.items() if not str(k).startswith('_') and k != 'features'
              and k != 'feature_channels'}
    if isinstance(o, (list, tuple)):
      return [clean(v) for v in o]
    return o
  return clean(data)
