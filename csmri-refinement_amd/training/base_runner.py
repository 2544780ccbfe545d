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
    if not getattr(loader, 'presharded', False):      # (a loader that holds only this rank's rows says so)
      batch = dist_utils.shard_batch(batch)
    return {k: v.to(self.device, non_blocking=True) for k, v in batch.items()}

  def _get_model_input_fn(self, model, batch_transform=None):
    params = list(inspect.signature(model.forward).parameters)

    def input_fn(batch, use_batch_transform=True):
      if use_batch_transform and batch_transform is not None:
        batch = batch_transform(self, dict(batch))
      return [batch[name] for name in params]

    return input_fn

  # -- the epoch passes ----------------------------------------------------------------------------------
  # One pass = a loop of steps over the loader with two name -> Metric tables (losses, metrics) that absorb each
  # step's values (metrics.accumulate_metric: device-side sums, one readback per report).  Contract of the reference
  # (training/base_runner.py:65-147): train_epoch -> (losses, metrics) averaged over the epoch, a step may consume
  # several batches (_train_step returns how many), a progress line every `steps_per_train_summary` batches and the
  # same scalars to the summary writer under 'train/<name>'; validate -> (first n batches on the host, losses,
  # metrics) under no_grad in eval mode; infer -> every batch on the host.

  @staticmethod
  def _absorb(table, named):
    for name, m in named.items():
      accumulate_metric(table, name, m)

  @staticmethod
  def _averaged(table):
    return {name: m.average() for name, m in table.items()}

  def _report(self, epoch, seen, total, losses, metrics, writer, verbose):
    head = '===> Epoch[{}]({}/{}): '.format(epoch, seen, total)
    logging.info(head + ', '.join('{}: {}'.format(*kv) for kv in losses.items()) +
                 ''.join('\n     {}: {}'.format(*kv) for kv in (metrics.items() if verbose else ())))
    if writer is not None:
      at = total * (epoch - 1) + seen
      for name, m in chain(losses.items(), metrics.items()):
        writer.add_scalar('train/{}'.format(name), m.value, at)

  def train_epoch(self, loader, epoch, summary_writer=None, steps_per_train_summary=1, verbose=False):
    self.epoch = epoch
    self._set_train()
    self.data_iter = iter(loader)
    total, seen = len(loader), 0
    sums = ({}, {})
    while seen < total:
      took, step_losses, data = self._train_step(loader)
      if not took:
        break                       # the loader ran dry inside the step
      seen += took
      step_metrics = self._compute_train_metrics(data)
      data = None
      self._absorb(sums[0], step_losses)
      self._absorb(sums[1], step_metrics)
      if seen % steps_per_train_summary == 0:
        self._report(epoch, seen, total, step_losses, step_metrics, summary_writer, verbose)
    return self._averaged(sums[0]), self._averaged(sums[1])

  def _eval_batches(self, loader, compute_metrics):
    """(loss_metrics, data) of every validation batch in eval mode (the caller holds torch.no_grad())."""
    self._set_test()
    self.data_iter = iter(loader)
    for _ in range(len(loader)):
      step_losses, data = self._val_step(loader, compute_metrics=compute_metrics)
      if data is None:
        return
      yield step_losses, data

  def validate(self, loader, num_batches_to_return=0):
    kept, sums = [], ({}, {})
    with torch.no_grad():
      for step_losses, data in self._eval_batches(loader, True):
        if len(kept) < num_batches_to_return:
          kept.append(utils.cpuify(_strip_internal(data)))
        self._absorb(sums[0], step_losses)
        self._absorb(sums[1], self._compute_test_metrics(data))
    return kept, self._averaged(sums[0]), self._averaged(sums[1])

  def infer(self, loader):
    with torch.no_grad():
      return [utils.cpuify(_strip_internal(data)) for _, data in self._eval_batches(loader, False)]

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
