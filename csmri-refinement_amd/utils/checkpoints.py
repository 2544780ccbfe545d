"""Checkpoints (reference utils/checkpoints.py:9-121).  A checkpoint is a torch pickle
{'conf', 'runner', 'epoch', 'best_val_metrics'}; 'runner' holds the runner's state_dict() -- the models'
state dicts under 'model' (standard runner) or 'generator' / 'discriminator' plus the optimizer states
(SURVEY A-12) -- so files written by the reference load here and the other way round (the Configuration
object inside unpickles into this package's utils.config.Configuration, same module path, same attribute
bag).  Same function names, arguments and return values as the reference."""
import logging
import os

import torch

from utils.checkpoint_paths import is_checkpoint_path


def _load(path):
  # weights_only=False: checkpoints carry the Configuration object (reference checkpoints.py:10-16)
  return torch.load(path, map_location='cpu', weights_only=False)


def save_checkpoint(log_file_path, conf, runner, epoch, best_val_metrics=None):
  state = {'conf': conf, 'runner': runner.state_dict(), 'epoch': epoch, 'best_val_metrics': best_val_metrics}
  torch.save(state, log_file_path)


def restore_checkpoint(checkpoint_path, runner, cuda=None):
  """Loads the runner state; returns {'conf', 'start_epoch', 'best_val_metrics'} (keys present as in the file)."""
  checkpoint = _load(checkpoint_path)
  if 'runner' in checkpoint:
    runner.load_state_dict(checkpoint['runner'])
  else:                                           # pre-runner checkpoints (reference :27-30)
    runner.load_state_dict({'model': checkpoint['model'], 'optimizer': checkpoint['optimizer']})
  state = {'conf': checkpoint['conf']}
  if 'epoch' in checkpoint:
    state['start_epoch'] = checkpoint['epoch']
  if 'best_val_metrics' in checkpoint:
    state['best_val_metrics'] = checkpoint['best_val_metrics']
  return state


def inference_checkpoint_from_training_checkpoint(checkpoint, runner_type):
  """Strip a training checkpoint down to the network that inference needs (reference :44-63)."""
  inference_net_by_runner_type = {'standard': 'model', 'adversarial': 'generator'}
  assert runner_type in inference_net_by_runner_type, 'Unknown runner_type {}'.format(runner_type)
  net = inference_net_by_runner_type[runner_type]
  assert net in checkpoint['runner'], 'Checkpoint does not support runner_type {}'.format(runner_type)
  return {'conf': checkpoint['conf'], 'runner': {net: checkpoint['runner'][net]}}


def prune_checkpoints(run_dir, num_checkpoints_to_retain=1):
  """Keep the newest checkpoints of a run directory (names sort by time stamp; reference :66-76)."""
  checkpoints = sorted(f for f in os.listdir(run_dir) if is_checkpoint_path(f))
  for f in checkpoints[:max(0, len(checkpoints) - num_checkpoints_to_retain)]:
    path = os.path.join(run_dir, f)
    try:
      os.remove(path)
    except OSError:
      logging.warning('Could not remove old checkpoint {}'.format(path))


def load_model_state_dict(checkpoint_path, model_key, cuda=None):
  checkpoint = _load(checkpoint_path)
  if 'runner' not in checkpoint:
    raise ValueError('Did not find runner in checkpoint {}. Old checkpoint?'.format(checkpoint_path))
  runner_state = checkpoint['runner']
  if model_key not in runner_state:
    raise ValueError('Did not find model {} in checkpoint {}'.format(model_key, checkpoint_path))
  return runner_state[model_key]


def initialize_pretrained_model(model_conf, model, cuda, conf_path):
  if not model_conf.has_attr('pretrained_weights'):
    return
  if model_conf.pretrained_weights is None:
    logging.info('Skipping loading pretrained weights for %s, as explicitly no checkpoint was given',
                 model_conf.name)
    return
  path, model_key = model_conf.pretrained_weights
  if not os.path.isabs(path) and conf_path is not None:
    path = os.path.join(os.path.dirname(conf_path), path)
  if not os.path.exists(path):
    logging.warning('pretrained weights %s not found; keeping the seeded initialisation '
                    '(the shipped config carries a placeholder path, SURVEY A-1)', path)
    return
  model.load_state_dict(load_model_state_dict(path, model_key, cuda))
  logging.info('Loaded pretrained weights from checkpoint %s, key %s', path, model_key)
