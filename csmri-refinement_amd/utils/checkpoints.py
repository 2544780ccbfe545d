"""Checkpoint helpers on the hot path: pretrained-weight hand-off of the RecNet into
the RefinementWrapper (reference utils/checkpoints.py:9-41,96-121).  Checkpoints are
torch pickles {'conf','runner','epoch','best_val_metrics'}; runner state holds the
state dicts under 'model' / 'generator' / 'discriminator' (SURVEY A-12)."""
import logging
import os

import torch


def save_checkpoint(path, conf, runner, epoch, best_val_metrics=None):
  torch.save({'conf': conf, 'runner': runner.state_dict(), 'epoch': epoch,
              'best_val_metrics': best_val_metrics}, path)


def load_model_state_dict(checkpoint_path, model_key, cuda=None):
  ckpt = torch.load(checkpoint_path, map_location='cpu', weights_only=False)
  state = ckpt['runner']
  if model_key not in state:
    raise ValueError('Did not find model {} in checkpoint {}'.format(model_key, checkpoint_path))
  return state[model_key]


def restore_checkpoint(path, runner):
  ckpt = torch.load(path, map_location='cpu', weights_only=False)
  runner.load_state_dict(ckpt['runner'])
  return ckpt['conf'], ckpt['epoch'], ckpt.get('best_val_metrics')


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
