"""File-name contract of a run directory (reference utils/checkpoint_paths.py): run directories
``{run_name}_{time}``, ``periodic-chkpt_{time}_{epoch}.pth``, ``best-chkpt_{time}_{epoch}_{metric:.4f}.pth``,
``config_{time}.json``, ``log_{mode}_{time}.txt`` with time = YYYY-MM-DD-hh-mm-ss; an existing name gets a
``.2``, ``.3`` ... suffix.  Tools that scan a run directory rely on these names, nothing else."""
import os
import re
import time as _time

CHKPT_EXT = 'pth'
CHKPT_REGEXP = re.compile(r'.+\.{}(\.[\d]+)?$'.format(CHKPT_EXT))


def _stamp():
  return _time.strftime('%Y-%m-%d-%H-%M-%S', _time.localtime())


def _unique(base_dir, name):
  base = os.path.join(base_dir, name)
  path, idx = base, 2
  while os.path.exists(path):
    path = '{}.{}'.format(base, idx)
    idx += 1
  return path


def get_run_dir(base_dir, run_name):
  return _unique(base_dir, '{}_{}'.format(run_name, _stamp()))


def get_config_path(run_dir):
  return _unique(run_dir, 'config_{}.json'.format(_stamp()))


def get_periodic_checkpoint_path(run_dir, epoch):
  return _unique(run_dir, 'periodic-chkpt_{}_{}.{}'.format(_stamp(), epoch, CHKPT_EXT))


def get_best_checkpoint_path(best_dir, epoch, metric):
  return _unique(best_dir, 'best-chkpt_{}_{}_{:.4f}.{}'.format(_stamp(), epoch, metric, CHKPT_EXT))


def get_logfile_path(run_dir, mode):
  return _unique(run_dir, 'log_{}_{}.txt'.format(mode, _stamp()))


def is_checkpoint_path(path):
  return CHKPT_REGEXP.match(path) is not None
