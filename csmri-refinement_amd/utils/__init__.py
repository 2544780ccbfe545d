"""Small helpers with the reference's names (utils/__init__.py:1-72)."""
import os
import random


def set_cuda_env(gpu_idx):
  """The build runs one process per GPU; for a multi-GPU spec ('0,1,...') the
  launcher in train.py spawns ranks, each pinned through LOCAL_RANK."""
  if gpu_idx and 'LOCAL_RANK' not in os.environ:
    os.environ.setdefault('HIP_VISIBLE_DEVICES', gpu_idx)
  return gpu_idx


def set_random_seeds(seed):
  import numpy as np
  import torch
  random.seed(seed)
  np.random.seed(seed)
  torch.manual_seed(seed)


def device_for(cuda):
  import torch
  if cuda == '':
    raise RuntimeError('csmri-refinement_amd has no CPU path: the training step runs on '
                       'libcsmri_hip.so (gfx950); pass -c 0')
  local = int(os.environ.get('LOCAL_RANK', '0'))
  return torch.device('cuda', local if torch.cuda.device_count() > local else 0)


def cudaify(obj, device_ids=None):
  """Move tensors / modules / containers to this rank's GPU."""
  import torch
  dev = device_for(device_ids if device_ids is not None else '0')
  if isinstance(obj, dict):
    return {k: cudaify(v, device_ids) for k, v in obj.items()}
  if isinstance(obj, (list, tuple)):
    return [cudaify(v, device_ids) for v in obj]
  if obj is None:
    return None
  if isinstance(obj, torch.Tensor):
    return obj.to(dev, non_blocking=True)
  return obj.to(dev)


def cpuify(obj):
  if isinstance(obj, dict):
    return {k: cpuify(v) for k, v in obj.items()}
  if isinstance(obj, (list, tuple)):
    return [cpuify(v) for v in obj]
  return obj.cpu() if obj is not None and hasattr(obj, 'cpu') else obj


def import_function_from_path(import_path):
  import importlib
  parts = import_path.split('.')
  module = importlib.import_module('.'.join(parts[:-1]))
  return getattr(module, parts[-1])
