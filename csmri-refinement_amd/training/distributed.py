"""Data parallelism: one process per GPU, gradients exchanged as ONE flat fp32
bucket per model over RCCL (torch.distributed backend 'nccl' on ROCm).

The reference's only parallelism is single-process nn.DataParallel
(utils/custom_data_parallel.py:6-35, utils/__init__.py:59-68): parameters are
re-broadcast every forward and gradients reduced to GPU 0.  Here every rank holds
replicas, each step moves exactly sum(numel) gradient elements once (all-reduce of
the flat gradient buffer, divided by world size inside the fused Adam kernel), and
the discriminator bucket is launched asynchronously right after the discriminator
backward so it overlaps the generator-side forward work.  BatchNorm statistics,
dropout masks and the image pool stay per rank."""
import os

import torch
import torch.distributed as dist


def world_size():
  return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
  return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def init_from_env(backend=None):
  """Initialise the default process group from RANK/WORLD_SIZE/MASTER_* (torchrun)."""
  ws = int(os.environ.get('WORLD_SIZE', '1'))
  if ws <= 1 or dist.is_initialized():
    return ws
  if backend is None:
    # CSMRI_DIST_BACKEND=gloo: several ranks on ONE GPU (functional tests of the multi-rank path)
    backend = os.environ.get('CSMRI_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
  os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
  os.environ.setdefault('MASTER_PORT', '29500')
  if backend == 'nccl':
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local if torch.cuda.device_count() > local else 0)
  dist.init_process_group(backend=backend, init_method='env://')
  return ws


def shard_batch(batch, r=None, n=None):
  """Rank r's contiguous share of a global batch dict (equal shards required so that
  the mean of per-rank mean-losses equals the global mean)."""
  r = rank() if r is None else r
  n = world_size() if n is None else n
  if n == 1:
    return batch
  out = {}
  for k, v in batch.items():
    b = v.shape[0]
    assert b % n == 0, 'global batch {} not divisible by {} ranks'.format(b, n)
    per = b // n
    out[k] = v[r * per:(r + 1) * per]
  return out


class GradBucket(object):
  """Asynchronous sum all-reduce of one flat gradient buffer."""

  def __init__(self, flat_grad):
    self.flat = flat_grad
    self.work = None

  def start(self):
    if world_size() > 1:
      self.work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)

  def wait(self):
    """Returns the factor the summed gradient must be scaled by (1/world)."""
    if self.work is not None:
      self.work.wait()
      self.work = None
    return 1.0 / world_size()


def reduce_scalars(values):
  """Mean over ranks of a 1-D tensor of scalars (loss logging)."""
  if world_size() > 1:
    dist.all_reduce(values, op=dist.ReduceOp.SUM)
    values = values / world_size()
  return values


def broadcast_module(module, src=0):
  """Make every rank start from rank src's parameters and buffers."""
  if world_size() == 1:
    return
  for t in list(module.parameters()) + list(module.buffers()):
    dist.broadcast(t.data, src)
