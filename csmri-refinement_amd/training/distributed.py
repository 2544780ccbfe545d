"""Data parallelism: one process per GPU, gradients exchanged over RCCL (torch.distributed backend
'nccl' on ROCm) out of each model's flat fp32 gradient buffer, in sub-buckets, as bf16.

The reference's only parallelism is single-process nn.DataParallel
(utils/custom_data_parallel.py:6-35, utils/__init__.py:59-68): parameters are
re-broadcast every forward and gradients reduced to GPU 0.  Here every rank holds
replicas and each step moves every gradient element once as bf16 (GradBucket: direct
all-to-all + fp32 accumulation on the owning rank + all-gather; the mean's 1/N is folded into the
fused Adam kernel).  The discriminator's deep layers (convs.17/22: 90 % of its 27.9 M elements)
finish first in its backward and their sub-bucket leaves while the shallow layers are still
being differentiated; the rest overlaps the generator-side forward work.  BatchNorm statistics,
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


def _backend():
  return dist.get_backend() if dist.is_available() and dist.is_initialized() else None


# Tests (tests/test_distributed_gpu.py): run the complete exchange machinery -- comm stream, library pack / reduce /
# unpack kernels, the two RCCL collectives, the four-segment graphed step -- on ONE rank of the real 'nccl' backend,
# where every early-out below would otherwise skip it.  Set by the test (or CSMRI_FORCE_EXCHANGE=1), never by the product.
FORCE_EXCHANGE = bool(os.environ.get('CSMRI_FORCE_EXCHANGE'))


def exchange_active():
  """True when gradient buckets have to be exchanged: more than one rank (or the single-rank test mode)."""
  if world_size() > 1:
    return True
  return FORCE_EXCHANGE and dist.is_available() and dist.is_initialized()


class GradBucket(object):
  """Gradient exchange of one model: its flat fp32 gradient buffer, cut into SUB-BUCKETS (contiguous element
  ranges, listed in the order the backward finishes them) that are exchanged independently so the first ones
  travel while the backward still computes the rest.

  Transport per sub-bucket (``payload``):
    'bf16'  (default with more than one rank) reduce-scatter + all-gather written out over the DIRECT
            collectives: csmri_bucket_pack_bf16 (fp32 -> bf16, RNE) -> all_to_all of the N chunks (every pair of
            GPUs talks over its own xGMI link: all 7 links of an MI355X carry 1/8 of the bytes each, where a ring
            would push everything over one link per direction) -> csmri_bucket_reduce (the N received chunks summed
            in fp32 on the owning rank, rounded once) -> all_gather of the reduced chunks ->
            csmri_bucket_unpack_bf16 back into the fp32 buffer.  2 collectives + 3 library launches, half the bytes
            of an fp32 all-reduce, and no rank-count-dependent chain of bf16 roundings: every element is
            bf16(sum_r bf16(g_r)) with the sum carried in fp32.
    'fp32'  dist.all_reduce on the range (exact sums; CSMRI_GRAD_PAYLOAD=fp32).
  All device work of an exchange runs on a communication stream behind an event recorded when start() is
  called; wait() only makes the caller's stream wait for the exchanges' end events (the host never blocks
  on the nccl backend).  The 1/N of the mean is returned by wait() and folded into the fused Adam kernel.
  When nothing has been started early (graph mode: the collectives sit between the captured segments), start()
  exchanges the whole buffer as ONE range: 2 collectives + 3 launches per model and step."""

  TIMING = None          # bench.py: a list that collects (event, event) pairs around every wait() (exposed communication)

  def __init__(self, flat_grad, splits=None, payload=None):
    self.flat = flat_grad
    n = flat_grad.numel()
    self.splits = [(0, n)] if not splits else [(int(a), int(b)) for a, b in splits]
    assert self.splits[0][0] >= 0 and all(a < b for a, b in self.splits)
    assert sorted(self.splits) == sorted(set(self.splits)) and \
        sum(b - a for a, b in self.splits) == n and min(a for a, _ in self.splits) == 0 and \
        max(b for _, b in self.splits) == n, 'sub-buckets must tile the flat buffer'
    self.payload = payload or os.environ.get('CSMRI_GRAD_PAYLOAD', 'bf16')
    assert self.payload in ('bf16', 'fp32')
    self._started = [False] * len(self.splits)
    self._done = []                               # end-of-exchange events of the exchanges in flight
    self._stage = {}
    self._stream = None
    self.exchanges = 0                            # exchanges issued so far (tests)

  # -- one range ---------------------------------------------------------------------------------------
  def _buffers(self, a, b, world):
    per = ((b - a + world - 1) // world + 7) // 8 * 8
    key = (a, b, world)
    if key not in self._stage:
      dev = self.flat.device
      self._stage[key] = (torch.empty(world * per, dtype=torch.bfloat16, device=dev),
                          torch.empty(world * per, dtype=torch.bfloat16, device=dev),
                          torch.empty(per, dtype=torch.bfloat16, device=dev))
    return (per,) + self._stage[key]

  def _exchange(self, a, b):
    world = world_size()
    view = self.flat[a:b]
    self.exchanges += 1
    host_hop = _backend() == 'gloo' and view.is_cuda     # functional tests: several ranks on one GPU over gloo
    if self.payload == 'fp32':
      if host_hop:
        t = view.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        view.copy_(t)
      else:
        dist.all_reduce(view, op=dist.ReduceOp.SUM)
      return
    per, send, recv, mine = self._buffers(a, b, world)
    n = b - a
    if view.is_cuda:
      from csmri_hip import lib
      st = torch.cuda.current_stream().cuda_stream
      lib.call('csmri_bucket_pack_bf16', view.data_ptr(), n, send.data_ptr(), world * per, st)
    else:                                                # CPU tensors (gloo unit tests of the arithmetic)
      send[:n].copy_(view)
      send[n:].zero_()
    if host_hop:
      s_h, r_h = send.cpu(), torch.empty(world * per, dtype=torch.bfloat16)
      dist.all_to_all_single(r_h, s_h)
      recv.copy_(r_h)
    elif world == 1 and not view.is_cuda:
      recv.copy_(send)
    else:
      dist.all_to_all_single(recv, send)
    if view.is_cuda:
      lib.call('csmri_bucket_reduce', recv.data_ptr(), world, per, mine.data_ptr(), st)
    else:
      mine.copy_(recv.view(world, per).float().sum(0))   # the N contributions to my chunk, summed in fp32
    if host_hop:
      g_h = torch.empty(world * per, dtype=torch.bfloat16)
      dist.all_gather_into_tensor(g_h, mine.cpu())
      send.copy_(g_h)
    else:
      dist.all_gather_into_tensor(send, mine)
    if view.is_cuda:
      lib.call('csmri_bucket_unpack_bf16', send.data_ptr(), n, view.data_ptr(), st)
    else:
      view.copy_(send[:n])

  def _issue(self, a, b):
    if not self.flat.is_cuda:
      self._exchange(a, b)
      return
    if self._stream is None:
      self._stream = torch.cuda.Stream()
    ready = torch.cuda.Event()
    ready.record()
    self._stream.wait_event(ready)
    with torch.cuda.stream(self._stream):
      self._exchange(a, b)
      done = torch.cuda.Event()
      done.record(self._stream)
    self._done.append(done)

  def start(self, i=None):
    """Begin the exchange of sub-bucket i (all not yet started ones when None).  The gradients of the range
    must have been ISSUED on the calling stream (or on streams it has joined)."""
    if not exchange_active():
      return
    if i is None and not any(self._started):
      self._started = [True] * len(self.splits)
      self._issue(0, self.flat.numel())                 # nothing left early: one exchange for the whole buffer
      return
    todo = [j for j in range(len(self.splits)) if not self._started[j]] if i is None else [i]
    for j in todo:
      if self._started[j]:
        continue
      self._started[j] = True
      self._issue(*self.splits[j])

  def wait(self):
    """Make the caller's stream wait for every started exchange; returns the factor the summed gradient
    must be scaled by (1/world)."""
    if exchange_active():
      self.start()                                        # anything nobody started early
      timing = GradBucket.TIMING
      timed = timing is not None and self.flat.is_cuda and not torch.cuda.is_current_stream_capturing()
      if timed:
        # EXPOSED communication: how long the consumer's stream stands at this wait (0 when the exchange ended under
        # the compute issued before it); bench.py sums these event pairs over the timed steps
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
      for ev in self._done:
        torch.cuda.current_stream().wait_event(ev)
      if timed:
        e1.record()
        timing.append((e0, e1))
      self._done = []
      self._started = [False] * len(self.splits)
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


def decorrelate_rng_streams(seed):
  """Data parallelism: after the replicas have been built from the SAME seed (identical initial weights, then
  broadcast), give every rank its own random streams for what stays per rank -- Dropout2d masks (the Philox key of
  models/discriminators.py is drawn from torch's CPU generator at first use) and the image pool's swap decisions
  (python `random`).  The reference's single-process DataParallel drew independent masks across the global batch
  (utils/custom_data_parallel.py:26-35); identical streams on every rank would apply the same masks and the same
  pool swaps to every shard.  No-op on one rank, so single-GPU runs keep the seeds of utils.set_random_seeds."""
  if world_size() <= 1:
    return
  import random
  r = rank()
  mixed = (int(seed) + 0x9E3779B97F4A7C15 * (r + 1)) % (2 ** 63 - 1)
  random.seed(mixed)
  torch.manual_seed(mixed)
