"""History pool of generated images (reference utils/image_pool.py:8-60).

Same decision process -- python ``random``: one uniform draw per image once the pool is
full, then one randint for the slot; images are processed in batch order, so an image
stored by item i can be drawn by item j > i -- but the pool lives in ONE device buffer and
a query is one launch driven by small index tensors (csmri_image_pool_exchange; host tensors: three gathers +
one scatter in torch).  The host computes
the indices ("plan") from the decisions; the device part has a fixed shape and can be
captured into a hipGraph (the runner then only refreshes the index tensors per step)."""
import random

import torch


class ImagePool(object):
  def __init__(self, pool_size, p_pool_image=0.5):
    self.pool_size = pool_size
    self.p_pool_image = p_pool_image
    self.count = 0
    self.buffer = None            # [pool_size + 1, ...]; last slot is a write-only dummy
    self._idx = None              # device int64 [4, n]: kind, pool_idx, x_idx / wslot, wsrc
    self.external_plan = False    # True while a captured graph owns the device part

  # -- host side ---------------------------------------------------------------
  def decide(self, n):
    """Decisions for a batch of n images: list of (use_pool, slot)."""
    out, filled = [], self.count
    for _ in range(n):
      if filled < self.pool_size:
        out.append((False, filled))
        filled += 1
      elif random.uniform(0, 1) < self.p_pool_image:
        out.append((True, random.randint(0, self.pool_size - 1)))
      else:
        out.append((False, -1))
    return out

  def plan(self, decisions):
    """Sequential semantics resolved into gather/scatter indices (and self.count)."""
    n = len(decisions)
    kind, pidx, xidx = [0] * n, [0] * n, list(range(n))
    owner = {}                                   # slot -> index of the last writer in this batch
    filled = self.count
    for i, (use, slot) in enumerate(decisions):
      if use:
        if slot in owner:
          kind[i], xidx[i] = 2, owner[slot]      # drawn image was stored earlier in this batch
        else:
          kind[i], pidx[i] = 1, slot
        owner[slot] = i
      elif filled < self.pool_size:
        owner[filled] = i                        # filling phase: store, return the image itself
        filled += 1
    self.count = filled
    wslot = list(owner.keys()) + [self.pool_size] * (n - len(owner))
    wsrc = list(owner.values()) + [0] * (n - len(owner))
    return kind, pidx, xidx, wslot, wsrc

  def prepare(self, x, decisions=None):
    n = x.shape[0]
    if self.buffer is None:
      self.buffer = torch.zeros((self.pool_size + 1,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    if self._idx is None or self._idx.shape[1] != n:
      self._idx = torch.zeros(5, n, dtype=torch.int64, device=x.device)
    if decisions is None:
      decisions = self.decide(n)
    rows = self.plan(decisions)
    # A FRESH pinned staging tensor per step: the host runs many steps ahead of the GPU (graph replay,
    # no per-step read-back), so a persistent staging buffer would be rewritten by step t+k before the
    # asynchronous H2D copy of step t has executed.  The caching host allocator keeps a block busy
    # until the copy that reads it has completed (it records an event on the copying stream).
    host = torch.tensor(rows, dtype=torch.int64)
    if x.is_cuda:
      host = host.pin_memory()
    self._idx.copy_(host, non_blocking=True)

  # -- device side (fixed shape) -------------------------------------------------
  def apply_plan(self, x, out=None):
    if x.is_cuda and x.shape[0] <= 64 and (x[0].numel() * x.element_size()) % 16 == 0:
      # one launch (csmri_image_pool_exchange: at most 64 images of a multiple of 16 bytes) instead of three
      # gathers, two selects and a scatter; larger per-GPU batches take the index_select path below
      from csmri_hip import ops
      return ops.image_pool_exchange(x, self.buffer, self._idx, out)
    kind, pidx, xidx, wslot, wsrc = self._idx.unbind(0)
    from_pool = self.buffer.index_select(0, pidx)
    from_x = x.index_select(0, xidx)
    k = kind.view(-1, *([1] * (x.dim() - 1)))
    dst, out = out, torch.where(k == 1, from_pool, torch.where(k == 2, from_x, x))
    self.buffer.index_copy_(0, wslot, x.index_select(0, wsrc))
    return out if dst is None else dst.copy_(out)

  def query(self, image_batch, decisions=None, out=None):
    """``out``: optional tensor of the batch's shape that receives the result (and is returned)."""
    if self.pool_size == 0:
      return image_batch if out is None else out.copy_(image_batch)
    x = image_batch.detach()
    if not self.external_plan:
      self.prepare(x, decisions)
    return self.apply_plan(x, out)
