"""History pool of generated images (reference utils/image_pool.py:8-60).

Same decision process -- python ``random``: one uniform draw per image once the
pool is full, then one randint for the slot -- but the pool lives in ONE device
buffer [pool_size,1,H,W] and the swap is a batched gather/scatter instead of
per-image unsqueeze/cat/clone."""
import random

import torch


class ImagePool(object):
  def __init__(self, pool_size, p_pool_image=0.5):
    self.pool_size = pool_size
    self.p_pool_image = p_pool_image
    self.count = 0
    self.buffer = None

  def decide(self, n):
    """Host-side decisions for a batch of n images: list of (use_pool, idx)."""
    out = []
    filled = self.count
    for _ in range(n):
      if filled < self.pool_size:
        out.append((False, filled))
        filled += 1
      else:
        if random.uniform(0, 1) < self.p_pool_image:
          out.append((True, random.randint(0, self.pool_size - 1)))
        else:
          out.append((False, -1))
    return out

  def query(self, image_batch, decisions=None):
    if self.pool_size == 0:
      return image_batch
    x = image_batch.detach()
    if self.buffer is None:
      self.buffer = torch.empty((self.pool_size,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    if decisions is None:
      decisions = self.decide(x.shape[0])
    result = x.clone()
    # sequential semantics (an image stored by item i can be drawn by item j>i)
    for i, (use, idx) in enumerate(decisions):
      if self.count < self.pool_size and not use:
        self.buffer[self.count].copy_(x[i])
        self.count += 1
      elif use:
        result[i].copy_(self.buffer[idx])
        self.buffer[idx].copy_(x[i])
    return result
