"""World-size-2 gloo tests of the data-parallel plumbing (training/distributed.py):
batch sharding, flat gradient bucket all-reduce + averaging, scalar reduction,
parameter broadcast.  Runs on CPU; the compute kernels are not involved."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG


def _worker(rank, world, port, q):
  sys.path.insert(0, PKG)
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                    MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  from training import distributed as D
  assert D.init_from_env(backend='gloo') == world
  assert D.world_size() == world and D.rank() == rank
  # sharding: contiguous equal shares
  batch = {'inp': torch.arange(8.).reshape(8, 1), 'mask': torch.arange(8.).reshape(8, 1) * 2}
  sh = D.shard_batch(batch)
  assert sh['inp'].shape[0] == 4 and float(sh['inp'][0]) == 4.0 * rank and float(sh['mask'][-1]) == 2 * (4 * rank + 3)
  # gradient bucket: sum over ranks, scale = 1/world makes it the mean
  flat = torch.full((1000,), float(rank + 1))
  b = D.GradBucket(flat)
  b.start()
  scale = b.wait()
  assert abs(scale - 0.5) < 1e-12 and torch.allclose(flat * scale, torch.full((1000,), 1.5))
  # mean-of-shard-means equals the global mean loss for equal shards
  g = torch.Generator().manual_seed(0)
  full = torch.randn(8, 5, generator=g)
  local = full[rank * 4:(rank + 1) * 4].pow(2).mean().reshape(1)
  red = D.reduce_scalars(local.clone())
  assert torch.allclose(red, full.pow(2).mean().reshape(1), atol=1e-6)
  # broadcast
  lin = torch.nn.Linear(3, 2)
  with torch.no_grad():
    lin.weight.fill_(float(rank))
  D.broadcast_module(lin, 0)
  assert float(lin.weight.abs().max()) == 0.0
  # per-rank random streams for dropout keys / image-pool swaps: same seed in, different streams out, reproducibly
  import random
  D.decorrelate_rng_streams(1)
  mine = torch.tensor([float(torch.randint(0, 2 ** 30, (1,)).item()), float(random.randint(0, 2 ** 30))])
  both = [torch.zeros(2), torch.zeros(2)]
  dist.all_gather(both, mine)
  assert both[0][0] != both[1][0] and both[0][1] != both[1][1], both
  D.decorrelate_rng_streams(1)
  again = torch.tensor([float(torch.randint(0, 2 ** 30, (1,)).item()), float(random.randint(0, 2 ** 30))])
  assert torch.equal(mine, again)
  dist.barrier()
  dist.destroy_process_group()
  q.put((rank, 'ok'))


def test_gloo_world2_plumbing():
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = 29600 + os.getpid() % 200
  procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  for p in procs:
    p.join(180)
    assert p.exitcode == 0, p.exitcode
  got = sorted(q.get(timeout=5) for _ in range(2))
  assert got == [(0, 'ok'), (1, 'ok')]


def test_single_process_is_world_one():
  sys.path.insert(0, PKG)
  from training import distributed as D
  assert D.world_size() == 1 and D.rank() == 0
  b = {'inp': torch.zeros(4, 1)}
  assert D.shard_batch(b) is b
  bucket = D.GradBucket(torch.ones(4))
  bucket.start()
  assert bucket.wait() == 1.0


def test_presharded_synthetic_loader_holds_exactly_the_ranks_rows():
  """train.py gives every rank a loader that synthesises ONLY its contiguous share of the global batch
  (SyntheticLoader(shard=(rank, world))): the rows are bit for bit the ones shard_batch would cut out of the global
  batch (reference base_runner.py:29-41 hands the whole batch to DataParallel, which scatters it the same way), and
  the runner does not cut a presharded batch again."""
  sys.path.insert(0, PKG)
  from data.synthetic import SyntheticLoader
  from training import distributed as D
  whole = SyntheticLoader(4, 32, 32, 3, acc=4, seed=5, distinct=2, pin=False)
  assert not whole.presharded
  for r in range(2):
    mine = SyntheticLoader(4, 32, 32, 3, acc=4, seed=5, distinct=2, pin=False, shard=(r, 2))
    assert mine.presharded and mine.batch_size == 4 and len(mine) == 3
    for bw, bm in zip(whole, mine):
      want = D.shard_batch(bw, r, 2)
      assert set(bm) == set(want)
      for k in want:
        assert bm[k].shape[0] == 2 and torch.equal(bm[k], want[k]), (r, k)
  one = SyntheticLoader(4, 32, 32, 1, pin=False, shard=(0, 1))
  assert not one.presharded and next(iter(one))['inp'].shape[0] == 4
