"""Two data-parallel ranks sharing the one GPU of the test box (gloo backend for the collectives,
CSMRI_DIST_BACKEND=gloo): the multi-rank training path end to end -- per-rank shards, hipGraph
segments with the eager gradient all-reduces between them, 1/N folded into the Adam kernel.
After two steps both ranks must hold bit-identical parameters (same summed gradients, same
update), different from the initial ones, and the step must equal an un-graphed 2-rank run."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _worker(rank, world, port, q, graphs):
  sys.path.insert(0, PKG)
  sys.path.insert(0, ROOT)
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                    MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), CSMRI_DIST_BACKEND='gloo',
                    HSA_ENABLE_IPC_MODE_LEGACY='0')
  import torch.distributed as dist
  from training import distributed as D
  assert D.init_from_env() == world
  torch.cuda.set_device(0)
  import csmri_hip  # noqa: F401
  from utils.config import Configuration
  from models.utils import set_default_compute_dtype
  from training import build_runner
  import utils
  from data.synthetic import synth_batch
  set_default_compute_dtype('bf16')
  conf = Configuration.from_json(os.path.join(PKG, 'configs', '2-refinement.json'))
  conf.batch_size = 2
  conf.vgg_loss = {'seed': 19}
  g, d = conf.generator_model, conf.discriminator_model
  g['pretrained_model']['num_filters'] = 8
  g['learnable_model']['encode_filters'] = [8, 16, 32]
  g['learnable_model']['decode_filters'] = [16, 8]
  d['num_filters_per_layer'] = [8, 16, 32, 64, 64, 64]
  utils.set_random_seeds(conf.seed)                    # same initial weights on every rank
  runner = build_runner(conf, 'adversarial', '0', 'train')
  dev = torch.device('cuda', 0)
  full = synth_batch(4, 128, 128, acc=4, seed=11)
  mine = {k: v.to(dev) for k, v in D.shard_batch(full).items()}
  p0 = torch.cat([p.detach().float().reshape(-1) for p in runner.disc.parameters()]).clone()

  class Loader(list):
    batch_size = 2
  torch.manual_seed(100 + rank)                        # per-rank dropout / pool draws
  runner.overlap_streams = bool(graphs)                # side streams (VGG branch, weight gradients) too
  if graphs:
    runner.enable_graphs(mine)
  runner.train_epoch(Loader([mine, mine]), 1)
  torch.cuda.synchronize()
  flat = torch.cat([p.detach().float().reshape(-1) for net in (runner.disc, runner.gen)
                    for p in net.parameters() if p.requires_grad]).cpu()
  moved = float((torch.cat([p.detach().float().reshape(-1) for p in runner.disc.parameters()]) - p0).abs().max())
  dist.barrier()
  dist.destroy_process_group()
  q.put((rank, flat.numpy().tobytes(), moved))     # bytes: no shared-memory handle to outlive us


@pytest.mark.gpu
@pytest.mark.parametrize('graphs', [False, True])
def test_two_ranks_one_gpu_stay_in_sync(graphs):
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = 29800 + os.getpid() % 150 + (50 if graphs else 0)
  procs = [ctx.Process(target=_worker, args=(r, 2, port, q, graphs)) for r in range(2)]
  for p in procs:
    p.start()
  got = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
  for p in procs:
    p.join(120)
    assert p.exitcode == 0, p.exitcode
  (r0, b0, m0), (r1, b1, m1) = got
  import numpy as np
  f0, f1 = torch.from_numpy(np.frombuffer(b0, dtype=np.float32).copy()), torch.from_numpy(np.frombuffer(b1, dtype=np.float32).copy())
  assert torch.isfinite(f0).all() and m0 > 0 and m1 > 0
  assert torch.equal(f0, f1), float((f0 - f1).abs().max())
