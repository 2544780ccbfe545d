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


# ---------------------------------------------------------------------------------------------
# SURVEY 8(e) parity: 2 ranks against the CPU simulation of 2 replicas with averaged gradients
# ---------------------------------------------------------------------------------------------


def _masks_for(rank, B=2, chans=(64, 64, 64)):
  g = torch.Generator().manual_seed(500 + rank)
  return [(torch.rand(B, c, 1, 1, generator=g) < 0.5).float() * 2.0 for _ in range(3) for c in chans]


def _oracle_worker(rank, world, port, q, payload):
  sys.path.insert(0, PKG)
  sys.path.insert(0, ROOT)
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                    MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), CSMRI_DIST_BACKEND='gloo',
                    HSA_ENABLE_IPC_MODE_LEGACY='0', CSMRI_GRAD_PAYLOAD=payload)
  import numpy as np
  import torch.distributed as dist
  from training import distributed as D
  assert D.init_from_env() == world
  torch.cuda.set_device(0)
  import csmri_hip  # noqa: F401
  from csmri_hip import ops
  from utils.config import Configuration
  from models.utils import set_default_compute_dtype
  from training import build_runner
  from data.synthetic import synth_batch
  set_default_compute_dtype('fp32')
  conf = Configuration.from_json(os.path.join(PKG, 'configs', '2-refinement.json'))
  conf.batch_size = 2
  conf.vgg_loss = {'seed': 19}
  g, d = conf.generator_model, conf.discriminator_model
  for m in (g['pretrained_model'], g['learnable_model'], d):
    m['compute_dtype'] = 'fp32'
  g['pretrained_model']['num_filters'] = 8
  g['learnable_model']['encode_filters'] = [8, 16, 32]
  g['learnable_model']['decode_filters'] = [16, 8]
  d['num_filters_per_layer'] = [8, 16, 32, 64, 64, 64]
  runner = build_runner(conf, 'adversarial', '0', 'train')
  f = np.load(os.path.join(ROOT, 'tests', 'golden', 'F7_gan_step.npz'))
  sub = lambda pre: {k[len(pre):]: torch.from_numpy(np.asarray(v)) for k, v in f.items() if k.startswith(pre)}
  runner.gen.load_state_dict(sub('G0.'))
  runner.disc.load_state_dict(sub('D0.'))
  ops.bump_weight_epoch()
  full = synth_batch(4, 128, 128, acc=4, seed=11)       # the GLOBAL batch: the runner takes this rank's shard
  grads = {}

  def snap(opt, model, tag):
    orig, names = opt.apply, {id(p): n for n, p in model.named_parameters()}

    def apply():
      grads[tag] = {names[id(p)]: (p.grad.detach().float() * opt._scale).cpu().numpy() for p in opt.params}
      orig()
    opt.apply = apply
  snap(runner.gen_optimizer, runner.gen, 'G')
  snap(runner.disc_optimizer, runner.disc, 'D')
  runner.disc.injected_dropout = [m.clone() for m in _masks_for(rank)]

  class Loader(list):
    batch_size = 2
  losses, metrics = runner.train_epoch(Loader([full]), 1)
  torch.cuda.synchronize()
  vals = {k: float(v.value) for k, v in losses.items()}
  splits = list(runner.disc_optimizer.bucket.splits)
  dist.barrier()
  dist.destroy_process_group()
  q.put((rank, vals, grads if rank == 0 else None, splits))


@pytest.mark.gpu
@pytest.mark.parametrize('payload', ['fp32', 'bf16'])
def test_two_ranks_match_the_data_parallel_oracle(payload):
  """Two data-parallel ranks (one step, fp32 compute, reduced widths, injected per-rank dropout masks) against
  oracle.data_parallel_gan_step: two CPU replicas of the reference step whose gradients are averaged before each
  optimizer step (the reference's own multi-GPU path, utils/custom_data_parallel.py:26-35, is single-process
  DataParallel).  Per-rank losses as in the single-GPU F7 test (2e-5); the AVERAGED gradients the optimizers
  consume, per tensor: fp32 payload relative L2 <= 2e-2 / cos >= 0.9998 (the bound of the single-GPU gradient
  test), bf16 payload relative L2 <= 1e-2 on top of it (two bf16 roundings of 2^-9 around an fp32 sum)."""
  import numpy as np
  sys.path.insert(0, os.path.join(ROOT, 'oracle'))
  import csmri_oracle as O
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = 29950 + os.getpid() % 40 + (5 if payload == 'bf16' else 0)
  procs = [ctx.Process(target=_oracle_worker, args=(r, 2, port, q, payload)) for r in range(2)]
  for p in procs:
    p.start()
  got = sorted((q.get(timeout=900) for _ in range(2)), key=lambda t: t[0])
  for p in procs:
    p.join(120)
    assert p.exitcode == 0, p.exitcode
  f = np.load(os.path.join(ROOT, 'tests', 'golden', 'F7_gan_step.npz'))
  sub = lambda pre: {k[len(pre):]: torch.from_numpy(np.asarray(v)) for k, v in f.items() if k.startswith(pre)}
  small_unet = dict(O.UNET_CONF, encode_filters=[8, 16, 32], decode_filters=[16, 8])
  small_disc = dict(O.DISC_CONF, filters=[8, 16, 32, 64, 64, 64])
  u_def, d_def = O.unet_forward.__defaults__, O.disc_forward.__defaults__
  O.unet_forward.__defaults__ = tuple(small_unet if isinstance(x, dict) else x for x in u_def)
  O.disc_forward.__defaults__ = tuple(small_disc if isinstance(x, dict) else x for x in d_def)
  try:
    PV = O.init_vgg(gen=torch.Generator().manual_seed(19))
    full = O.synth_batch(4, 128, 128, acc=4, seed=11)
    shards = [{k: v[2 * r:2 * r + 2] for k, v in full.items()} for r in range(2)]
    reps = []
    for r in range(2):
      G0, D0 = sub('G0.'), sub('D0.')
      PG = {k: (v.clone().requires_grad_(True) if not k.startswith('pretrained_model') else v.clone())
            for k, v in G0.items() if 'running' not in k and 'num_batches' not in k}
      PD = {k: v.clone().requires_grad_(True) for k, v in D0.items() if 'running' not in k and 'num_batches' not in k}
      reps.append(dict(PG=PG, SG={k: v.clone() for k, v in G0.items() if 'running' in k},
                       PD=PD, SD={k: v.clone() for k, v in D0.items() if 'running' in k},
                       gen_opt=O.make_adam([v for v in PG.values() if v.requires_grad], 2e-4, 0.5, 0.999),
                       disc_opt=O.make_adam(PD.values(), 2e-4, 0.5, 0.999), pool=O.ImagePool(80)))
    dms = []
    for r in range(2):
      m = _masks_for(r)
      dms.append([m[0:3], m[3:6], m[6:9]])
    out, avg = O.data_parallel_gan_step(reps, PV, shards, dms)
  finally:
    O.unet_forward.__defaults__, O.disc_forward.__defaults__ = u_def, d_def
  for r in range(2):
    for k, v in out[r][0].items():
      assert abs(got[r][1][k] - v) < 2e-5 * max(1.0, abs(v)), (r, k, got[r][1][k], v)
  worst = 0.0
  for tag in ('G', 'D'):
    for k, gr in avg[tag].items():
      gh = torch.from_numpy(got[0][2][tag][k]).reshape(gr.shape)
      err = float((gh - gr).norm() / (gr.norm() + 1e-30))
      cos = float((gh * gr).sum() / (gh.norm() * gr.norm() + 1e-30))
      worst = max(worst, err)
      assert err < (2e-2 if payload == 'fp32' else 3e-2) and (gr.numel() == 1 or cos > 0.9995), (tag, k, err, cos)
  print('2-rank averaged gradients vs data-parallel oracle (%s payload): worst rel_l2 %.3e; D sub-buckets %s'
        % (payload, worst, got[0][3]))


# ---------------------------------------------------------------------------------------------
# The real RCCL backend ('nccl'), one rank: the exchange machinery end to end on device collectives
# ---------------------------------------------------------------------------------------------


@pytest.mark.gpu
@pytest.mark.parametrize('world', [1, 2, 8])
@pytest.mark.parametrize('n', [8, 1000, 4099, 1 << 20])
def test_bucket_kernels_match_the_torch_arithmetic(world, n):
  """csmri_bucket_pack_bf16 / _reduce / _unpack_bf16 against the torch expressions they replace
  (cast, .float().sum(0) rounded once, widen): bit-exact, including ragged tails and the zeroed pad."""
  from csmri_hip import lib
  dev = torch.device('cuda', 0)
  g = torch.Generator(device='cpu').manual_seed(n + world)
  x = (torch.randn(n, generator=g) * 3).to(dev)
  per = ((n + world - 1) // world + 7) // 8 * 8
  send = torch.full((world * per,), 7.0, dtype=torch.bfloat16, device=dev)      # stale content must not survive
  st = torch.cuda.current_stream().cuda_stream
  lib.call('csmri_bucket_pack_bf16', x.data_ptr(), n, send.data_ptr(), world * per, st)
  want = torch.zeros(world * per, dtype=torch.bfloat16, device=dev)
  want[:n] = x.to(torch.bfloat16)
  assert torch.equal(send.view(torch.int16), want.view(torch.int16))
  recv = (torch.randn(world, per, generator=g) * 3).to(torch.bfloat16).to(dev)
  mine = torch.empty(per, dtype=torch.bfloat16, device=dev)
  lib.call('csmri_bucket_reduce', recv.data_ptr(), world, per, mine.data_ptr(), st)
  acc = torch.zeros(per, dtype=torch.float32, device=dev)
  for r in range(world):                                   # rank order, fp32
    acc += recv[r].float()
  assert torch.equal(mine.view(torch.int16), acc.to(torch.bfloat16).view(torch.int16))
  out = torch.full((n + 3,), -1.0, device=dev)
  lib.call('csmri_bucket_unpack_bf16', send.data_ptr(), n, out.data_ptr(), st)
  assert torch.equal(out[:n], want[:n].float()) and bool((out[n:] == -1.0).all())
  assert lib.raw('csmri_bucket_pack_bf16')(x.data_ptr() + 4, n, send.data_ptr(), world * per, None) == -3   # CSMRI_E_ALIGN


class _NoHostSync(object):
  """Inside the block every host-blocking torch call raises: the exchange must stay asynchronous."""

  NAMES = [(torch.cuda, 'synchronize'), (torch.cuda.Stream, 'synchronize'), (torch.cuda.Event, 'synchronize'),
           (torch.Tensor, 'cpu'), (torch.Tensor, 'item'), (torch.Tensor, 'tolist')]

  def __enter__(self):
    self.saved = [(o, n, getattr(o, n)) for o, n in self.NAMES]

    def boom(*a, **k):
      raise AssertionError('host-blocking call inside the gradient exchange')
    for o, n, _ in self.saved:
      setattr(o, n, boom)

  def __exit__(self, *exc):
    for o, n, f in self.saved:
      setattr(o, n, f)
    return False


def _nccl_world1_worker(port, q):
  sys.path.insert(0, PKG)
  sys.path.insert(0, ROOT)
  os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                    HSA_ENABLE_IPC_MODE_LEGACY='0')
  os.environ.pop('CSMRI_DIST_BACKEND', None)
  import random
  import torch.distributed as dist
  torch.cuda.set_device(0)
  dist.init_process_group(backend='nccl', init_method='env://', world_size=1, rank=0)
  assert dist.get_backend() == 'nccl'
  from training import distributed as D
  res = {}
  dev = torch.device('cuda', 0)
  # -- 1. GradBucket alone: both payloads, sub-buckets and the merged range -------------------------------
  assert not D.exchange_active()
  D.FORCE_EXCHANGE = True
  assert D.exchange_active() and D.world_size() == 1
  gen = torch.Generator().manual_seed(3)
  g0 = (torch.randn(300000 + 4, generator=gen) * 1e-2).to(dev)
  for payload in ('bf16', 'fp32'):
    for early in (False, True):
      flat = g0.clone()
      b = D.GradBucket(flat, splits=[(200000, 300004), (0, 200000)], payload=payload)
      with _NoHostSync():
        if early:
          b.start(0)                         # a sub-bucket leaves from inside the backward
        b.start()
        scale = b.wait()
      torch.cuda.synchronize()
      want = g0.to(torch.bfloat16).float() if payload == 'bf16' else g0
      assert scale == 1.0 and torch.equal(flat, want), (payload, early, float((flat - want).abs().max()))
      assert b.exchanges == (2 if early else 1), (payload, early, b.exchanges)
  res['bucket'] = 'ok'
  # -- 2. the training step: dp1 vs forced exchange (eager, both payloads) and the 4-segment graphed step --
  import csmri_hip  # noqa: F401
  from utils.config import Configuration
  from models.utils import set_default_compute_dtype
  from training import build_runner
  import utils
  from data.synthetic import synth_batch
  set_default_compute_dtype('bf16')

  def make(payload):
    os.environ['CSMRI_GRAD_PAYLOAD'] = payload
    conf = Configuration.from_json(os.path.join(PKG, 'configs', '2-refinement.json'))
    conf.batch_size = 2
    conf.vgg_loss = {'seed': 19}
    g, d = conf.generator_model, conf.discriminator_model
    g['pretrained_model']['num_filters'] = 8
    g['learnable_model']['encode_filters'] = [8, 16, 32]
    g['learnable_model']['decode_filters'] = [16, 8]
    d['num_filters_per_layer'] = [8, 16, 32, 64, 64, 64]
    utils.set_random_seeds(conf.seed)
    return build_runner(conf, 'adversarial', '0', 'train')

  batch = {k: v.to(dev) for k, v in synth_batch(2, 128, 128, acc=4, seed=11).items()}

  class Loader(list):
    batch_size = 2

  def run(runner, steps=2, graphs=False):
    torch.manual_seed(7)
    random.seed(7)
    runner.overlap_streams = bool(graphs)
    if graphs:
      runner.enable_graphs(batch)
    per_step = []
    for _ in range(steps):
      losses, _ = runner.train_epoch(Loader([batch]), 1)
      per_step.append({k: float(v.value) for k, v in losses.items()})
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().float().reshape(-1) for net in (runner.disc, runner.gen)
                      for p in net.parameters() if p.requires_grad]).cpu()
    return per_step, flat

  D.FORCE_EXCHANGE = False
  la, pa = run(make('bf16'))                                 # dp1: no exchange at all
  D.FORCE_EXCHANGE = True
  r32 = make('fp32')
  l32, p32 = run(r32)                                        # forced, exact payload: must be the dp1 run bit for bit
  assert r32.disc_optimizer.bucket.exchanges >= 2 and r32.gen_optimizer.bucket.exchanges >= 2
  assert l32 == la and torch.equal(p32, pa), 'fp32 exchange at world 1 changed the run'
  r16 = make('bf16')
  l16, p16 = run(r16)
  assert l16[0] == la[0], (l16[0], la[0])                    # step 1's losses precede the first exchange
  for k, v in la[1].items():
    assert abs(l16[1][k] - v) <= 2e-2 * max(1.0, abs(v)), (k, l16[1][k], v)
  assert not torch.equal(p16, pa) and float((p16 - pa).abs().max()) < 1e-3
  res['eager'] = 'ok'
  rg = make('bf16')
  orig_start, orig_wait = D.GradBucket.start, D.GradBucket.wait

  def guarded_start(self, i=None):
    with _NoHostSync():
      return orig_start(self, i)

  def guarded_wait(self):
    with _NoHostSync():
      return orig_wait(self)
  lg, pg = None, None
  D.GradBucket.start, D.GradBucket.wait = guarded_start, guarded_wait
  try:
    lg, pg = run(rg, steps=3, graphs=True)
  finally:
    D.GradBucket.start, D.GradBucket.wait = orig_start, orig_wait
  assert len(rg._graph['graphs']) == 4, len(rg._graph['graphs'])     # collectives sit BETWEEN captured segments
  # 3 eager warm-up steps inside enable_graphs start D's two sub-buckets from the backward (2-3 exchanges each);
  # every replayed step exchanges each model's buffer as ONE range
  d_ex, g_ex = rg.disc_optimizer.bucket.exchanges, rg.gen_optimizer.bucket.exchanges
  assert g_ex == 3 + 3 and 3 + 3 <= d_ex <= 3 * len(rg.disc_optimizer.bucket.splits) + 3, (d_ex, g_ex)
  assert all(abs(v) < 1e4 and v == v for step in lg for v in step.values()) and torch.isfinite(pg).all()
  assert float((pg - pa).abs().max()) < 5e-3
  # every gradient the Adam kernels consumed went through the bf16 transport
  fg = rg.disc_optimizer.flat_g
  assert torch.equal(fg, fg.to(torch.bfloat16).float())
  res['graphs'] = 'ok'
  # -- 3. the standard (RecNet MSE) runner with data parallelism in GRAPH mode: [zero_grad, forward, backward] and [Adam]
  # as two hipGraphs with the eager gradient exchange between them (bench.py --config c2 with more than one rank) --
  def make_recnet(payload):
    os.environ['CSMRI_GRAD_PAYLOAD'] = payload
    conf = Configuration.from_json(os.path.join(PKG, 'configs', '1-recnet.json'))
    conf.model['num_blocks'], conf.model['num_convs'], conf.model['num_filters'] = 2, 3, 32
    conf.model['compute_dtype'] = 'bf16'
    conf.batch_size = 2
    utils.set_random_seeds(conf.seed)
    return build_runner(conf, 'standard', '0', 'train')

  def run_recnet(runner, graphs):
    if graphs:
      runner.enable_graphs(batch, warmup=1)                  # (one real eager step on the batch, then the capture)
    out = []
    for _ in range(3 if graphs else 4):
      losses, _ = runner.train_epoch(Loader([batch]), 1)
      out.append(float(losses['loss_MSE'].value))
    torch.cuda.synchronize()
    return out, torch.cat([p.detach().float().reshape(-1) for p in runner.model.parameters()]).cpu()
  D.FORCE_EXCHANGE = False
  le, pe = run_recnet(make_recnet('fp32'), False)            # dp1 eager
  D.FORCE_EXCHANGE = True
  rr = make_recnet('fp32')
  lgr, pgr = run_recnet(rr, True)
  assert rr._graph['graph_adam'] is not None and rr.optimizer.bucket.exchanges >= 3
  assert lgr == le[1:] and torch.equal(pgr, pe), 'two-graph data-parallel RecNet step differs from the dp1 eager run'
  res['recnet_graphs'] = 'ok'
  dist.barrier()
  dist.destroy_process_group()
  q.put(res)


@pytest.mark.gpu
def test_nccl_backend_forced_exchange_world1():
  """the gradient exchange on the REAL device collectives.  One rank of the 'nccl' (= RCCL)
  backend with training.distributed.FORCE_EXCHANGE: GradBucket (both payloads, early sub-buckets and the merged
  range) returns bf16(g) / g bit for bit without a single host-blocking call between start() and wait(); the eager
  step with the exact payload equals the dp1 run bit for bit; the graphed step becomes four hipGraph segments with
  the collectives between them."""
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  p = ctx.Process(target=_nccl_world1_worker, args=(29700 + os.getpid() % 90, q))
  p.start()
  p.join(900)
  assert p.exitcode == 0, p.exitcode
  assert q.get(timeout=10) == {'bucket': 'ok', 'eager': 'ok', 'graphs': 'ok', 'recnet_graphs': 'ok'}


# ---------------------------------------------------------------------------------------------
# the first test a multi-GPU box runs: two ranks, two GPUs, the real 'nccl' (RCCL) backend
# ---------------------------------------------------------------------------------------------


def _nccl_world2_worker(rank, world, port, q):
  sys.path.insert(0, PKG)
  sys.path.insert(0, ROOT)
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                    MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
  os.environ.pop('CSMRI_DIST_BACKEND', None)
  import torch.distributed as dist
  from training import distributed as D
  assert D.init_from_env() == world and dist.get_backend() == 'nccl'
  dev = torch.device('cuda', rank)
  torch.cuda.set_device(dev)
  # bucket arithmetic across two GPUs: every element bf16(sum_r bf16(g_r)) / the exact fp32 sum
  gen = torch.Generator().manual_seed(3)
  base = torch.randn(300000 + 4, generator=gen) * 1e-2
  mine = (base * (rank + 1)).to(dev)
  for payload in ('bf16', 'fp32'):
    flat = mine.clone()
    b = D.GradBucket(flat, splits=[(200000, 300004), (0, 200000)], payload=payload)
    b.start(0)
    b.start()
    scale = b.wait()
    torch.cuda.synchronize()
    if payload == 'bf16':
      want = (base.to(torch.bfloat16).float() + (base * 2).to(torch.bfloat16).float()).to(torch.bfloat16).float()
    else:
      want = base + base * 2
    assert scale == 0.5 and torch.equal(flat.cpu(), want), (payload, float((flat.cpu() - want).abs().max()))
  # the GAN step, graphed (four segments, collectives between them), per-rank shards: both ranks end bit-identical
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
  utils.set_random_seeds(conf.seed)
  runner = build_runner(conf, 'adversarial', str(rank), 'train')
  full = synth_batch(4, 128, 128, acc=4, seed=11)
  shard = {k: v.to(dev) for k, v in D.shard_batch(full).items()}
  p0 = torch.cat([p.detach().float().reshape(-1) for p in runner.disc.parameters()]).clone()

  class Loader(list):
    batch_size = 2
  runner.overlap_streams = True
  runner.enable_graphs(shard)
  runner.train_epoch(Loader([shard, shard]), 1)
  torch.cuda.synchronize()
  flat = torch.cat([p.detach().float().reshape(-1) for net in (runner.disc, runner.gen)
                    for p in net.parameters() if p.requires_grad]).cpu()
  moved = float((torch.cat([p.detach().float().reshape(-1) for p in runner.disc.parameters()]) - p0).abs().max())
  dist.barrier()
  dist.destroy_process_group()
  q.put((rank, flat.numpy().tobytes(), moved))


@pytest.mark.gpu
def test_nccl_backend_two_gpus():
  """Runs wherever two GPUs are visible (skipped on the one-GPU test boxes): the gradient buckets over RCCL between two
  devices -- all_to_all_single / all_gather_into_tensor with real inter-rank traffic, which a world of one cannot
  exercise -- and two graphed data-parallel GAN steps after which both ranks hold bit-identical parameters
  (reference utils/custom_data_parallel.py:26-35 is what this replaces)."""
  if torch.cuda.device_count() < 2:
    pytest.skip('needs two GPUs (the first multi-GPU box that sees this repository runs it before any benchmark)')
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = 29900 + os.getpid() % 90
  procs = [ctx.Process(target=_nccl_world2_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  got = sorted((q.get(timeout=900) for _ in range(2)), key=lambda t: t[0])
  for p in procs:
    p.join(120)
    assert p.exitcode == 0, p.exitcode
  import numpy as np
  (_, b0, m0), (_, b1, m1) = got
  f0, f1 = torch.from_numpy(np.frombuffer(b0, dtype=np.float32).copy()), torch.from_numpy(np.frombuffer(b1, dtype=np.float32).copy())
  assert torch.isfinite(f0).all() and m0 > 0 and m1 > 0
  assert torch.equal(f0, f1), float((f0 - f1).abs().max())
