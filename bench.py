#!/usr/bin/env python3
"""Headline benchmark: train slices/s of the 256x256 GAN-refinement step (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W

N>1 is launched by the driver as torch.distributed.run with one rank per GPU (RCCL).
A "step" is one AdversarialRunner._train_single_step on a batch of synthetic
undersampled k-space resident in HBM: generator forward (frozen 3-block RecNet with 3
data-consistency layers + U-Net), three discriminator forwards, two VGG19 forwards,
discriminator backward + Adam, generator backward (through D and VGG) + Adam, gradient
all-reduce.  Per-GPU batch is fixed at 8 slices (BASELINE C4: 64 over 8 GPUs) -> weak
scaling.  Rank 0 prints ONE JSON line.

Extra legs (rank 0, outside the timed region):
  roofline     per-launch HIP-event timing of the conv kernels over instrumented steps
               of the same workload; the dominant kernel's algorithmic FLOP/s vs the
               dense bf16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PFLOP/s)
  cpu_baseline the CPU oracle's (plain torch fp32) GAN step on a bounded sample
               (N=1 only), plus the PSNR of both paths on the same batch/weights.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'csmri-refinement_amd')
sys.path.insert(0, PKG)

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA, MI355X_MICROARCH.md chip table
GAN_GFLOP_PER_SLICE = 299.2   # algorithmic conv FLOPs of one 256^2 GAN step
SIZE, PER_GPU_BATCH = 256, 8
CPU_SAMPLE_SLICES, CPU_SAMPLE_STEPS = 4, 4


def parse():
  p = argparse.ArgumentParser()
  p.add_argument('--gpus', type=int, default=1)
  p.add_argument('--steps', type=int, default=20)
  p.add_argument('--warmup', type=int, default=5)
  p.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
  p.add_argument('--no-cpu-baseline', action='store_true')
  p.add_argument('--no-roofline', action='store_true')
  p.add_argument('--no-prefetch', action='store_true',
                 help='do not issue the frozen RecNet forward of the next batch during the current step')
  p.add_argument('--batch', type=int, default=PER_GPU_BATCH)
  p.add_argument('--no-graphs', action='store_true', help='eager launches instead of hipGraph replay')
  p.add_argument('--no-overlap', action='store_true', help='keep the VGG branch on the main stream')
  return p.parse_args()


def build_runner(dtype, batch):
  import torch
  from utils.config import Configuration
  from models.utils import set_default_compute_dtype
  from training import build_runner as _build
  import utils
  set_default_compute_dtype(dtype)
  conf = Configuration.from_json(os.path.join(PKG, 'configs', '2-refinement.json'))
  conf.batch_size = batch
  utils.set_random_seeds(conf.seed)
  return _build(conf, 'adversarial', '0', 'train'), conf


class DeviceLoader(object):
  """Synthetic batches already resident in HBM (cycled)."""

  def __init__(self, batches, n):
    self.batches, self.n, self.batch_size = batches, n, batches[0]['inp'].shape[0]

  def __len__(self):
    return self.n

  def __iter__(self):
    for i in range(self.n):
      yield self.batches[i % len(self.batches)]


def cpu_baseline(runner, host_batch, sample_b=CPU_SAMPLE_SLICES, steps=CPU_SAMPLE_STEPS):
  """The oracle's GAN step on the host cores, same weights, first `sample_b` slices."""
  sys.path.insert(0, os.path.join(ROOT, 'oracle'))
  import torch
  import csmri_oracle as O
  # small-batch conv2d on the host stops scaling (and then collapses) well before the
  # GPU box's 256 hardware threads; 32 is what the timing below uses
  torch.set_num_threads(min(32, os.cpu_count() or 1))

  def split(sd):
    P = {k: v.detach().cpu().clone() for k, v in sd.items() if 'running' not in k and 'num_batches' not in k}
    S = {k: v.detach().cpu().clone() for k, v in sd.items() if 'running' in k}
    return P, S

  PG, SG = split(runner.gen.state_dict())
  PD, SD = split(runner.disc.state_dict())
  PV = {k: v.detach().cpu().clone() for k, v in
        runner.gen_criteria['VGG19'].criterion.vgg.state_dict().items() if k.startswith('blocks')}
  PG = {k: (v.requires_grad_(True) if not k.startswith('pretrained_model') else v) for k, v in PG.items()}
  PD = {k: v.requires_grad_(True) for k, v in PD.items()}
  gopt = O.make_adam([v for v in PG.values() if v.requires_grad], 2e-4, 0.5, 0.999)
  dopt = O.make_adam(PD.values(), 2e-4, 0.5, 0.999)
  batch = {k: v[:sample_b].clone() for k, v in host_batch.items()}
  with torch.no_grad():
    out0 = O.refinement_forward(PG, {k: v.clone() for k, v in SG.items()}, batch['inp'], batch['kspace'],
                                batch['mask'], True)
  psnr_cpu = O.psnr_batch(out0['pred'], batch['target'])
  pool = O.ImagePool(80)
  O.gan_train_step(PG, SG, PD, SD, PV, gopt, dopt, batch, pool=pool)          # warm-up
  t0 = time.time()
  for _ in range(steps):
    O.gan_train_step(PG, SG, PD, SD, PV, gopt, dopt, batch, pool=pool)
  dt = time.time() - t0
  cpu_model = ''
  try:
    with open('/proc/cpuinfo') as f:
      for line in f:
        if line.startswith('model name'):
          cpu_model = line.split(':', 1)[1].strip()
          break
  except OSError:
    pass
  return {'value': round(sample_b * steps / dt, 4), 'unit': 'slices/s', 'cores': torch.get_num_threads(),
          'kind': 'port', 'cpu': cpu_model,
          'sample': 'oracle (plain torch fp32) GAN step, %d slices of 256x256, %d timed steps after 1 warm-up'
                    % (sample_b, steps)}, psnr_cpu


def roofline(runner, loader, steps=2):
  """Instrumented steps: HIP events around every conv-library launch."""
  import torch
  from csmri_hip import ops
  runner.disable_graphs()       # per-launch events need eager launches
  runner.overlap_streams = False  # ... on ONE stream: concurrent side-stream kernels would inflate the brackets
  runner.prefetch_pretrained = False
  ops.PROFILE = []
  runner.train_epoch(DeviceLoader(loader.batches, steps), 1)
  torch.cuda.synchronize()
  recs, ops.PROFILE = ops.PROFILE, None
  # an empty event pair costs this much by itself; it is REPORTED, not subtracted: with a kernel
  # between the markers most of it overlaps the kernel, and the raw brackets are what agrees with
  # rocprofv3's per-kernel average of a single-stream run (profiles/*_single_stream.csv)
  pairs = []
  for _ in range(200):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); b.record()
    pairs.append((a, b))
  torch.cuda.synchronize()
  ovh = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2] * 1e-3
  agg = {}
  for label, flops, e0, e1 in recs:
    a = agg.setdefault(label, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += flops
    a[2] += e0.elapsed_time(e1) * 1e-3
  table = {k: {'launches_per_step': v[0] // steps, 'gflop_per_step': round(v[1] / steps / 1e9, 2),
               'ms_per_step': round(v[2] / steps * 1e3, 4),
               'tflops': round(v[1] / v[2] / 1e12, 1) if v[2] > 0 else None} for k, v in agg.items()}
  dom = max(agg, key=lambda k: agg[k][2])
  n, fl, sec = agg[dom]
  achieved = fl / sec / 1e12
  # HBM bytes per launch of that kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this very
  # command (tools/pmc_bench.sh; FETCH_SIZE doubled per the gfx950 correction), committed under profiles/
  traffic, traffic_src = None, None
  tpath = os.path.join(ROOT, 'profiles', 'r01_pmc_bench_traffic.json')
  if os.path.exists(tpath):
    for name, rec in json.load(open(tpath)).items():
      if dom in name:
        traffic = round(rec['fetch_bytes_per_launch'] + rec['write_bytes_per_launch'])
        traffic_src = 'profiles/r01_pmc_bench_traffic.json (rocprofv3 --pmc, %d launches)' % rec['launches']
  rl = {'bound': 'mfma', 'kernel': dom, 'achieved': round(achieved, 2), 'peak': PEAK_BF16_TFLOPS,
        'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_BF16_TFLOPS, 4), 'traffic': traffic,
        'traffic_unit': 'B/launch (HBM, PMC)', 'traffic_source': traffic_src,
        'avg_launch_us': round(sec / n * 1e6, 2), 'event_pair_overhead_us': round(ovh * 1e6, 2),
        'launches': n // steps,
        'algorithmic_gflop_per_launch': round(fl / n / 1e9, 3)}
  conv_ms = sum(v[2] for v in agg.values()) / steps * 1e3
  return rl, table, conv_ms


def main():
  args = parse()
  import torch
  from training import distributed as dist_utils
  ws = dist_utils.init_from_env()
  rank = dist_utils.rank()
  local = int(os.environ.get('LOCAL_RANK', '0'))
  torch.cuda.set_device(local if torch.cuda.device_count() > local else 0)
  assert ws == max(1, args.gpus) or ws == 1, (ws, args.gpus)

  from data.synthetic import synth_batch
  runner, conf = build_runner(args.dtype, args.batch)
  host_batches = [synth_batch(args.batch, SIZE, SIZE, acc=4, seed=conf.seed + 97 * rank + 100000 * i)
                  for i in range(2)]
  dev = torch.device('cuda', torch.cuda.current_device())
  batches = [{k: v.to(dev) for k, v in b.items()} for b in host_batches]
  runner._request_data_orig = runner._request_data

  def request(loader, volatile=False):      # inputs are already this rank's shard, in HBM
    try:
      return next(runner.data_iter)
    except StopIteration:
      runner.data_iter = None
      return None
  runner._request_data = request

  # PSNR of the untrained generator on batch 0 (compared with the oracle below)
  runner._set_train()
  with torch.no_grad():
    out0 = runner.gen(batches[0]['inp'], batches[0]['kspace'], batches[0]['mask'])
  from metrics import PSNRMetric
  psnr_hip_all = PSNRMetric()(out0, batches[0]).value
  ns = min(CPU_SAMPLE_SLICES, args.batch)
  psnr_hip = PSNRMetric()({'pred': out0['pred'][:ns]}, {'target': batches[0]['target'][:ns]}).value
  # undo the BN running-stat update of that probe forward? it does not affect training outputs

  runner.overlap_streams = not args.no_overlap
  runner.prefetch_pretrained = not (args.no_prefetch or args.no_overlap)
  if not args.no_graphs:
    # capture the step once (3 eager steps inside); the timed region replays hipGraphs
    try:
      runner.enable_graphs(batches[0])
    except Exception as e:            # keep the measurement alive: eager launches, same kernels
      sys.stderr.write('bench: hipGraph capture failed (%r); running eager\n' % (e,))
      runner.disable_graphs()
      args.no_graphs = True
  loader = DeviceLoader(batches, args.warmup)
  if args.warmup > 0:
    runner.train_epoch(loader, 1)
  torch.cuda.synchronize()
  if ws > 1:
    torch.distributed.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  losses, metrics = runner.train_epoch(DeviceLoader(batches, args.steps), 1, steps_per_train_summary=10 ** 9)
  torch.cuda.synchronize()
  if ws > 1:
    torch.distributed.barrier()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  if ws > 1:
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())

  # the instrumented roofline pass trains too (its steps all-reduce): every rank takes part
  prefetch_on = bool(runner.prefetch_pretrained)
  rl_out = None
  if not args.no_roofline:
    rl_out = roofline(runner, loader)
  if ws > 1:
    torch.distributed.barrier()
  if rank != 0:
    return
  slices = ws * args.batch * args.steps
  value = slices / dt
  line = {
      'metric': 'train slices/sec, 256x256 GAN refinement step', 'value': round(value, 2), 'unit': 'slices/s',
      'n_gpus': ws, 'steps': args.steps, 'warmup': args.warmup,
      'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
      'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
      'prefetch': 'frozen RecNet forward of batch t+1 on a side stream during step t' if prefetch_on else None,
      'launch_mode': 'eager' if args.no_graphs else ('hipGraph replay (one graph per step)' if ws == 1 else
                                                      'hipGraph replay (4 segments, collectives eager)'),
      'config': {'workload': 'C3/C4 2-refinement GAN step: frozen RecNet(3,3,32)+3 DC, UNET, CNNDiscriminator, '
                             'VGG19 loss, Adam x2; 256x256, 4x Cartesian, %d slices/GPU' % args.batch,
                 'per_gpu_batch': args.batch, 'global_batch': ws * args.batch,
                 'parallelism': 'dp%d' % ws, 'image': [SIZE, SIZE]},
      'algorithmic_tflops': round(value * GAN_GFLOP_PER_SLICE / 1e3, 2),
      'final_losses': {k: round(v.value, 5) for k, v in losses.items()},
      'gen_psnr': round(metrics['gen_psnr'].value, 4) if 'gen_psnr' in metrics else None,
  }
  if rl_out is not None:
    rl, table, conv_ms = rl_out
    line['roofline'] = rl
    line['conv_kernels'] = table
    line['conv_ms_per_step'] = round(conv_ms, 3)
  if ws == 1 and not args.no_cpu_baseline:
    # fresh runner with the same seed = same initial weights as the HIP run started from
    ref_runner, _ = build_runner(args.dtype, args.batch)
    base, psnr_cpu = cpu_baseline(ref_runner, host_batches[0], sample_b=min(CPU_SAMPLE_SLICES, args.batch))
    line['cpu_baseline'] = base
    line['psnr_hip_db'] = round(psnr_hip, 5)
    line['psnr_cpu_db'] = round(psnr_cpu, 5)
    line['psnr_delta_db'] = round(abs(psnr_hip - psnr_cpu), 5)
    line['gpu_over_cpu'] = round(value / base['value'], 1)
  print(json.dumps(line))


if __name__ == '__main__':
  main()
