#!/usr/bin/env python3
"""Headline benchmark: train slices/s of the 256x256 GAN-refinement step (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W [--config c3|c2|c5] [--dtype bf16|fp32|fp8]

N>1 is launched by the driver as torch.distributed.run with one rank per GPU (RCCL).

--config c3 (default; BASELINE configs C3/C4): a "step" is one AdversarialRunner._train_single_step on
  8 slices/GPU of 256x256 synthetic undersampled k-space: generator forward (frozen 3-block RecNet with
  3 data-consistency layers + U-Net), three discriminator forwards, two VGG19 forwards, discriminator
  backward + Adam, generator backward (through D and VGG) + Adam, gradient all-reduce.  Weak scaling.
--config c5 (BASELINE config 5's data format; bf16, NOT the fp8 variant): the c3 step at 512x512 with
  golden-angle radial undersampling (70 spokes), 2 slices/GPU.
--config c2 (BASELINE config C2): one Runner._train_step of RecNet(5 blocks, 3 convs, 32 filters) with
  MSE loss on 64 slices/GPU of 256x256: 5 conv blocks + 5 data-consistency layers forward, their
  adjoints backward, Adam.

Input (the metric "includes H2D of the batch", reference training/base_runner.py:29-41): the batches
start in PINNED HOST memory (8 distinct batches cycled) and the H2D copy of batch t+1 is issued on a copy stream
while step t runs, INSIDE the timed region -- that is `value`.  The same K steps are then timed once more with the
batches resident in HBM (--device-resident makes that the only leg) and both rates are printed as `input_ab`
{host, resident}, so the PCIe cost is on the line.  After the W warm-up steps the replay settles for --settle-s
seconds (untimed; the step count is reported as `settle_steps` and included in `warmup_total_steps`), then EXACTLY
K steps are timed.  Rank 0 prints ONE JSON line; on the default invocation (c3, bf16, one GPU) it also carries
`other_configs`: short legs of C2, C5 and C5 with compute_dtype fp8.

Extra legs (rank 0, outside the timed region):
  roofline      HIP-event brackets around every conv-library launch over instrumented eager steps of the
                same workload; the dominant kernel's algorithmic FLOP/s vs the dense MFMA peak of the dtype
  roofline_hbm  the same brackets around the HBM-bound entry points (data consistency, BatchNorm passes,
                Adam) with their algorithmic bytes vs 8 TB/s
  cpu_baseline  the CPU oracle's (plain torch fp32) step on the SAME batch (N=1 only): >= 5 timed steps after a
                warm-up at the better of 32 / 64 threads (probed, one step each; --cpu-all-threads adds os.cpu_count());
                plus the PSNR of both paths on the same batch/weights.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'csmri-refinement_amd')
sys.path.insert(0, PKG)

PEAK_TFLOPS = {'bf16': 2500.0, 'fp32': 157.3, 'fp8': 5000.0}   # dense MFMA, MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0                           # HBM3E spec (same table; ~6.3 TB/s achievable)
GAN_GFLOP_PER_SLICE = 299.2   # algorithmic conv FLOPs of one 256^2 GAN step
C2_GFLOP_PER_SLICE = 20.31    # RecNet 5/3/32 MSE step at 256^2
SIZE = 256
DEFAULT_BATCH = {'c3': 8, 'c2': 64, 'c5': 2}
C5_GFLOP_PER_SLICE = 1196.8   # the GAN step at 512^2
C5_SPOKES = 70                # ~8x undersampling needs ~70 radial spokes at 512^2
N_HOST_BATCHES = 8


def parse():
  p = argparse.ArgumentParser()
  p.add_argument('--gpus', type=int, default=1)
  p.add_argument('--steps', type=int, default=250)
  p.add_argument('--warmup', type=int, default=10)
  p.add_argument('--config', default='c3', choices=['c3', 'c2', 'c5'])
  p.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32', 'fp8'])
  p.add_argument('--no-cpu-baseline', action='store_true')
  p.add_argument('--no-roofline', action='store_true')
  p.add_argument('--no-prefetch', action='store_true',
                 help='do not issue the frozen RecNet forward of the next batch during the current step')
  p.add_argument('--batch', type=int, default=0)
  p.add_argument('--no-graphs', action='store_true', help='eager launches instead of hipGraph replay')
  p.add_argument('--no-overlap', action='store_true', help='keep the VGG branch on the main stream')
  p.add_argument('--cpu-all-threads', action='store_true',
                 help='also probe the CPU baseline at 64 and at os.cpu_count() threads (256 on the GPU box: ~3 min per step)')
  p.add_argument('--device-resident', action='store_true',
                 help='A/B: batches resident in HBM when the timed region starts (default: pinned host batches, '
                      'H2D on a copy stream INSIDE the timed region, as SURVEY 8d defines the metric)')
  p.add_argument('--copy-streams', type=int, default=0,
                 help='HIP streams the H2D copies of a batch are spread over (0 = by batch size: one, two from 64 MB on)')
  p.add_argument('--no-input-ab', action='store_true',
                 help='skip the second timed pass (batches resident in HBM) that fills `input_ab`')
  p.add_argument('--settle-s', type=float, default=0.6,
                 help='untimed graph-replay settling after the W warm-up steps, seconds (0 = none)')
  p.add_argument('--lookahead-last', action='store_true',
                 help='A/B: launch the look-ahead graph (frozen RecNet of batch t+1) AFTER the step graph instead of before')
  p.add_argument('--finish-multi', default='auto', choices=['auto', '1', '0'],
                 help='A/B: one slab-reduction launch per backward pass (csmri_wgrad_finish_multi) instead of per layer')
  p.add_argument('--inprocess-legs', action='store_true',
                 help='diagnostic: run the fp8 side leg in this process too (tools/segv_hunt.sh)')
  p.add_argument('--no-other-configs', action='store_true',
                 help='skip the short C2 and C5 legs attached to the default (c3, bf16, N=1) line')
  a = p.parse_args()
  if a.batch <= 0:
    a.batch = DEFAULT_BATCH[a.config]
  a.other_cpu_fast = False
  return a


def build_runner(config, dtype, batch):
  import warnings
  from utils.config import Configuration
  from models.utils import set_default_compute_dtype
  from training import build_runner as _build
  import utils
  set_default_compute_dtype(dtype)
  if config in ('c3', 'c5'):
    conf = Configuration.from_json(os.path.join(PKG, 'configs', '2-refinement.json'))
    kind = 'adversarial'
  else:
    conf = Configuration.from_json(os.path.join(PKG, 'configs', '1-recnet.json'))
    conf.model['num_blocks'], conf.model['num_convs'], conf.model['num_filters'] = 5, 3, 32
    conf.model['compute_dtype'] = dtype
    kind = 'standard'
  conf.batch_size = batch
  utils.set_random_seeds(conf.seed)
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')      # seeded VGG weights: documented in DESIGN.md (no ImageNet file offline)
    return _build(conf, kind, '0', 'train'), conf


NUMA_INFO = {}


class _near_gpu(object):
  """Run the enclosed host allocations on the CPUs of the GPU's NUMA node (first-touch places the pinned pages there),
  then restore the thread's affinity.  Best effort: does nothing when sysfs does not say where the GPU sits."""

  def __init__(self, device):
    self.device, self.old = device, None

  def __enter__(self):
    try:
      import torch
      props = torch.cuda.get_device_properties(self.device)
      bdf = '%04x:%02x:%02x.0' % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
      node = int(open('/sys/bus/pci/devices/%s/numa_node' % bdf).read())
      NUMA_INFO.update(gpu_pci=bdf, gpu_numa_node=node)
      if node < 0:
        return self
      cpus = set()
      for part in open('/sys/devices/system/node/node%d/cpulist' % node).read().strip().split(','):
        a, _, b = part.partition('-')
        cpus.update(range(int(a), int(b or a) + 1))
      self.old = os.sched_getaffinity(0)
      use = cpus & self.old
      if use:
        os.sched_setaffinity(0, use)
        NUMA_INFO['pinned_on_node'] = node
    except Exception as e:          # no sysfs entry, no permission: plain pinning
      NUMA_INFO['error'] = repr(e)
    return self

  def __exit__(self, *exc):
    if self.old is not None:
      try:
        os.sched_setaffinity(0, self.old)
      except OSError:
        pass
    return False


class PinnedHostLoader(object):
  """Synthetic batches in PINNED HOST memory; batch t+1 is copied host->device on a copy stream while
  step t runs (4 rotating device buffer sets).  The consumer's stream waits for the copy's event."""

  _pinned = {}          # host batches are pinned ONCE (page-locking ~1 GB takes most of a second)
  _resident = {}        # device copies of the resident A/B leg, made once
  COPY_STREAMS = 2      # (--copy-streams)

  def __init__(self, host_batches, n, device, resident=False):
    import torch
    self.n, self.batch_size = n, host_batches[0]['inp'].shape[0]
    self.resident = resident
    # keyed by id() -- and the entry KEEPS the list alive (first element of the tuple), so the id cannot be handed to the
    # next leg's list while the entry exists.  Round 5 keyed by the bare id of a list that died with its leg: a later
    # leg whose list landed on the same address got the earlier leg's batches.
    key = id(host_batches)
    if resident:
      if key not in PinnedHostLoader._resident:
        PinnedHostLoader._resident[key] = (host_batches, [{k: v.to(device) for k, v in b.items()} for b in host_batches])
      assert PinnedHostLoader._resident[key][0] is host_batches
      self.dev = PinnedHostLoader._resident[key][1]
      return
    if key not in PinnedHostLoader._pinned:
      with _near_gpu(device):        # page-lock on the GPU's own NUMA node (2-socket hosts: H2D from the far node is slower)
        PinnedHostLoader._pinned[key] = (host_batches, [{k: v.pin_memory() for k, v in b.items()} for b in host_batches])
    assert PinnedHostLoader._pinned[key][0] is host_batches
    self.host = PinnedHostLoader._pinned[key][1]
    # the tensors of a batch are copied on COPY_STREAMS streams in turn: one hipMemcpyAsync stream is served by one
    # SDMA engine (17-28 GB/s on the boxes measured), two move a 134 MB C2 batch in parallel
    self.copy_streams = [torch.cuda.Stream() for _ in range(max(1, PinnedHostLoader.COPY_STREAMS))]
    self.dev = [{k: torch.empty_like(v, device=device) for k, v in host_batches[0].items()} for _ in range(4)]
    self.ready = [[torch.cuda.Event() for _ in self.copy_streams] for _ in self.dev]
    self.primed = False

  @staticmethod
  def forget(host_batches):
    """End of a leg: release its pinned pages and resident copies (and with them the key)."""
    PinnedHostLoader._pinned.pop(id(host_batches), None)
    PinnedHostLoader._resident.pop(id(host_batches), None)

  def prime(self):
    """Steady state of the copy pipeline at the moment the clock starts: in a running epoch the copy of batch 0 was issued
    during the step before.  The iterator then issues the copy of batch t+1 during EVERY step t, the last one included
    (the batch the step after the window would train on), so a window of K steps still holds K batch copies."""
    if not self.resident and not self.primed:
      self._issue(0)
      self.primed = True
    return self

  def __len__(self):
    return self.n

  def _issue(self, i):
    import torch
    j = i % len(self.dev)
    # the buffer set was last read by step i-4; everything enqueued so far on the consumer's stream
    # (steps <= i-2) must be done before it is overwritten
    guard = torch.cuda.Event()
    guard.record()
    src = self.host[i % len(self.host)]
    items = list(self.dev[j].items())
    for c, st in enumerate(self.copy_streams):
      st.wait_event(guard)
      with torch.cuda.stream(st):
        for k, d in items[c::len(self.copy_streams)]:
          d.copy_(src[k], non_blocking=True)
        self.ready[j][c].record(st)

  def __iter__(self):
    import torch
    if self.resident:
      for i in range(self.n):
        yield self.dev[i % len(self.dev)]
      return
    if not self.primed:
      self._issue(0)
    for i in range(self.n):
      if i + 1 < self.n or self.primed:
        self._issue(i + 1)
      for ev in self.ready[i % len(self.dev)]:
        torch.cuda.current_stream().wait_event(ev)
      yield self.dev[i % len(self.dev)]


def _cpu_model():
  try:
    with open('/proc/cpuinfo') as f:
      for line in f:
        if line.startswith('model name'):
          return line.split(':', 1)[1].strip()
  except OSError:
    pass
  return ''


def _split_sd(sd):
  P = {k: v.detach().cpu().clone() for k, v in sd.items() if 'running' not in k and 'num_batches' not in k}
  S = {k: v.detach().cpu().clone() for k, v in sd.items() if 'running' in k}
  return P, S


def _thread_counts(all_threads):
  """Thread counts the CPU baseline is probed at.  Small-batch conv2d stops scaling well before the box's hardware
  threads and then collapses: measured on the GPU boxes (2 x EPYC 9575F, 256 threads) 1.5-1.8 slices/s at 32 threads,
  1.0 at 64 (profiles/r04_bench_n1.json: every round's probe picked 32) and 0.046 at 256 (174 s per step,
  profiles/r02_bench_n1_a.json): the default probes 32 and 64 (one step each: the reported baseline is the best
  count on THIS box, not an assumed one), --cpu-all-threads adds os.cpu_count()."""
  n = os.cpu_count() or 1
  counts = {min(32, n), min(64, n)}
  if all_threads:
    counts.add(n)
  return sorted(counts)


def cpu_baseline_c3(runner, host_batch, steps=5, all_threads=False):
  """The oracle's GAN step on the host cores: same weights, the SAME batch (all slices).  One timed step per thread
  count of _thread_counts() picks the best count; the reported value is `steps` (>= 5) timed steps at
  that count."""
  sys.path.insert(0, os.path.join(ROOT, 'oracle'))
  import torch
  import csmri_oracle as O
  b = host_batch['inp'].shape[0]
  PV = {k: v.detach().cpu().clone() for k, v in
        runner.gen_criteria['VGG19'].criterion.vgg.state_dict().items() if k.startswith('blocks')}

  def fresh():
    PG, SG = _split_sd(runner.gen.state_dict())
    PD, SD = _split_sd(runner.disc.state_dict())
    PG = {k: (v.requires_grad_(True) if not k.startswith('pretrained_model') else v) for k, v in PG.items()}
    PD = {k: v.requires_grad_(True) for k, v in PD.items()}
    gopt = O.make_adam([v for v in PG.values() if v.requires_grad], 2e-4, 0.5, 0.999)
    dopt = O.make_adam(PD.values(), 2e-4, 0.5, 0.999)
    batch = {k: v.clone() for k, v in host_batch.items()}
    pool = O.ImagePool(80)
    return lambda: O.gan_train_step(PG, SG, PD, SD, PV, gopt, dopt, batch, pool=pool)

  probe, counts = {}, _thread_counts(all_threads)
  for threads in (counts if len(counts) > 1 else ()):     # (one candidate: nothing to probe)
    torch.set_num_threads(threads)
    step = fresh()
    step()                                       # warm-up
    t0 = time.time()
    step()
    probe[threads] = b / (time.time() - t0)
  best = max(probe, key=probe.get) if probe else counts[0]
  torch.set_num_threads(best)
  step = fresh()
  step()
  t0 = time.time()
  for _ in range(steps):
    step()
  value = b * steps / (time.time() - t0)
  return {'value': round(value, 4), 'unit': 'slices/s', 'cores': best, 'kind': 'port', 'cpu': _cpu_model(),
          'probe_by_threads': {str(k): round(v, 4) for k, v in probe.items()}, 'host_threads': os.cpu_count(),
          'timed_steps': steps,
          'sample': 'oracle (plain torch fp32) GAN step on the same %d slices of 256x256 as the GPU run: %d timed steps '
                    'after 1 warm-up at %d threads%s' % (b, steps, best, ' (the best of the probed counts)' if probe else '')}


def psnr_probe_c3(runner, host_batch, scale):
  """PSNR of the HIP generator vs the fp32 oracle on the same weights/batch with `scale` preset (train-mode
  BatchNorm); at the reference's initial scale = 0 the U-Net is multiplied by zero (SURVEY A-10)."""
  import torch
  import csmri_oracle as O
  with torch.no_grad():
    old = float(runner.gen.scale)
    runner.gen.scale.fill_(scale)
    PG, SG = _split_sd(runner.gen.state_dict())
    runner._set_train()
    dev = {k: v.cuda() for k, v in host_batch.items()}
    pred = runner.gen(dev['inp'], dev['kspace'], dev['mask'])['pred'].float().cpu()
    want = O.refinement_forward(PG, SG, host_batch['inp'], host_batch['kspace'], host_batch['mask'], True)
    runner.gen.scale.fill_(old)
  return O.psnr_batch(pred, host_batch['target']), O.psnr_batch(want['pred'], host_batch['target'])


def cpu_baseline_c2(runner, host_batch, sample_b=16, steps=5, all_threads=False, fast=False):
  sys.path.insert(0, os.path.join(ROOT, 'oracle'))
  import torch
  import csmri_oracle as O
  batch = {k: v[:sample_b].clone() for k, v in host_batch.items()}

  def fresh():
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in runner.model.state_dict().items()}
    opt = O.make_adam(P.values(), 2e-4, 0.9, 0.999)
    return P, (lambda: O.recnet_mse_step(P, opt, batch, 5))

  probe = {}
  counts = [min(32, os.cpu_count() or 1)] if fast else _thread_counts(all_threads)
  for threads in (counts if len(counts) > 1 else ()):
    torch.set_num_threads(threads)
    _, step = fresh()
    step()
    t0 = time.time()
    step()
    probe[threads] = sample_b / (time.time() - t0)
  best = max(probe, key=probe.get) if probe else counts[0]
  torch.set_num_threads(best)
  P, step = fresh()
  with torch.no_grad():
    psnr = O.psnr_batch(O.recnet_forward(P, batch['inp'], batch['kspace'], batch['mask'], 5), batch['target'])
  step()
  t0 = time.time()
  for _ in range(steps):
    step()
  value = sample_b * steps / (time.time() - t0)
  return {'value': round(value, 4), 'unit': 'slices/s', 'cores': best, 'kind': 'port', 'cpu': _cpu_model(),
          'probe_by_threads': {str(k): round(v, 4) for k, v in probe.items()}, 'host_threads': os.cpu_count(),
          'timed_steps': steps,
          'sample': 'oracle (plain torch fp32) RecNet(5,3,32) MSE step, first %d slices of the batch: %d timed steps '
                    'after 1 warm-up at %d threads%s' % (sample_b, steps, best, ' (the best of the probed counts)' if probe else '')}, psnr


def roofline(runner, loader_factory, dtype, steps=2, config='c3'):
  """Instrumented eager steps on ONE stream: HIP events (torch's current stream = the stream the library
  launches on) around every conv-library launch and every HBM-bound entry point."""
  import torch
  from csmri_hip import ops, lib
  if hasattr(runner, 'disable_graphs'):
    runner.disable_graphs()       # per-launch events need eager launches
    runner.overlap_streams = False  # concurrent side-stream kernels would inflate the brackets
    runner.prefetch_pretrained = False
  ops.PROFILE, lib.HBM_PROFILE = [], []
  runner.train_epoch(loader_factory(steps), 1)
  torch.cuda.synchronize()
  recs, ops.PROFILE = ops.PROFILE, None
  hrecs, lib.HBM_PROFILE = lib.HBM_PROFILE, None
  # cost of an empty event pair (reported, not subtracted)
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
  flop_kernels = {k: v for k, v in agg.items() if v[1] > 0}
  dom = max(flop_kernels, key=lambda k: flop_kernels[k][2])
  n, fl, sec = agg[dom]
  achieved = fl / sec / 1e12
  # --dtype fp8 only moves the eligible forward products to fp8: price the dominant kernel by what it multiplies
  peak = PEAK_TFLOPS['fp8' if 'fp8' in dom else ('bf16' if dtype == 'fp8' else dtype)]
  # HBM bytes per launch of that kernel: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this very
  # command (tools/pmc_bench.sh; FETCH_SIZE doubled per the gfx950 correction), committed under profiles/
  traffic, traffic_src = None, None
  tnames = {'c3': ('r06_pmc_bench_traffic.json', 'r05_pmc_bench_traffic.json', 'r04_pmc_bench_traffic.json'),
            'c2': ('r06_pmc_bench_traffic_c2.json', 'r05_pmc_bench_traffic_c2.json', 'r04_pmc_bench_traffic_c2.json')}.get(config, ())
  for tname in tnames:
    tpath = os.path.join(ROOT, 'profiles', tname)
    if dtype == 'bf16' and traffic is None and os.path.exists(tpath):   # the PMC passes ran this workload
      for name, rec in json.load(open(tpath)).items():
        if dom in name:
          traffic = round(rec['fetch_bytes_per_launch'] + rec['write_bytes_per_launch'])
          traffic_src = 'profiles/%s (rocprofv3 --pmc, %d launches)' % (tname, rec['launches'])
  rl = {'bound': 'mfma', 'kernel': dom, 'achieved': round(achieved, 2), 'peak': peak,
        'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4), 'traffic': traffic,
        'traffic_unit': 'B/launch (HBM, PMC)', 'traffic_source': traffic_src,
        'avg_launch_us': round(sec / n * 1e6, 2), 'event_pair_overhead_us': round(ovh * 1e6, 2),
        'launches': n // steps,
        'algorithmic_gflop_per_launch': round(fl / n / 1e9, 3)}
  conv_ms = sum(v[2] for v in agg.values()) / steps * 1e3
  hagg = {}
  for label, nbytes, e0, e1 in hrecs:
    a = hagg.setdefault(label, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += nbytes
    a[2] += e0.elapsed_time(e1) * 1e-3
  hbm = []
  for k, (cnt, nbytes, sec) in sorted(hagg.items(), key=lambda kv: -kv[1][2]):
    gbs = nbytes / sec / 1e9 if sec > 0 else 0.0
    hbm.append({'bound': 'hbm', 'kernel': k, 'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                'frac': round(gbs / PEAK_HBM_GBS, 4), 'launches': cnt // steps,
                'algorithmic_mb_per_launch': round(nbytes / cnt / 1e6, 3),
                'avg_launch_us': round(sec / cnt * 1e6, 2), 'ms_per_step': round(sec / steps * 1e3, 4)})
  return rl, table, conv_ms, hbm


def dist_forced():
  from training import distributed as dist_utils
  return dist_utils.exchange_active()


def run_leg(args, config, dtype, batch, steps, warmup, ws, rank, want_roofline=True, want_cpu=True, min_timed_s=0.0):
  """One measured leg: build the runner of `config`, W warm-up steps, settle, EXACTLY `steps` timed steps between
  barrier + synchronize pairs, then (outside the timed region) the roofline brackets and the CPU-oracle legs.
  Returns the JSON line's dict on rank 0 (None elsewhere)."""
  import torch
  from data.synthetic import synth_batch, synth_batch_radial
  runner, conf = build_runner(config, dtype, batch)
  size = 512 if config == 'c5' else SIZE
  if config == 'c5':
    host_batches = [synth_batch_radial(batch, size, size, spokes=C5_SPOKES, seed=conf.seed + 97 * rank + 100000 * i)
                    for i in range(N_HOST_BATCHES)]
  else:
    host_batches = [synth_batch(batch, size, size, acc=4, seed=conf.seed + 97 * rank + 100000 * i)
                    for i in range(N_HOST_BATCHES)]
  dev = torch.device('cuda', torch.cuda.current_device())
  resident = bool(args.device_resident)
  batch_bytes = sum(v.numel() * v.element_size() for v in host_batches[0].values())
  # one hipMemcpyAsync stream is served by one SDMA engine: the 134 MB C2 batch needs two to fit under the step
  # (8,000 -> 12,000 slices/s); the 17 MB C3 batch is done long before the step ends, and a second stream only
  # adds interference (same box, host-batch leg: 1 stream 1500-1515 slices/s, 2 streams 1408-1479, 3 streams 1445;
  # resident 1520)
  PinnedHostLoader.COPY_STREAMS = args.copy_streams if args.copy_streams > 0 else (2 if batch_bytes >= (64 << 20) else 1)

  def loader_factory(n, resident=resident):
    return PinnedHostLoader(host_batches, n, dev, resident=resident)

  def request(loader, volatile=False):      # batches come off the loader as this rank's shard, on the device
    try:
      return next(runner.data_iter)
    except StopIteration:
      runner.data_iter = None
      return None
  runner._request_data = request

  gan = config in ('c3', 'c5')
  no_graphs = args.no_graphs
  from csmri_hip import ops as _ops
  _ops.WGRAD_FINISH_MULTI = args.finish_multi
  if gan:
    runner.lookahead_first = not args.lookahead_last
    runner.overlap_streams = not args.no_overlap
    runner.prefetch_pretrained = not (args.no_prefetch or args.no_overlap)
  if not no_graphs:
    # capture the step once (eager warm-up steps inside); the timed region replays hipGraphs (with more than one rank
    # the gradient collectives stay eager between the captured segments)
    # (a capture failure is an ERROR: an eager run is ~2x slower and must not be recorded as this build's number;
    #  --no-graphs asks for the eager run explicitly)
    runner.enable_graphs({k: v.to(dev) for k, v in host_batches[0].items()})
  if warmup > 0:
    runner.train_epoch(loader_factory(warmup), 1, steps_per_train_summary=10 ** 9)
  torch.cuda.synchronize()
  # Settling (untimed, after the W warm-up steps): the first replays after a capture run below the steady rate
  # (clock ramp, allocator and pack caches), which a 20-step timed region reads as -7 %.  Replay until SETTLE_S of
  # wall time has passed, in chunks, so every rank runs the same number of steps.
  settle_steps = 0
  if args.settle_s > 0:
    t_s = time.perf_counter()
    est = None
    while True:
      n_chunk = 10 if est is None else max(10, min(200, int((args.settle_s - (time.perf_counter() - t_s)) / est)))
      if ws > 1:                                      # every rank runs the SAME number of steps (a step holds collectives)
        nc = torch.tensor([float(n_chunk)], device=dev)
        torch.distributed.all_reduce(nc, op=torch.distributed.ReduceOp.MAX)
        n_chunk = int(nc.item())
      t_c = time.perf_counter()
      runner.train_epoch(loader_factory(n_chunk), 1, steps_per_train_summary=10 ** 9)
      torch.cuda.synchronize()
      settle_steps += n_chunk
      est = max(1e-4, (time.perf_counter() - t_c) / n_chunk)
      # (at least two chunks: the first one holds the slow first replays and would spoil `est`)
      done = torch.tensor([1.0 if time.perf_counter() - t_s >= args.settle_s and settle_steps > 10 else 0.0], device=dev)
      if ws > 1:
        torch.distributed.all_reduce(done, op=torch.distributed.ReduceOp.MIN)     # all ranks leave together
      if float(done.item()) > 0:
        break
  if min_timed_s > 0 and settle_steps > 0:
    steps = max(steps, int(1.35 * min_timed_s / est) + 1)     # (est includes per-chunk sync overhead: margin)

  def timed(resident_leg):
    """EXACTLY `steps` steps between barrier + synchronize pairs; max over ranks."""
    if ws > 1:
      torch.distributed.barrier()
    torch.cuda.synchronize()
    timed_loader = loader_factory(steps, resident_leg)          # buffers and streams exist before the clock starts
    timed_loader.prime()                                        # ... and the copy pipeline is in its steady state
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = runner.train_epoch(timed_loader, 1, steps_per_train_summary=10 ** 9)
    torch.cuda.synchronize()
    if ws > 1:
      torch.distributed.barrier()
    torch.cuda.synchronize()
    dt_ = time.perf_counter() - t0
    if ws > 1:
      t = torch.tensor([dt_], dtype=torch.float64, device=dev)
      torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
      dt_ = float(t.item())
    return dt_, res

  from training.distributed import GradBucket
  GradBucket.TIMING = [] if ws > 1 or dist_forced() else None
  dt, (losses, metrics) = timed(resident)
  exposed_comm, grad_payload = None, None
  if GradBucket.TIMING is not None:
    # event pairs around every wait for a gradient exchange (GradBucket.wait): the time the step's stream STOOD there
    pairs, GradBucket.TIMING = GradBucket.TIMING, None
    exposed_comm = round(sum(a.elapsed_time(b) for a, b in pairs) / max(1, steps), 4)
    grad_payload = os.environ.get('CSMRI_GRAD_PAYLOAD', 'bf16')
  part = getattr(args, 'participation', None) or {}
  dt_resident = None
  if not resident and not args.no_input_ab:
    # the A/B leg: the same K steps with the batches already in HBM (its rate is reported, never `value`)
    dt_resident, _ = timed(True)

  # the instrumented roofline pass trains too (its steps all-reduce): every rank takes part
  prefetch_on = bool(getattr(runner, 'prefetch_pretrained', False))
  rl_out = None
  if want_roofline:
    rl_out = roofline(runner, loader_factory, dtype, config=config)
  if ws > 1:
    torch.distributed.barrier()
  if rank != 0:
    return None
  slices = ws * batch * steps
  value = slices / dt
  gf = C5_GFLOP_PER_SLICE if config == 'c5' else (GAN_GFLOP_PER_SLICE if gan else C2_GFLOP_PER_SLICE)
  if gan:
    workload = ('C3/C4 2-refinement GAN step: frozen RecNet(3,3,32)+3 DC, UNET, CNNDiscriminator, '
                'VGG19 loss, Adam x2; 256x256, 4x Cartesian, %d slices/GPU' % batch)
    metric = 'train slices/sec, 256x256 GAN refinement step'
    if config == 'c5':
      workload = ('C5 data format: the 2-refinement GAN step at 512x512, golden-angle radial undersampling '
                  '(%d spokes), %d slices/GPU, %s' % (C5_SPOKES, batch,
                  'fp8 forward products where eligible + bf16-storage FFT' if dtype == 'fp8' else
                  'bf16 convolutions + fp32 FFT'))
      metric = 'train slices/sec, 512x512 radial GAN refinement step'
    mode = 'eager' if no_graphs else ('hipGraph replay (one graph per step)' if ws == 1 else
                                      'hipGraph replay (4 segments, collectives eager between them)')
  else:
    workload = ('C2 RecNet(5 blocks,3 convs,32 filters)+5 DC MSE training step incl. DC adjoints, Adam; '
                '256x256, 4x Cartesian, %d slices/GPU' % batch)
    metric = 'train slices/sec, 256x256 RecNet (5-cascade DC-CNN) MSE step'
    mode = 'eager' if no_graphs else ('hipGraph replay (one graph per step)' if ws == 1 else
                                      'hipGraph replay (backward | Adam, gradient collectives eager between them)')
  line = {
      'metric': metric, 'value': round(value, 2), 'unit': 'slices/s',
      'n_gpus': ws, 'steps': steps, 'warmup': warmup, 'warmup_total_steps': warmup + settle_steps,
      'backend': part.get('backend'), 'collective_ranks': part.get('collective_ranks'),
      'rccl_ranks': part.get('rccl_ranks'), 'distinct_gpus': part.get('distinct_gpus'),
      'grad_payload': grad_payload, 'exposed_comm_ms_per_step': exposed_comm,
      'ms_per_step': round(dt / steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
      'vs_baseline': None, 'dtype': dtype, 'data': 'synthetic',
      'input': 'batches resident in HBM when the timed region starts (%d distinct batches cycled)' % N_HOST_BATCHES
               if resident else
               'pinned host batches, H2D on copy streams inside the timed region: the copy of batch t+1 is issued during '
               'step t, K copies in a window of K steps (pipeline primed with batch 0 before the clock, as in a running '
               'epoch; %d distinct batches cycled)' % N_HOST_BATCHES,
      'input_ab': None if dt_resident is None else
                  {'host': round(slices / dt, 2), 'resident': round(slices / dt_resident, 2), 'unit': 'slices/s',
                   'note': 'value = host (H2D of every batch inside the timed region); resident = the same K steps '
                           'timed again with the batches already in HBM'},
      'host_numa': dict(NUMA_INFO) or None, 'copy_streams': None if resident else PinnedHostLoader.COPY_STREAMS,
      'timed_region_s': round(dt, 3), 'settle_steps': settle_steps,
      'prefetch': 'frozen RecNet forward of batch t+1 on a side stream during step t' if prefetch_on else None,
      'launch_mode': mode,
      'config': {'workload': workload, 'per_gpu_batch': batch, 'global_batch': ws * batch,
                 'parallelism': 'dp%d' % ws, 'image': [size, size]},
      'algorithmic_tflops': round(value * gf / 1e3, 2),
      'final_losses': {k: round(v.value, 5) for k, v in losses.items()},
  }
  for k in ('gen_psnr', 'psnr'):
    if k in metrics:
      line[k] = round(metrics[k].value, 4)
  if rl_out is not None:
    rl, table, conv_ms, hbm = rl_out
    line['roofline'] = rl
    line['roofline_hbm'] = hbm
    line['conv_kernels'] = table
    line['conv_ms_per_step'] = round(conv_ms, 3)
  if ws == 1 and want_cpu:
    # fresh runner with the same seed = same initial weights as the HIP run started from
    ref_runner, _ = build_runner(config, dtype, batch)
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    if gan:
      if not args.other_cpu_fast:        # (side legs of the default line: PSNR probes only)
        line['cpu_baseline'] = cpu_baseline_c3(ref_runner, host_batches[0], all_threads=args.cpu_all_threads)
      # the 0.01 dB criterion where the U-Net contributes (scale preset; the reference starts at scale = 0
      # where pred == pretrained, reported beside it)
      probes = (('', 0.02), ('_scale0p25', 0.25), ('_scale0', 0.0)) if config == 'c3' else (('', 0.02),)
      for tag, sc in probes:
        ph, pc = psnr_probe_c3(ref_runner, host_batches[0], sc)
        line['psnr_hip_db' + tag], line['psnr_cpu_db' + tag] = round(ph, 5), round(pc, 5)
        line['psnr_delta_db' + tag] = round(abs(ph - pc), 5)
      line['psnr_probe'] = ('generator forward (train-mode BatchNorm) on batch 0, initial weights, RefinementWrapper.scale '
                            'preset to 0.02 (headline psnr_delta_db), 0.25 and 0 (the reference\'s initial value)')
    else:
      fast = args.other_cpu_fast and config != args.config
      base, psnr_cpu = cpu_baseline_c2(ref_runner, host_batches[0], all_threads=args.cpu_all_threads,
                                       steps=3 if fast else 5, fast=fast)
      line['cpu_baseline'] = base
      with torch.no_grad():
        ref_runner.model.train()
        d0 = {k: v[:16].cuda() for k, v in host_batches[0].items()}
        import csmri_oracle as O
        ph = O.psnr_batch(ref_runner.model(d0['inp'], d0['kspace'], d0['mask']).float().cpu(), host_batches[0]['target'][:16])
      line['psnr_hip_db'], line['psnr_cpu_db'] = round(ph, 5), round(psnr_cpu, 5)
      line['psnr_delta_db'] = round(abs(ph - psnr_cpu), 5)
    if 'cpu_baseline' in line:
      line['gpu_over_cpu'] = round(value / line['cpu_baseline']['value'], 1)
    del ref_runner
  del runner
  torch.cuda.synchronize()
  PinnedHostLoader.forget(host_batches)
  import gc
  gc.collect()
  torch.cuda.empty_cache()
  return line


def spawn_ranks(args, argv):
  """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves (one per GPU, RCCL), as a
  CHILD process and BEFORE this process has imported torch or touched the GPU, and exit with its code.  A bare
  `--gpus 8` must never quietly measure one GPU."""
  import subprocess
  port = os.environ.get('MASTER_PORT') or str(29500 + os.getpid() % 2000)
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
         '--master-addr', '127.0.0.1', '--master-port', port, os.path.abspath(__file__)] + list(argv)
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL needs it on this host driver
  return subprocess.call(cmd, env=env)


def participation(ws, rank, dev):
  """Who really took part: every rank adds a one-hot row over the job's process group (the backend that also carries
  the gradient exchange) -- `collective_ranks` counts the ranks whose contribution arrived; `distinct_gpus` counts the
  different (host, PCI bus id) pairs behind them (2 ranks on one GPU over gloo, the functional-test mode, says 1)."""
  import socket
  import torch
  import torch.distributed as dist
  props = torch.cuda.get_device_properties(dev)
  me = '%s/%04x:%02x:%02x' % (socket.gethostname(), props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
  if ws == 1:
    return {'backend': None, 'collective_ranks': 1, 'rccl_ranks': 0, 'distinct_gpus': 1, 'gpus': [me]}
  hot = torch.zeros(ws, device=dev)
  hot[rank] = 1.0
  dist.all_reduce(hot, op=dist.ReduceOp.SUM)
  names = [None] * ws
  dist.all_gather_object(names, me)
  backend = dist.get_backend()
  n = int((hot > 0.5).sum().item())
  assert float(hot.max().item()) == 1.0, 'a rank id was used twice: %r' % hot.tolist()
  return {'backend': backend, 'collective_ranks': n, 'rccl_ranks': n if backend == 'nccl' else 0,
          'distinct_gpus': len(set(names)), 'gpus': names}


def main():
  args = parse()
  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    sys.exit(spawn_ranks(args, sys.argv[1:]))
  import torch
  from training import distributed as dist_utils
  ws = dist_utils.init_from_env()
  rank = dist_utils.rank()
  local = int(os.environ.get('LOCAL_RANK', '0'))
  torch.cuda.set_device(local if torch.cuda.device_count() > local else 0)
  if ws != max(1, args.gpus):
    raise SystemExit('bench.py: --gpus %d but the process group has %d rank(s) (WORLD_SIZE=%s): refusing to report a '
                     'number for a job that is not the one asked for' % (args.gpus, ws, os.environ.get('WORLD_SIZE')))
  args.participation = participation(ws, rank, torch.device('cuda', torch.cuda.current_device()))
  if ws > 1 and args.participation['collective_ranks'] != ws:
    raise SystemExit('bench.py: %d of %d ranks answered the collective' % (args.participation['collective_ranks'], ws))
  line = run_leg(args, args.config, args.dtype, args.batch, args.steps, args.warmup, ws, rank,
                 want_roofline=not args.no_roofline, want_cpu=not args.no_cpu_baseline)
  if rank != 0:
    return
  if ws == 1 and args.config == 'c3' and args.dtype == 'bf16' and not args.no_other_configs:
    # BASELINE configs 2 and 5 in front of the driver: short legs of their own (warm-up, settling, >= 0.5 s timed),
    # attached to the ONE JSON line.  Their CPU-oracle legs are cut to the PSNR probe (+ one timed step for C2).
    args.other_cpu_fast = True
    others = []
    # (third leg: config 5's fp8 variant -- the frozen VGG stack on e4m3fn operands + bf16-storage DC -- timed right behind
    #  its bf16 leg on the same box; no roofline / CPU passes of its own)
    # ('c3', 'fp32'): the headline workload in the REFERENCE's arithmetic (fp32 operands on the fp32-matrix MFMA path,
    #  same kernels' exact mode), driver-timed on the same box; its roofline is priced against the 157.3 TFLOP/s fp32 peak
    for cfg, dt in (('c3', 'fp32'), ('c2', 'bf16'), ('c5', 'bf16'), ('c5', 'fp8')):
      try:
        f8 = dt == 'fp8'
        if f8 and not args.inprocess_legs:
          # the fp8 variant runs as a CHILD process (same interpreter, same file): the newest code path of the tree
          # cannot take the headline down with it, whatever it does
          import subprocess
          ref = [x for x in others if x.get('dtype') == 'bf16' and 'radial' in x.get('metric', '')]
          r = subprocess.run([sys.executable, os.path.abspath(__file__), '--config', cfg, '--dtype', dt,
                              '--steps', str(ref[0]['steps'] if ref else 80),     # the bf16 leg's K, W and settling
                              '--warmup', '5', '--settle-s', str(args.settle_s),
                              '--no-other-configs', '--no-cpu-baseline', '--no-roofline'],
                             capture_output=True, text=True, timeout=600)
          lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
          if r.returncode != 0 or not lines:
            raise RuntimeError('fp8 leg: child exited with %d: %s' % (r.returncode, r.stderr[-300:]))
          o = json.loads(lines[-1])
        else:
          o = run_leg(args, cfg, dt, DEFAULT_BATCH[cfg], 20, 5, 1, 0, want_roofline=not args.no_roofline,
                      want_cpu=not args.no_cpu_baseline, min_timed_s=0.5)
        keep = ('metric', 'value', 'unit', 'steps', 'warmup', 'settle_steps', 'ms_per_step', 'dtype', 'config',
                'algorithmic_tflops', 'launch_mode', 'input', 'input_ab', 'roofline', 'roofline_hbm', 'psnr_delta_db',
                'psnr_hip_db', 'psnr_cpu_db', 'cpu_baseline', 'final_losses')
        others.append({k: o[k] for k in keep if k in o})
        if f8:
          ref = [x for x in others if x.get('dtype') == 'bf16' and x.get('metric') == o.get('metric')]
          if ref:
            others[-1]['vs_bf16_leg_same_run'] = round(o['value'] / ref[0]['value'], 4)
            others[-1]['vs_bf16_note'] = ('same box, same K / W / settling; the fp8 leg is a child process with its own '
                                          'graph capture: indicative, +-1 %')
      except Exception as e:                      # the headline must survive a failing side leg
        others.append({'config': {'workload': cfg, 'dtype': dt}, 'error': repr(e)})
    line['other_configs'] = others
  print(json.dumps(line))


if __name__ == '__main__':
  main()
