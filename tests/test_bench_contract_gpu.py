"""bench.py prints ONE JSON line with the contract's keys (short runs, no CPU-timing legs)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
        'vs_baseline', 'dtype', 'data', 'config', 'roofline')


def _one_line(out):
  assert out.returncode == 0, out.stderr[-3000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, out.stdout[-2000:]
  return json.loads(lines[0])


def _check_headline(d, steps, warmup):
  for k in KEYS:
    assert k in d, k
  assert d['unit'] == 'slices/s' and d['n_gpus'] == 1 and d['steps'] == steps and d['warmup'] == warmup
  assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
  assert d['dtype'] == 'bf16' and d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
  assert abs(d['value'] - 8 * steps / (d['ms_per_step'] * steps * 1e-3)) < 0.02 * d['value']
  # the metric includes the H2D of every batch (reference training/base_runner.py:29-41); the resident
  # A/B leg is reported beside it, never as `value`
  assert 'H2D on copy streams inside the timed region' in d['input']
  ab = d['input_ab']
  assert abs(ab['host'] - d['value']) < 1e-6 * d['value'] + 0.02 and ab['resident'] > 0
  assert d['warmup_total_steps'] == d['warmup'] + d['settle_steps']
  rl = d['roofline']
  for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
    assert k in rl, k
  assert rl['bound'] == 'mfma' and rl['unit'] == 'TFLOP/s' and rl['peak'] == 2500.0
  assert abs(rl['frac'] - rl['achieved'] / rl['peak']) < 1e-3 and 0.0 < rl['frac'] < 1.0


def test_bench_json_line_schema_with_other_configs():
  """The default invocation: the C3 headline plus short C2, C5 and C5-with-compute_dtype-fp8 legs (`other_configs`)."""
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '2',
                        '--no-cpu-baseline', '--settle-s', '0.2'], capture_output=True, text=True, timeout=1200,
                       cwd=ROOT)
  d = _one_line(out)
  _check_headline(d, 3, 2)
  others = d['other_configs']
  assert len(others) == 4 and not any('error' in o for o in others[:3]), others
  c3f32, c2, c5, c5f8 = others
  # the headline workload in the reference's arithmetic (fp32), driver-timed beside the bf16 line; its roofline is
  # priced against the fp32-matrix peak
  assert c3f32['dtype'] == 'fp32' and 'C3' in c3f32['config']['workload'] and c3f32['value'] > 0
  assert c3f32['roofline']['peak'] == 157.3 and 0.0 < c3f32['roofline']['frac'] < 1.0
  # (--no-cpu-baseline skips the oracle legs, the PSNR probe among them; the default invocation carries psnr_delta_db)
  assert c3f32['value'] < d['value'] and c3f32.get('psnr_delta_db', 0.0) <= 0.01, c3f32
  assert 'C2' in c2['config']['workload'] and 'C5' in c5['config']['workload']
  # BASELINE config 5's fp8 variant right behind its bf16 leg (a child process of bench.py): frozen VGG stack on e4m3fn
  # operands + bf16-storage FFT
  assert 'error' not in c5f8, c5f8
  assert 'C5' in c5f8['config']['workload']
  assert c5f8['dtype'] == 'fp8' and c5f8['value'] > 0 and 0.8 < c5f8['vs_bf16_leg_same_run'] < 1.3, c5f8
  others = [c2, c5]
  # the C2 leg's roofline describes the kernel the timed step runs (the fused conv-block backward), not the per-layer
  # backward it replaces
  assert c2['roofline']['kernel'] == 'convblock_bwd_kernel', c2['roofline']
  for o in others:
    for k in ('config', 'value', 'ms_per_step', 'dtype', 'roofline', 'roofline_hbm', 'input', 'input_ab'):
      assert k in o, k
    assert 'H2D on copy streams inside the timed region' in o['input']
    assert o['dtype'] == 'bf16' and o['value'] > 0 and o['steps'] * o['ms_per_step'] >= 400.0    # ~0.5 s timed (the step count comes from the settle phase's rate)
    b = o['config']['per_gpu_batch']
    assert abs(o['value'] - b / (o['ms_per_step'] * 1e-3)) < 0.02 * o['value']


def test_bench_under_torch_distributed_run_single_rank():
  """How the driver launches N > 1, with one process: the same line comes out."""
  port = 29400 + os.getpid() % 100
  out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
                        '--master-addr', '127.0.0.1', '--master-port', str(port),
                        os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '2',
                        '--no-cpu-baseline', '--no-other-configs', '--settle-s', '0.2'],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
  d = _one_line(out)
  _check_headline(d, 3, 2)
  assert d['config']['parallelism'] == 'dp1' and 'one graph per step' in d['launch_mode']
  assert d['collective_ranks'] == 1 and d['distinct_gpus'] == 1 and d['exposed_comm_ms_per_step'] is None


def _two_rank_bench(extra):
  """bench.py as the driver launches it for N = 2 -- torch.distributed.run, two processes -- on the ONE GPU of the test
  box: CSMRI_DIST_BACKEND=gloo carries the collectives (both ranks share device 0), everything else is the N > 1 code
  path: per-rank shards, the barrier + max-over-ranks timing, the settle loop's all-reduced step count, rank 0's line."""
  port = 29500 + os.getpid() % 100
  env = dict(os.environ, CSMRI_DIST_BACKEND='gloo')
  return subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                         '--master-addr', '127.0.0.1', '--master-port', str(port),
                         os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2',
                         '--no-cpu-baseline', '--no-other-configs', '--no-roofline', '--settle-s', '0.2'] + extra,
                        capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)


def test_bench_under_torch_distributed_run_two_ranks():
  """the first multi-GPU driver run must not be the first time `bench.py --gpus 2` executes."""
  d = _one_line(_two_rank_bench([]))
  assert d['n_gpus'] == 2 and d['steps'] == 3 and d['scaling'] == 'weak' and d['unit'] == 'slices/s'
  assert d['config']['parallelism'] == 'dp2' and d['config']['per_gpu_batch'] == 8 and d['config']['global_batch'] == 16
  # with more than one rank the step is four captured segments with the gradient collectives between them
  assert '4' in d['launch_mode'] or 'four' in d['launch_mode'], d['launch_mode']
  # whole-job aggregate: both ranks' slices over the max-over-ranks time
  assert abs(d['value'] - 16 / (d['ms_per_step'] * 1e-3)) < 0.02 * d['value']
  assert d['warmup_total_steps'] == d['warmup'] + d['settle_steps']
  _check_two_rank_fields(d)


def _check_two_rank_fields(d):
  # who took part, proven by a collective over the job's process group: two ranks answered; the functional-test mode
  # (both ranks on the box's one GPU, gloo) is visible as such -- backend gloo, 0 RCCL ranks, ONE distinct GPU
  assert d['n_gpus'] == d['collective_ranks'] == 2, d
  assert d['backend'] == 'gloo' and d['rccl_ranks'] == 0 and d['distinct_gpus'] == 1
  assert d['grad_payload'] in ('bf16', 'fp32') and d['exposed_comm_ms_per_step'] >= 0.0


def test_bare_gpus_2_spawns_two_ranks():
  """`python bench.py --gpus 2` WITHOUT torch.distributed.run starts its own two ranks (before it imports torch) and the
  line says so; it can no longer run one rank and report n_gpus = 1 under a --gpus 2 command line."""
  env = dict(os.environ, CSMRI_DIST_BACKEND='gloo', MASTER_PORT=str(29700 + os.getpid() % 100))
  for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
    env.pop(k, None)
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2',
                        '--no-cpu-baseline', '--no-other-configs', '--no-roofline', '--settle-s', '0.2'],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
  d = _one_line(out)
  _check_two_rank_fields(d)
  assert d['config']['parallelism'] == 'dp2' and d['config']['global_batch'] == 16


def test_world_size_mismatch_is_refused():
  """--gpus 2 inside a ONE-rank torch.distributed.run job: no line, non-zero exit (round 5 accepted it and reported one GPU)."""
  port = 29800 + os.getpid() % 100
  out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
                        '--master-addr', '127.0.0.1', '--master-port', str(port),
                        os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1'],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
  assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith('{')]
  assert 'refusing' in out.stderr


def test_bench_c2_under_torch_distributed_run_two_ranks():
  d = _one_line(_two_rank_bench(['--config', 'c2']))
  assert d['n_gpus'] == 2 and d['config']['parallelism'] == 'dp2' and 'C2' in d['config']['workload']
  assert d['config']['global_batch'] == 2 * d['config']['per_gpu_batch']
  assert abs(d['value'] - d['config']['global_batch'] / (d['ms_per_step'] * 1e-3)) < 0.02 * d['value']
