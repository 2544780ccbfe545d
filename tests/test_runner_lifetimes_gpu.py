"""Runners built, captured into hipGraphs, replayed and dropped over and over in ONE process.

Round 5's default `bench.py` run (four runners with captured graphs one after the other in one process) died once
with SIGSEGV at the first graph replay of its last leg; `tools/segv_hunt.sh` reproduces that form (DESIGN.md section 4
has the backtrace and what it was narrowed to).  Whatever a runner leaves behind -- process-wide role streams, the
weight-gradient queues of csmri_hip.ops, packed-weight tables, graph memory pools, the caching allocator's blocks --
the next runner's capture and replays must not trip over it, in any order of configurations and compute dtypes."""
import gc
import math
import os
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench():
  if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
  import bench
  return bench


def _one_runner(bench, config, dtype, batch, steps=3):
  from data.synthetic import synth_batch, synth_batch_radial
  runner, conf = bench.build_runner(config, dtype, batch)
  size = 512 if config == 'c5' else 256
  if config == 'c5':
    host = [synth_batch_radial(batch, size, size, spokes=bench.C5_SPOKES, seed=7 + i) for i in range(2)]
  else:
    host = [synth_batch(batch, size, size, acc=4, seed=7 + i) for i in range(2)]
  dev = [{k: v.cuda() for k, v in b.items()} for b in host]

  class Loader(list):
    batch_size = batch
  if config in ('c3', 'c5'):
    runner.overlap_streams = True
    runner.prefetch_pretrained = True
  runner.enable_graphs(dev[0])
  losses, _ = runner.train_epoch(Loader([dev[i % 2] for i in range(steps)]), 1, steps_per_train_summary=10 ** 9)
  torch.cuda.synchronize()
  vals = {k: v.value for k, v in losses.items()}
  assert vals and all(math.isfinite(v) for v in vals.values()), (config, dtype, vals)
  runner.disable_graphs()
  del runner
  gc.collect()
  torch.cuda.empty_cache()
  return vals


def test_build_capture_replay_drop_all_configs_ten_times():
  bench = _bench()
  order = [('c3', 'bf16', 2), ('c2', 'bf16', 4), ('c5', 'bf16', 1), ('c5', 'fp8', 1)]
  first = {}
  for rep in range(10):
    for config, dtype, batch in (order if rep % 2 == 0 else order[::-1]):
      vals = _one_runner(bench, config, dtype, batch)
      key = (config, dtype)
      if key not in first:
        first[key] = vals
      elif config == 'c2':
        # same seeds, same batches, no randomness in the RecNet step: every repetition reproduces the first bit for bit
        assert vals == first[key], (rep, key, vals, first[key])
  assert len(first) == 4
