"""bench.py prints ONE JSON line with the contract's keys (short run, no CPU baseline leg)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_json_line_schema():
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '2',
                        '--no-cpu-baseline'], capture_output=True, text=True, timeout=900, cwd=ROOT)
  assert out.returncode == 0, out.stderr[-2000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, out.stdout[-2000:]
  d = json.loads(lines[0])
  for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
    assert k in d, k
  assert d['unit'] == 'slices/s' and d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 2
  assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
  assert d['dtype'] == 'bf16' and d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
  assert abs(d['value'] - 8 * 3 / (d['ms_per_step'] * 3e-3)) < 0.02 * d['value']
  rl = d['roofline']
  for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
    assert k in rl, k
  assert rl['bound'] == 'mfma' and rl['unit'] == 'TFLOP/s' and rl['peak'] == 2500.0
  assert abs(rl['frac'] - rl['achieved'] / rl['peak']) < 1e-3 and 0.0 < rl['frac'] < 1.0
