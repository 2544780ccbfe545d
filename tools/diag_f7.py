"""Diagnostic: per-tensor deviation of the HIP GAN step from the reference golden F7."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'csmri-refinement_amd'), os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
import csmri_oracle as O
from test_hip_path import gan_conf, load, sub, T, Loader
from utils.config import Configuration
from models.utils import set_default_compute_dtype
from training import build_runner
from csmri_hip import ops
f = load('F7_gan_step')
dtype = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
set_default_compute_dtype(dtype)
runner = build_runner(gan_conf(Configuration, dtype), 'adversarial', '0', 'train')
runner.gen.load_state_dict(sub(f, 'G0.')); runner.disc.load_state_dict(sub(f, 'D0.')); ops.bump_weight_epoch()
for step in range(2):
  batch = O.synth_batch(2, 128, 128, acc=4, seed=40 + step)
  runner.disc.injected_dropout = [T(f['step%d.mask%d' % (step, j)]) for j in range(9)]
  g_before = {k: v.clone() for k, v in runner.gen.state_dict().items()}
  d_before = {k: v.clone() for k, v in runner.disc.state_dict().items()}
  runner.train_epoch(Loader([batch]), 1)
  for tag, sd, before in (('G', runner.gen.state_dict(), g_before), ('D', runner.disc.state_dict(), d_before)):
    for k, v in sub(f, '%s%d.' % (tag, step + 1)).items():
      if 'num_batches' in k or k.startswith('pretrained'): continue
      cur = sd[k].cpu().float(); v = v.float()
      prev = sub(f, '%s%d.' % (tag, step)).get(k, None)
      d = (cur - v).abs()
      upd = (v - prev).abs().max().item() if prev is not None else float('nan')
      frac_bad = float((d > 2e-5).float().mean())
      print('step%d %s %-55s maxdev %.2e  ref_update %.2e  frac>2e-5 %.4f  n %d' % (step, tag, k, d.max().item(), upd, frac_bad, v.numel()))
