"""Training trajectories, not single steps: the bf16 compute path trained next to
the fp32 path (itself pinned to the CPU oracle by tests/test_hip_path.py) from the same weights, over the same
batches, with the same dropout masks and pool decisions -- tools/trajectory.py.  Full-size runs (256^2, >= 500
steps) are committed under profiles/r03_trajectory_*.json; these are the short forms that run with the suite."""
import os
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def test_recnet_mse_training_bf16_tracks_fp32_128sq_60_steps():
  """RecNet(5,3,32) MSE, 128^2, batch 8, 60 Adam steps over 8 distinct batches: the bf16 run's held-out PSNR stays
  within 0.05 dB of the fp32 run's and its smoothed loss curve within 3 % (measured: see the printed summary)."""
  import trajectory
  out = trajectory.main(['--config', 'c2', '--steps', '60', '--size', '128', '--batch', '8', '--distinct', '8',
                         '--heldout', '2'])
  s = out['summary']['bf16']
  f32, b16 = out['runs']['fp32'], out['runs']['bf16']
  # training happened: the loss fell substantially in both runs (60 steps at lr 2e-4)
  assert f32['curves']['loss_MSE'][-1] < 0.7 * f32['curves']['loss_MSE'][0]
  assert b16['curves']['loss_MSE'][-1] < 0.7 * b16['curves']['loss_MSE'][0]
  assert s['final_delta_psnr_heldout_train_bn_db'] < 0.05, s
  assert s['max_rel_delta_smoothed_loss_MSE'] < 0.03, s


def test_gan_refinement_training_bf16_tracks_fp32_reduced_width_40_steps():
  """The GAN refinement step at reduced widths (every structure of the step: frozen pretrained RecNet through the
  checkpoint hand-off, U-Net, three discriminator passes with Philox dropout and the image pool, VGG loss), 128^2,
  batch 4, 40 steps: same dropout masks and pool decisions in both runs; the generator's PSNR curve of the bf16 run
  stays within 0.1 dB of the fp32 run's, held-out PSNR within 0.1 dB."""
  import trajectory
  out = trajectory.main(['--config', 'c3', '--steps', '40', '--size', '128', '--batch', '4', '--distinct', '8',
                         '--heldout', '2', '--width', 'reduced', '--pretrain-steps', '40'])
  s = out['summary']['bf16']
  assert s['final_delta_psnr_heldout_train_bn_db'] < 0.1, s
  assert s['max_delta_smoothed_gen_psnr_db'] < 0.1, s
  for k, v in out['runs']['bf16']['curves'].items():
    assert all(x == x and abs(x) < 1e4 for x in v), k
