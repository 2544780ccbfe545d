#!/usr/bin/env python3
"""(test infrastructure: lives under tests/ because it imports the oracle)
How much gradient fidelity does a 16-bit activation format cost on THIS training step?

Runs the CPU oracle's GAN step (oracle/csmri_oracle.py, fp32) three times on the same weights,
batch and injected dropout masks: plain fp32, and with every tensor the HIP path stores in the
compute dtype rounded to bf16 resp. fp16 at the same points (conv / BatchNorm+activation outputs,
max-pool outputs, weights; gradients of the same tensors on the way back; fp32 accumulation, fp32
BatchNorm statistics, fp32 losses, fp32 weight gradients).  Prints per-tensor gradient cosine /
relative L2 against the fp32 run -- the error floor ANY implementation with that storage format has.

  python tests/lowprec_sensitivity.py [--size 128] [--batch 4] [--small]

Test infrastructure (uses the oracle): never imported by the product."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import csmri_oracle as O  # noqa: E402

import csmri_lowprec as LP  # noqa: E402


def run(mode, PG0, SG0, PD0, SD0, PV, batch, masks, gscale):
  PG = {k: (v.clone().requires_grad_(True) if not k.startswith('pretrained_model') else v.clone()) for k, v in PG0.items()}
  PD = {k: v.clone().requires_grad_(True) for k, v in PD0.items()}
  SG = {k: v.clone() for k, v in SG0.items()}
  SD = {k: v.clone() for k, v in SD0.items()}
  gopt = O.make_adam([v for v in PG.values() if v.requires_grad], 2e-4, 0.5, 0.999)
  dopt = O.make_adam(PD.values(), 2e-4, 0.5, 0.999)
  grads = {}
  for opt, P, tag in ((gopt, PG, 'G'), (dopt, PD, 'D')):
    orig = opt.step

    def step(orig=orig, P=P, tag=tag):
      grads[tag] = {k: v.grad.detach().clone() for k, v in P.items() if v.requires_grad and v.grad is not None}
      orig()
    opt.step = step
  dm = [masks[0:3], masks[3:6], masks[6:9]]
  with LP.emulate(mode, gscale):
    losses, metrics, _ = O.gan_train_step(PG, SG, PD, SD, PV, gopt, dopt, batch, pool=O.ImagePool(80), dropout_masks=dm)
  return losses, metrics, grads


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--size', type=int, default=128)
  ap.add_argument('--batch', type=int, default=4)
  ap.add_argument('--scale', type=float, default=0.25)
  ap.add_argument('--gscale', type=float, default=1024.0, help='static loss scale used for the fp16 run')
  a = ap.parse_args()
  torch.manual_seed(1)
  g = torch.Generator().manual_seed(1)
  PG = O.init_recnet(3, 3, 32, gen=g, prefix='pretrained_model.conv_blocks')
  Pu, Su = O.init_unet(gen=g, prefix='learnable_model.')
  PG.update(Pu)
  PG['scale'] = torch.full((1,), a.scale)
  PD, SD = O.init_disc(gen=g)
  PV = O.init_vgg(gen=torch.Generator().manual_seed(19))
  batch = O.synth_batch(a.batch, a.size, a.size, acc=4, seed=123)
  chans = [512, 1024, 1024]
  masks = [(torch.rand(a.batch, c, 1, 1, generator=g) < 0.5).float() * 2.0 for _ in range(3) for c in chans]
  ref = run(None, PG, Su, PD, SD, PV, batch, masks, 1.0)
  for mode, gs in (('bf16', 1.0), ('fp16', a.gscale)):
    got = run(mode, PG, Su, PD, SD, PV, batch, masks, gs)
    print('==== %s (gradient scale %g)' % (mode, gs))
    for k in sorted(ref[0]):
      print('  loss %-26s %.6e vs %.6e  rel %.2e' % (k, got[0][k], ref[0][k], abs(got[0][k] - ref[0][k]) / abs(ref[0][k])))
    print('  gen_psnr %.5f vs %.5f' % (got[1]['gen_psnr'], ref[1]['gen_psnr']))
    for tag in ('G', 'D'):
      worst = (1.0, 0.0, '')
      for k, gr in ref[2][tag].items():
        gh = got[2][tag][k]
        cos = float((gh * gr).sum() / (gh.norm() * gr.norm() + 1e-30))
        err = float((gh - gr).norm() / (gr.norm() + 1e-30))
        if gr.numel() > 1 and cos < worst[0]:
          worst = (cos, err, k)
        print('  grad %s %-62s cos %.5f rel_l2 %.3e' % (tag, k, cos, err))
      print('  worst %s: cos %.5f rel_l2 %.3e %s' % ((tag,) + worst))


if __name__ == '__main__':
  main()
