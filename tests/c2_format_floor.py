#!/usr/bin/env python3
"""Which stored tensor of the bf16 RecNet training step costs what -- on the CPU ORACLE (test infrastructure: this
script imports oracle/, it is not part of the product and not collected by pytest).

The HIP bf16 path of BASELINE config C2 (RecNet 5/3/32 MSE training; reference models/recnet.py:139-161,
training/runner.py:154-178) keeps these tensors in bf16 and everything else (accumulators, the 2-channel block
output, DC, losses, weight gradients, Adam) in fp32:

  IN    the block input x            (the DC layer's channel-padded bf16 copy)
  W     the packed weight copies     (forward and data-gradient packs; fp32 masters)
  ACT   the two 32-channel activations a1, a2 (rounded once, after LeakyReLU)
  GOUT  the gradient of the block output as an MFMA operand
  GACT  the gradients dA2, dA1 (after the LeakyReLU derivative)
  GIN   the gradient dX of the block input (bf16, channel-padded)

This script trains the oracle's RecNet from the same initial weights over the same batches with any subset of those
roundings switched on (round-to-nearest-even to bf16, fp32 arithmetic otherwise) and reports the final held-out PSNR
of each variant next to the plain fp32 run and fp32 controls from 1e-6-perturbed weights -- i.e. the FORMAT's own
effect on the training trajectory, separated by class, with no kernel of the product involved.

  python tests/c2_format_floor.py --size 128 --batch 8 --steps 2000 --out profiles/r04_c2_format_floor_cpu.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))

import torch
import torch.nn.functional as F

import csmri_oracle as O

CLASSES = ('IN', 'W', 'ACT', 'GOUT', 'GACT', 'GIN')


def q(t):
  return t.to(torch.bfloat16).to(torch.float32)


class R(torch.autograd.Function):
  """identity with optional bf16 rounding of the value (forward) and of the gradient (backward)."""

  @staticmethod
  def forward(ctx, x, fwd, bwd):
    ctx.bwd = bwd
    return q(x) if fwd else x

  @staticmethod
  def backward(ctx, g):
    return (q(g) if ctx.bwd else g), None, None


class QW(torch.autograd.Function):
  """the bf16 compute copy of an fp32 master weight: rounded value forward, straight-through gradient."""

  @staticmethod
  def forward(ctx, w):
    return q(w)

  @staticmethod
  def backward(ctx, g):
    return g


def forward(P, inp, kspace, mask, on, nb=5, slope=0.01):
  x = inp
  for b in range(nb):
    x = R.apply(x, 'IN' in on, 'GIN' in on)
    for i in range(3):
      w = P['conv_blocks.%d.layers.%d.weight' % (b, 3 * i + 1)]
      bias = P['conv_blocks.%d.layers.%d.bias' % (b, 3 * i + 1)]
      if 'W' in on:
        w = QW.apply(w)
      x = F.conv2d(F.pad(x, (1, 1, 1, 1)), w, bias)
      if i < 2:
        x = R.apply(x, False, 'GACT' in on)
        x = R.apply(F.leaky_relu(x, slope), 'ACT' in on, False)
      else:
        x = R.apply(x, False, 'GOUT' in on)
    x = O.dc_layer(x, kspace, mask)
  return x


def run(on, args, train, held, pseed):
  torch.manual_seed(args.seed)
  gen = torch.Generator().manual_seed(args.seed)
  P = O.init_recnet(5, 3, 32, gen)
  if pseed:
    g2 = torch.Generator().manual_seed(args.seed + 77 + 1009 * pseed)
    P = {k: v * (1.0 + 1e-6 * torch.randn(v.shape, generator=g2)) for k, v in P.items()}
  P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
  opt = O.make_adam(P.values(), 2e-4, 0.9, 0.999)
  curve = []
  t0 = time.time()
  for s in range(args.steps):
    b = train[s % len(train)]
    opt.zero_grad()
    pred = forward(P, b['inp'], b['kspace'], b['mask'], on)
    loss = F.mse_loss(pred, b['target'])
    loss.backward()
    opt.step()
    if s % 20 == 0:
      curve.append(round(O.psnr_batch(pred.detach(), b['target']), 4))
  with torch.no_grad():
    # evaluation with the same storage format as training (what the product's forward would do)
    ps = [O.psnr_batch(forward(P, b['inp'], b['kspace'], b['mask'], on), b['target']) for b in held]
    # and the trained fp32 master weights evaluated in plain fp32
    ps32 = [O.psnr_batch(forward(P, b['inp'], b['kspace'], b['mask'], ()), b['target']) for b in held]
  return {'final_psnr_heldout_db': sum(ps) / len(ps), 'final_psnr_heldout_fp32_eval_db': sum(ps32) / len(ps32),
          'wall_s': round(time.time() - t0, 1), 'train_psnr_every_20_steps': curve}


def main():
  p = argparse.ArgumentParser()
  p.add_argument('--size', type=int, default=128)
  p.add_argument('--batch', type=int, default=8)
  p.add_argument('--steps', type=int, default=2000)
  p.add_argument('--distinct', type=int, default=32)
  p.add_argument('--heldout', type=int, default=4)
  p.add_argument('--seed', type=int, default=1)
  p.add_argument('--threads', type=int, default=6)
  p.add_argument('--replicas', type=int, default=2, help='members per variant (perturbation seeds 0..n-1)')
  p.add_argument('--variants', default='fp32;all;all-IN-GIN;all-W;all-ACT;all-GACT-GOUT;IN+GIN;W;ACT')
  p.add_argument('--out', default=None)
  args = p.parse_args()
  torch.set_num_threads(args.threads)
  train = [O.synth_batch(args.batch, args.size, args.size, acc=4, seed=5000 + args.seed + 1000 * i)
           for i in range(args.distinct)]
  held = [O.synth_batch(args.batch, args.size, args.size, acc=4, seed=900000 + args.seed + 1000 * i)
          for i in range(args.heldout)]
  out = {'size': args.size, 'batch': args.batch, 'steps': args.steps, 'classes': CLASSES, 'variants': {}}
  for v in args.variants.split(';'):
    if v == 'fp32':
      on = ()
    elif v.startswith('all'):
      on = tuple(c for c in CLASSES if c not in v.split('-')[1:])
    else:
      on = tuple(v.split('+'))
    members = []
    for k in range(args.replicas):
      r = run(on, args, train, held, k)
      members.append(r)
      print('%-18s member %d: held-out PSNR %.4f dB (fp32 eval of the masters %.4f)  %.0f s' %
            (v, k, r['final_psnr_heldout_db'], r['final_psnr_heldout_fp32_eval_db'], r['wall_s']), flush=True)
    vals = [m['final_psnr_heldout_db'] for m in members]
    out['variants'][v] = {'rounded': list(on), 'members': members, 'mean_db': sum(vals) / len(vals)}
    if args.out:
      with open(args.out, 'w') as f:
        json.dump(out, f)
  base = out['variants'].get('fp32', {}).get('mean_db')
  if base is not None:
    for v, r in out['variants'].items():
      r['delta_vs_fp32_db'] = r['mean_db'] - base
      print('%-18s mean %.4f dB  delta vs fp32 %+.4f dB' % (v, r['mean_db'], r['delta_vs_fp32_db']))
  if args.out:
    with open(args.out, 'w') as f:
      json.dump(out, f)


if __name__ == '__main__':
  main()
