#!/usr/bin/env python3
"""What the compute dtype does to TRAINING, not to one step (SURVEY 8d "bf16/fp8 judged on
PSNR + loss curves"; reference training/runner.py:154-178, training/adversarial_runner.py:322-389).

The same run is executed twice through the product path -- fp32 compute (the path pinned to the CPU oracle at
2e-6 by tests/test_hip_path.py) and bf16 compute (optionally fp8) -- from the same initial weights, over the same
sequence of batches, with the same Dropout2d masks (csmri_dropout2d_mask: same Philox seed and call counter) and the
same image-pool decisions (python `random`, re-seeded), eager launches:

  c2   RecNet(5 blocks, 3 convs, 32 filters) MSE training (incl. the DC adjoints)
  c3   2-refinement GAN step; the frozen RecNet(3,3,32) is first trained for --pretrain-steps fp32 MSE steps and
       handed to both runs through the reference's pretrained_weights mechanism
`--dtypes fp32,bf16,fp32p`: fp32p is the CONTROL, the fp32 run again from initial weights perturbed by 1e-6 relative --
how far two fp32 trajectories drift apart by themselves.  `fp32p1`, `fp32p2`, ..., `bf16p1`, ...: further members of
the ensemble (perturbation seed k); `--ensemble N` appends N perturbed members per dtype and `summary.ensemble`
compares the two DISTRIBUTIONS of the final held-out PSNR (mean, standard deviation, Welch t) -- one pair of
trajectories on the steep part of a learning curve differs by a time shift of a few steps, which is not a statement
about the format.

Recorded per dtype: every step's losses and training PSNR (curves down-sampled to <= 250 points), and at the end
the PSNR of the trained model on held-out batches (train-mode BatchNorm for the GAN generator as during training,
plus eval mode).  Written as one JSON file; `summary` holds the final |delta PSNR| and the largest relative distance
of the smoothed loss curves.

  python tools/trajectory.py --config c2 --steps 500 --out profiles/r03_trajectory_c2.json
  python tools/trajectory.py --config c3 --steps 500 --out profiles/r03_trajectory_c3.json
"""
import argparse
import json
import os
import random
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'csmri-refinement_amd')
sys.path.insert(0, PKG)


def build(config, dtype, batch, width, seed, pretrained=None):
  import warnings
  import torch
  import utils
  from utils.config import Configuration
  from models.utils import set_default_compute_dtype
  from training import build_runner
  set_default_compute_dtype(dtype)
  if config == 'c2':
    conf = Configuration.from_json(os.path.join(PKG, 'configs', '1-recnet.json'))
    conf.model.update(num_blocks=5, num_convs=3, num_filters=32, compute_dtype=dtype)
    kind = 'standard'
  elif config == 'recnet3':
    conf = Configuration.from_json(os.path.join(PKG, 'configs', '1-recnet.json'))
    conf.model.update(num_blocks=3, num_convs=3, num_filters=32, compute_dtype=dtype)
    kind = 'standard'
  else:
    conf = Configuration.from_json(os.path.join(PKG, 'configs', '2-refinement.json'))
    g, d = conf.generator_model, conf.discriminator_model
    for m in (g['pretrained_model'], g['learnable_model'], d):
      m['compute_dtype'] = dtype
    if width != 'full':       # reduced widths (tests): every structure of the step, a fraction of the FLOPs
      g['learnable_model'].update(encode_filters=[8, 16, 32], decode_filters=[16, 8])
      d['num_filters_per_layer'] = [8, 16, 32, 64, 64, 64]
    if pretrained is not None:
      g['pretrained_model']['pretrained_weights'] = [pretrained, 'model']
    kind = 'adversarial'
  conf.batch_size = batch
  utils.set_random_seeds(seed)
  torch.manual_seed(seed)
  random.seed(seed)
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    return build_runner(conf, kind, '0', 'train'), conf


def make_batches(n, batch, size, seed, dev):
  from data.synthetic import synth_batch
  return [{k: v.to(dev) for k, v in synth_batch(batch, size, size, acc=4, seed=seed + 1000 * i).items()}
          for i in range(n)]


class Loader(list):
  batch_size = 1


def psnr_of(runner, batches, train_mode):
  """mean PSNR (reference metrics/image_metrics.py:7-19 through the product's metric kernel) over batches."""
  import torch
  from csmri_hip import ops
  model = runner.gen if hasattr(runner, 'gen') else runner.model
  model.train(train_mode)
  vals = []
  with torch.no_grad():
    for b in batches:
      out = model(b['inp'], b['kspace'], b['mask'])
      pred = out['pred'] if isinstance(out, dict) else out
      mse = ops.psnr_mse(ops.nchw_to_nhwc(pred.float(), torch.float32, 2), ops.nchw_to_nhwc(b['target'], torch.float32, 2))
      vals.append(float((10.0 * torch.log10(1.0 / mse.double())).mean()))
  model.train(True)
  return sum(vals) / len(vals)


def run(config, dtype, args, train, held, pretrained, dev):
  import torch
  # 'fp32p', 'fp32p3', 'bf16p2' ...: the run from initial weights perturbed by 1e-6 relative, perturbation seed k
  import re
  m = re.match(r'^(fp32|bf16|fp8)p(\d*)$', dtype)
  perturbed = m is not None
  pseed = 0
  if perturbed:
    dtype, pseed = m.group(1), int(m.group(2) or 0)
  runner, conf = build(config, dtype, args.batch, args.width, args.seed, pretrained)
  if perturbed:
    # CONTROL: the fp32 run again from initial weights moved by 1e-6 relative (a few fp32 ulps): how far two fp32
    # trajectories drift apart by themselves -- the yardstick for the bf16 run's distance
    gen = torch.Generator(device='cpu').manual_seed(args.seed + 77 + 1009 * pseed)
    with torch.no_grad():
      for net in (getattr(runner, 'gen', None), getattr(runner, 'disc', None), getattr(runner, 'model', None)):
        if net is None:
          continue
        for prm in net.parameters():
          if prm.requires_grad:
            prm.mul_(1.0 + 1e-6 * torch.randn(prm.shape, generator=gen).to(prm.device))
    from csmri_hip import ops
    ops.bump_weight_epoch()
  if dtype == 'fp8':
    from models.utils import set_fp8_forward
    for net in (getattr(runner, 'gen', None), getattr(runner, 'disc', None), getattr(runner, 'model', None)):
      if net is not None:
        set_fp8_forward(net, True)
  random.seed(args.seed + 1)
  torch.manual_seed(args.seed + 1)
  curves = {}
  t0 = time.time()
  for s in range(args.steps):
    losses, metrics = runner.train_epoch(Loader([train[s % len(train)]]), 1, steps_per_train_summary=10 ** 9)
    for k, v in list(losses.items()) + list(metrics.items()):
      curves.setdefault(k, []).append(float(v.value))
  torch.cuda.synchronize()
  res = {'curves': curves, 'wall_s': round(time.time() - t0, 1),
         'final_psnr_heldout_train_bn': psnr_of(runner, held, True),
         'final_psnr_heldout_eval': psnr_of(runner, held, False),
         'final_psnr_train_batches': psnr_of(runner, train[:len(held)], True)}
  if config == 'c2' and dtype != 'fp32':
    # the SAME trained fp32 master weights evaluated through the fp32 compute path: separates what the storage format did
    # to the TRAINING (the weights it arrived at) from what it does to the FORWARD pass that evaluates them
    r32, _ = build(config, 'fp32', args.batch, args.width, args.seed, pretrained)
    r32.model.load_state_dict(runner.model.state_dict())
    from csmri_hip import ops
    ops.bump_weight_epoch()
    res['final_psnr_heldout_eval_fp32_compute_of_these_weights'] = psnr_of(r32, held, False)
    del r32
  if getattr(args, 'save_weights', None):
    model = runner.gen if hasattr(runner, 'gen') else runner.model
    os.makedirs(args.save_weights, exist_ok=True)
    torch.save({k: v.detach().float().cpu() for k, v in model.state_dict().items()},
               os.path.join(args.save_weights, '%s_%s.pth' % (config, dtype + ('p%d' % pseed if perturbed else ''))))
  return res, runner, conf


def smooth(v, w):
  out, acc = [], 0.0
  for i, x in enumerate(v):
    acc += x
    if i >= w:
      acc -= v[i - w]
    out.append(acc / min(i + 1, w))
  return out


def main(argv=None):
  p = argparse.ArgumentParser()
  p.add_argument('--config', default='c2', choices=['c2', 'c3'])
  p.add_argument('--steps', type=int, default=500)
  p.add_argument('--size', type=int, default=256)
  p.add_argument('--batch', type=int, default=0)
  p.add_argument('--distinct', type=int, default=32, help='distinct training batches, cycled')
  p.add_argument('--heldout', type=int, default=4)
  p.add_argument('--pretrain-steps', type=int, default=300)
  p.add_argument('--width', default='full', choices=['full', 'reduced'])
  p.add_argument('--dtypes', default='fp32,bf16')
  p.add_argument('--seed', type=int, default=1)
  p.add_argument('--ensemble', type=int, default=0,
                 help='append this many perturbed members (fp32p1.., bf16p1..) per base dtype in --dtypes')
  p.add_argument('--save-weights', default=None, help='directory for the trained state dicts (one small file per run)')
  p.add_argument('--variant', default='', help='free-text tag of the product variant under test (recorded)')
  p.add_argument('--out', default=None)
  args = p.parse_args(argv)
  if args.batch <= 0:
    args.batch = 16 if args.config == 'c2' else 8
  import torch
  import csmri_hip  # noqa: F401
  dev = torch.device('cuda', 0)
  train = make_batches(args.distinct, args.batch, args.size, 5000 + args.seed, dev)
  held = make_batches(args.heldout, args.batch, args.size, 900000 + args.seed, dev)
  pretrained = None
  info = {}
  if args.config == 'c3':
    # the frozen reconstruction network both runs refine: a short fp32 MSE training of RecNet(3,3,32), saved and
    # loaded through the reference's checkpoint / pretrained_weights mechanism (configs/2-refinement.json:29)
    from utils.checkpoints import save_checkpoint
    r, c = build('recnet3', 'fp32', args.batch, args.width, args.seed)
    for s in range(args.pretrain_steps):
      r.train_epoch(Loader([train[s % len(train)]]), 1, steps_per_train_summary=10 ** 9)
    pretrained = os.path.join(tempfile.mkdtemp(prefix='csmri_traj_'), 'recnet3.pth')
    save_checkpoint(pretrained, c, r, 1, None)
    info['pretrained_recnet_psnr_heldout'] = psnr_of(r, held, True)
    del r
  results = {}
  dts = args.dtypes.split(',')
  if args.ensemble > 0:
    for base in [d for d in dts if d in ('fp32', 'bf16')]:
      dts += ['%sp%d' % (base, k) for k in range(1, args.ensemble + 1)]
  for dt in dts:
    results[dt], _, _ = run(args.config, dt, args, train, held, pretrained, dev)
  ref = results['fp32']
  summary = {}
  w = max(1, args.steps // 20)
  for dt, r in results.items():
    if dt == 'fp32':
      continue
    s = {'final_delta_psnr_heldout_train_bn_db': abs(r['final_psnr_heldout_train_bn'] - ref['final_psnr_heldout_train_bn']),
         'final_delta_psnr_heldout_eval_db': abs(r['final_psnr_heldout_eval'] - ref['final_psnr_heldout_eval']),
         'final_delta_psnr_train_batches_db': abs(r['final_psnr_train_batches'] - ref['final_psnr_train_batches'])}
    for k, v in r['curves'].items():
      a, b = smooth(ref['curves'][k], w), smooth(v, w)
      tail = range(len(a) // 10, len(a))
      if 'psnr' in k:
        s['max_delta_smoothed_' + k + '_db'] = max(abs(a[i] - b[i]) for i in tail)
        s['final_delta_smoothed_' + k + '_db'] = abs(a[-1] - b[-1])
      else:
        s['max_rel_delta_smoothed_' + k] = max(abs(a[i] - b[i]) / max(abs(a[i]), 1e-12) for i in tail)
        s['final_rel_delta_smoothed_' + k] = abs(a[-1] - b[-1]) / max(abs(a[-1]), 1e-12)
    summary[dt] = s
  # ensemble view: members of a base dtype = the unperturbed run + its perturbed replicas
  import math
  import re
  ens = {}
  for dt, r in results.items():
    base = re.match(r'^(fp32|bf16|fp8)', dt).group(1)
    ens.setdefault(base, []).append(r['final_psnr_heldout_eval'])
  ens_summary = {}
  for base, v in ens.items():
    n = len(v)
    mean = sum(v) / n
    var = sum((x - mean) ** 2 for x in v) / (n - 1) if n > 1 else 0.0
    ens_summary[base] = {'n': n, 'mean_final_psnr_heldout_eval_db': mean, 'std_db': math.sqrt(var),
                         'min_db': min(v), 'max_db': max(v), 'members_db': [round(x, 5) for x in v]}
  if 'fp32' in ens_summary and 'bf16' in ens_summary and min(ens_summary['fp32']['n'], ens_summary['bf16']['n']) > 1:
    a, b = ens_summary['fp32'], ens_summary['bf16']
    se = math.sqrt(a['std_db'] ** 2 / a['n'] + b['std_db'] ** 2 / b['n'])
    ens_summary['bf16_minus_fp32'] = {'delta_of_means_db': b['mean_final_psnr_heldout_eval_db'] - a['mean_final_psnr_heldout_eval_db'],
                                      'standard_error_db': se,
                                      'welch_t': (b['mean_final_psnr_heldout_eval_db'] - a['mean_final_psnr_heldout_eval_db']) / se if se > 0 else None}
  summary['ensemble'] = ens_summary
  stride = max(1, args.steps // 250)
  out = {'config': args.config, 'variant': args.variant, 'steps': args.steps, 'batch': args.batch, 'size': args.size, 'width': args.width,
         'distinct_train_batches': args.distinct, 'heldout_batches': args.heldout, 'seed': args.seed,
         'smoothing_window_steps': w, 'curve_stride': stride, 'info': info, 'summary': summary,
         'runs': {dt: dict(r, curves={k: [round(x, 7) for x in v[::stride]] for k, v in r['curves'].items()})
                  for dt, r in results.items()}}
  txt = json.dumps(out)
  if args.out:
    with open(args.out, 'w') as f:
      f.write(txt)
  print(json.dumps({'summary': summary, 'final': {dt: {k: v for k, v in r.items() if k != 'curves'} for dt, r in results.items()},
                    'info': info}, indent=1))
  return out


if __name__ == '__main__':
  main()
