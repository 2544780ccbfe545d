#!/usr/bin/env python3
"""Where the bf16 FORWARD pass of a trained RecNet loses PSNR (CPU oracle; test infrastructure, not collected by pytest).

Input: state dicts of RecNet(5,3,32) trained by tools/trajectory.py --save-weights on the GPU (fp32 run and bf16 run).
For each, the held-out PSNR of the oracle's forward pass with every subset of the bf16 storage roundings of the product's
forward -- IN (block input), W (packed weights), ACT (the two 32-channel activations) -- next to plain fp32.

  python tests/c2_forward_floor.py gpurun_out/r04/weights/c2_fp32.pth gpurun_out/r04/weights/c2_bf16.pth
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import csmri_oracle as O
from c2_format_floor import forward

size, batch, seed, heldout = 256, 16, 1, 4
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
from data.synthetic import synth_batch      # (host-side synthetic data of the product: no GPU involved)
held = [synth_batch(batch, size, size, acc=4, seed=900000 + seed + 1000 * i) for i in range(heldout)]
out = {}
torch.set_num_threads(8)
for path in sys.argv[1:]:
  P = torch.load(path)
  res = {}
  for on in ((), ('IN',), ('W',), ('ACT',), ('IN', 'W'), ('W', 'ACT'), ('IN', 'ACT'), ('IN', 'W', 'ACT')):
    with torch.no_grad():
      ps = [O.psnr_batch(forward(P, b['inp'], b['kspace'], b['mask'], on), b['target']) for b in held]
    res['+'.join(on) or 'fp32'] = sum(ps) / len(ps)
    print('%-28s %-10s %.4f dB  (%+.4f vs fp32 forward)' % (os.path.basename(path), '+'.join(on) or 'fp32', res['+'.join(on) or 'fp32'],
                                                           res['+'.join(on) or 'fp32'] - res['fp32']), flush=True)
  out[os.path.basename(path)] = res
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'r04', 'c2_forward_floor.json'), 'w'), indent=1)
