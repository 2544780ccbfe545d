"""Diagnostic: D-phase gradients, HIP vs oracle, inputs captured from the HIP run."""
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
set_default_compute_dtype('fp32')
runner = build_runner(gan_conf(Configuration, 'fp32'), 'adversarial', '0', 'train')
runner.gen.load_state_dict(sub(f, 'G0.')); runner.disc.load_state_dict(sub(f, 'D0.')); ops.bump_weight_epoch()
masks = [T(f['step0.mask%d' % j]) for j in range(9)]
runner.disc.injected_dropout = list(masks)
cap = {}
orig_disc_fwd = runner.disc.forward
calls = []
def fwd(inp=None, nhwc=None):
  calls.append(nhwc.detach().float().cpu()[..., :1].permute(0, 3, 1, 2).contiguous())
  return orig_disc_fwd(inp, nhwc)
runner.disc.forward = fwd
grads = {}
orig = runner.disc_optimizer.step
names = {id(p): n for n, p in runner.disc.named_parameters()}
def step():
  grads.update({names[id(p)]: p.grad.detach().cpu().clone() for p in runner.disc_optimizer.params})
  orig()
runner.disc_optimizer.step = step
batch = O.synth_batch(2, 128, 128, acc=4, seed=40)
runner.train_epoch(Loader([batch]), 1)
small_disc = dict(O.DISC_CONF, filters=[8, 16, 32, 64, 64, 64])
PD = {k: v.clone().requires_grad_(True) for k, v in sub(f, 'D0.').items() if 'running' not in k and 'num_batches' not in k}
SD = {k: v.clone() for k, v in sub(f, 'D0.').items() if 'running' in k}
of = O.disc_forward(PD, SD, calls[0], True, small_disc, dropout_masks=masks[0:3])
orr = O.disc_forward(PD, SD, calls[1], True, small_disc, dropout_masks=masks[3:6])
ld = O.gan_loss_disc(of, orr, 0.1)
ld.backward()
for k, p in PD.items():
  g = p.grad
  print('%-22s rel_l2 %.3e' % (k, float((grads[k] - g).norm() / g.norm())))
# are the captured inputs what the oracle would feed?
small_unet = dict(O.UNET_CONF, encode_filters=[8, 16, 32], decode_filters=[16, 8])
PG = {k: v for k, v in sub(f, 'G0.').items() if 'running' not in k and 'num_batches' not in k}
SG = {k: v.clone() for k, v in sub(f, 'G0.').items() if 'running' in k}
O.unet_forward.__defaults__ = tuple(small_unet if isinstance(x, dict) else x for x in O.unet_forward.__defaults__)
with torch.no_grad():
  og = O.refinement_forward(PG, SG, batch['inp'], batch['kspace'], batch['mask'], True)
fa = O.complex_abs(og['pred']); re = O.complex_abs(batch['target'])
print('fake_in max diff %.3e  real_in max diff %.3e' % (float((fa - calls[0]).abs().max()), float((re - calls[1]).abs().max())))
