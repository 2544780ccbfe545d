"""Diagnostic (checker use of the oracle only): what an fp8 forward of the frozen VGG19 (ops.Fp8Chain) does to the perceptual
loss and to its gradient w.r.t. the prediction, as a function of how close the prediction is to its target.
profiles/r05_fp8_vgg.log section 3.  usage: python tools/probe_fp8_vgg_gradient.py"""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import torch, torch.nn.functional as F
import csmri_oracle as O
from csmri_hip import ops
from models.utils import set_default_compute_dtype
from models.vgg import VGG19
def rel_l2(a, b): return float((a - b).norm() / (b.norm() + 1e-30))
mean, std = torch.tensor(O.VGG_MEAN).view(1, 3, 1, 1), torch.tensor(O.VGG_STD).view(1, 3, 1, 1)
for H, B in ((128, 2), (256, 4)):
  for sigma in (None, 0.1, 0.03, 0.01):
    g = torch.Generator().manual_seed(17)
    t = torch.rand(B, 1, H, H, generator=g).repeat(1, 3, 1, 1)
    # smooth-ish target: blur
    t = F.avg_pool2d(F.pad(t, (2, 2, 2, 2), mode='reflect'), 5, 1)
    p = torch.rand(B, 3, H, H, generator=g) if sigma is None else (t + sigma * torch.randn(B, 1, H, H, generator=g)).clamp(0, 1)
    out = {}
    for mode in ('bf16', 'fp8'):
      set_default_compute_dtype(mode)
      vgg = VGG19(seed=3).cuda()
      set_default_compute_dtype('bf16')
      for it in range(2):
        pd = ops.ToNHWC.apply(((p - mean) / std).cuda().requires_grad_(True), torch.bfloat16, 8); pd.retain_grad()
        td = ops.ToNHWC.apply(((t - mean) / std).cuda(), torch.bfloat16, 8)
        fp, ft = vgg.features_pair(pd, td)
        loss = ((fp[0].float() - ft[0].float()) ** 2).mean()
        loss.backward(); torch.cuda.synchronize()
      out[mode] = (float(loss), pd.grad.float().cpu()[..., :3])
      if mode == 'bf16':
        PV = {k: v.detach().cpu().float() for k, v in vgg.state_dict().items() if k.startswith('blocks.')}
    pr = ((p - mean) / std).requires_grad_(True)
    lo = F.mse_loss(O.vgg_features(PV, pr * std + mean), O.vgg_features(PV, t).detach()); lo.backward()
    go = pr.grad.permute(0, 2, 3, 1)
    cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
    print('H %d B %d sigma %s: loss oracle %.4e bf16 %.4e fp8 %.4e | grad cos vs oracle: bf16 %.4f fp8 %.4f | fp8 vs bf16 %.4f' %
          (H, B, sigma, float(lo), out['bf16'][0], out['fp8'][0], cos(out['bf16'][1], go), cos(out['fp8'][1], go), cos(out['fp8'][1], out['bf16'][1])))
