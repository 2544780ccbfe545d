"""Split-K sweep of the small-M convolution launches of the C3 step (main kernel + csmri_gconv_reduce, HIP events,
back-to-back launches): what csmri_gconv_suggest_splitk should return for the shapes of the three-group discriminator pass.
  python tools/sk_sweep.py"""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
from csmri_hip import ops
ops.GCONV_FLAGS = int(os.environ.get('GCONV_FLAGS', '0'))      # 2 = gpipe where eligible, +4 / +8 = 256- / 192-row tiles
ONLY = [a for a in sys.argv[1:]]

S = 256
CASES = [   # name, cin, cout, k, stride, border, H, B, mode
    ('disc3_b24 fwd', 128, 256, 4, 2, 'reflection', S // 4, 24, 'fwd'),
    ('disc4_b24 fwd', 256, 512, 4, 2, 'reflection', S // 8, 24, 'fwd'),
    ('disc5_b24 fwd', 512, 1024, 4, 2, 'reflection', S // 16, 24, 'fwd'),
    ('disc6_b24 fwd', 1024, 1024, 4, 1, 'reflection', S // 32, 24, 'fwd'),
    ('disc4_b16 dgrad', 256, 512, 4, 2, 'reflection', S // 8, 16, 'dgrad'),
    ('disc5_b16 dgrad', 512, 1024, 4, 2, 'reflection', S // 16, 16, 'dgrad'),
    ('disc6_b16 dgrad', 1024, 1024, 4, 1, 'reflection', S // 32, 16, 'dgrad'),
    ('disc3_b8 dgrad', 128, 256, 4, 2, 'reflection', S // 4, 8, 'dgrad'),
    ('disc4_b8 dgrad', 256, 512, 4, 2, 'reflection', S // 8, 8, 'dgrad'),
    ('disc5_b8 dgrad', 512, 1024, 4, 2, 'reflection', S // 16, 8, 'dgrad'),
    ('disc6_b8 dgrad', 1024, 1024, 4, 1, 'reflection', S // 32, 8, 'dgrad'),
    ('vgg5_1_b16 fwd', 512, 512, 3, 1, 'zero', S // 16, 16, 'fwd'),
    ('vgg5_1_b8 dgrad', 512, 512, 3, 1, 'zero', S // 16, 8, 'dgrad'),
    ('vgg4_2_b8 dgrad', 512, 512, 3, 1, 'zero', S // 8, 8, 'dgrad'),
    ('vgg4_1_b8 dgrad', 256, 512, 3, 1, 'zero', S // 8, 8, 'dgrad'),
    ('unet_e2b_b8 fwd', 128, 128, 4, 1, 'reflection', S // 4, 8, 'fwd'),
    ('unet_e2b_b8 dgrad', 128, 128, 4, 1, 'reflection', S // 4, 8, 'dgrad'),
]


def pads_for(k, s):
  total = int(math.ceil((k - 1.0) / s)); lo = total // 2; hi = lo if total % 2 == 0 else lo + 1
  return (lo, hi, lo, hi)


def timeit(fn, iters=30):
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters):
    fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e3


for name, cin, cout, k, s, border, h, b, mode in CASES:
  if ONLY and not any(o in name for o in ONLY):
    continue
  wt = torch.nn.Parameter((torch.randn(cout, cin, k, k) / math.sqrt(cin * k * k)).cuda())
  layer = ops.ConvLayer(wt, None, s, pads_for(k, s), border, torch.bfloat16)
  x = torch.randn(b, h, h, ops.pad8(cin), device='cuda').bfloat16()
  y, _ = ops.conv_forward(layer, x, None, False)
  gy = torch.randn_like(y)
  fn = (lambda: ops.conv_forward(layer, x, None, False)) if mode == 'fwd' else (lambda: ops.conv_dgrad(layer, gy, (h, h)))
  res = []
  log = ops.LAUNCH_LOG = []
  ops.SPLITK_OVERRIDE = None
  fn()
  base = (log[0][1], log[0][2])
  ops.LAUNCH_LOG = None
  t0 = timeit(fn)
  for sk in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
    ops.SPLITK_OVERRIDE = sk
    log = ops.LAUNCH_LOG = []
    try:
      fn()
      kern = log[0][1]
      ops.LAUNCH_LOG = None
      res.append((sk, timeit(fn), kern.replace('gconv_glds_kernel', 'glds').replace('gpipe_kernel', 'gp')))
    except Exception as e:
      ops.LAUNCH_LOG = None
      res.append((sk, float('nan'), repr(e)[:30]))
  ops.SPLITK_OVERRIDE = None
  best = min(res, key=lambda r: r[1] if r[1] == r[1] else 1e9)
  print('%-18s default %s sk%d %.1f us | best sk%d %.1f us (%s) | ' % (name, base[0].replace('gconv_glds_kernel', 'glds'), base[1], t0, best[0], best[1], best[2]) +
        ' '.join('sk%d:%.1f' % (r[0], r[1]) for r in res), flush=True)
