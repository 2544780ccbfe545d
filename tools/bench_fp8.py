"""fp8 vs bf16 forward convolution at the workload's eligible layer shapes: kernel alone (HIP events around
the csmri_gconv launch) and the whole call (absmax + quantise + conv for fp8)."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
from csmri_hip import ops

S = int(os.environ.get('SIZE', '256'))
CASES = [  # name, cin, cout, k, stride, border, H, B
    ('vgg2_2', 128, 128, 3, 1, 'zero', S // 2, 16),
    ('vgg3_1', 128, 256, 3, 1, 'zero', S // 4, 16),
    ('vgg3_2', 256, 256, 3, 1, 'zero', S // 4, 16),
    ('vgg4_1', 256, 512, 3, 1, 'zero', S // 8, 16),
    ('vgg4_2', 512, 512, 3, 1, 'zero', S // 8, 16),
    ('unet_e2b', 128, 128, 4, 1, 'reflection', S // 4, 8),
    ('disc3', 128, 256, 4, 2, 'reflection', S // 4, 16),
    ('disc4', 256, 512, 4, 2, 'reflection', S // 8, 16),
    ('disc5', 512, 1024, 4, 2, 'reflection', S // 16, 16),
    ('disc6', 1024, 1024, 4, 1, 'reflection', S // 32, 16),
]
def same_pad(k, s):
  t = max(k - s, 0); return (t // 2, t - t // 2, t // 2, t - t // 2)
g = torch.Generator().manual_seed(0)
print('%-10s %9s %9s | %9s %9s | %7s %7s' % ('layer', 'bf16 us', 'TF', 'fp8 us', 'TF', 'call16', 'call8'))
for name, cin, cout, k, stride, border, h, b in CASES:
  wt = torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)
  layer = ops.ConvLayer(torch.nn.Parameter(wt.cuda()), None, stride, same_pad(k, stride), border, torch.bfloat16)
  x = torch.randn(b, h, h, cin, generator=g).bfloat16().cuda()
  res = {}
  for fp8 in (False, True):
    layer.fp8 = fp8
    for _ in range(3):
      ops.conv_forward(layer, x, None, False, 0.2, False, None)
    torch.cuda.synchronize()
    ops.PROFILE = []
    for _ in range(20):
      y, _ = ops.conv_forward(layer, x, None, False, 0.2, False, None)
    torch.cuda.synchronize()
    recs = ops.PROFILE; ops.PROFILE = None
    kern = sorted(r[2].elapsed_time(r[3]) * 1e3 for r in recs if 'reduce' not in r[0])
    red = sorted(r[2].elapsed_time(r[3]) * 1e3 for r in recs if 'reduce' in r[0])
    t_k = kern[len(kern) // 2] + (red[len(red) // 2] if red else 0.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
      ops.conv_forward(layer, x, None, False, 0.2, False, None)
    e1.record(); torch.cuda.synchronize()
    res[fp8] = (t_k, e0.elapsed_time(e1) * 1e3 / 50, recs[0][0])
  ho = y.shape[1]
  fl = 2.0 * b * ho * ho * cout * cin * k * k
  print('%-10s %9.1f %9.0f | %9.1f %9.0f | %7.1f %7.1f   %s / %s' % (
      name, res[False][0], fl / res[False][0] / 1e6, res[True][0], fl / res[True][0] / 1e6,
      res[False][1], res[True][1], res[False][2], res[True][2]))
