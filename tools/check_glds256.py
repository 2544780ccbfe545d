"""256-tile conv kernel against the 128-tile one (same inputs, two processes' worth of env)."""
import os, sys, math, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
if len(sys.argv) > 1 and sys.argv[1] == 'dump':
  from csmri_hip import ops
  sys.path.insert(0, os.path.join(ROOT, 'tools'))
  import bench_conv as bc
  out = {}
  for name in ('vgg3_2b16', 'vgg4_2b16', 'vgg4_1b16', 'vgg3_2', 'vgg4_2'):
    cin, cout, k, s, border, up, h, w, b = bc.CASES[name]
    g = torch.Generator().manual_seed(5)
    wt = torch.nn.Parameter((torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).cuda())
    bias = torch.nn.Parameter(torch.randn(cout, generator=g).cuda())
    layer = ops.ConvLayer(wt, bias, s, bc.pads_for(k, s), border, torch.bfloat16, upsample=up)
    x = torch.randn(b, h, w, ops.pad8(cin), generator=g).bfloat16().cuda()
    y, _ = ops.conv_forward(layer, x, None, True, 0.2)
    gy = torch.randn(y.shape, generator=g).bfloat16().cuda()
    dx = ops.conv_dgrad(layer, gy, (h, w), g_src=x, g_slope=0.1)
    out[name] = (y.float().cpu(), dx.float().cpu())
  torch.save(out, sys.argv[2])
else:
  env = dict(os.environ)
  subprocess.check_call([sys.executable, __file__, 'dump', '/tmp/g256_a.pt'], env=env)
  env['CSMRI_NO_GLDS256'] = '1'
  subprocess.check_call([sys.executable, __file__, 'dump', '/tmp/g256_b.pt'], env=env)
  a, b = torch.load('/tmp/g256_a.pt'), torch.load('/tmp/g256_b.pt')
  for k in a:
    for i, what in enumerate(('fwd', 'dgrad')):
      d = (a[k][i] - b[k][i]).abs().max().item()
      print(k, what, 'max abs diff', d, 'ref max', b[k][i].abs().max().item(), 'equal' if d == 0 else '')
