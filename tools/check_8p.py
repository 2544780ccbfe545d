"""gconv8p (phased 256 x 256 kernel) against the kernels it replaces: same inputs in two processes
(CSMRI_NO_8P toggles the dispatch), outputs compared, every output of 30 repeated launches compared
bit for bit with the first (a pipeline race shows up as run-to-run differences), and timings.
usage: python tools/check_8p.py [case ...]"""
import os, sys, math, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch

CASES = {
    # name: (cin, cout, k, stride, border, up, H, W, B)
    'vgg3_2b16': (256, 256, 3, 1, 'zero', False, 64, 64, 16),
    'vgg3_2b8': (256, 256, 3, 1, 'zero', False, 64, 64, 8),
    'vgg4_1b16': (256, 512, 3, 1, 'zero', False, 32, 32, 16),
    'vgg4_2b16': (512, 512, 3, 1, 'zero', False, 32, 32, 16),
    'vgg4_2b8': (512, 512, 3, 1, 'zero', False, 32, 32, 8),
    'vgg5_2b16': (512, 512, 3, 1, 'zero', False, 16, 16, 16),
    'vgg5_2b8': (512, 512, 3, 1, 'zero', False, 16, 16, 8),
    'disc3b16': (128, 256, 4, 2, 'reflection', False, 64, 64, 16),
    'disc4b16': (256, 512, 4, 2, 'reflection', False, 32, 32, 16),
    'disc5b16': (512, 1024, 4, 2, 'reflection', False, 16, 16, 16),
    'disc6b16': (1024, 1024, 4, 1, 'reflection', False, 8, 8, 16),
    'disc4b8': (256, 512, 4, 2, 'reflection', False, 32, 32, 8),
    'disc6b8': (1024, 1024, 4, 1, 'reflection', False, 8, 8, 8),
    'odd_m': (256, 256, 3, 1, 'zero', False, 61, 37, 3),          # partial last tile, non-power-of-two map
}


def pads_for(k, s):
  total = int(math.ceil((k - 1.0) / s)); lo = total // 2; hi = lo if total % 2 == 0 else lo + 1
  return (lo, hi, lo, hi)


def dump(path, names):
  from csmri_hip import ops
  out = {}
  for name in names:
    cin, cout, k, s, border, up, h, w, b = CASES[name]
    g = torch.Generator().manual_seed(5)
    wt = torch.nn.Parameter((torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).cuda())
    bias = torch.nn.Parameter(torch.randn(cout, generator=g).cuda())
    layer = ops.ConvLayer(wt, bias, s, pads_for(k, s), border, torch.bfloat16, upsample=up)
    x = torch.randn(b, h, w, ops.pad8(cin), generator=g).bfloat16().cuda()
    log = ops.LAUNCH_LOG = []
    y, _ = ops.conv_forward(layer, x, None, True, 0.2)
    gy = torch.randn(y.shape, generator=g).bfloat16().cuda()
    dx = ops.conv_dgrad(layer, gy, (h, w))
    ops.LAUNCH_LOG = None
    torch.cuda.synchronize()
    res = {'y': y.float().cpu(), 'dx': dx.float().cpu(), 'kern': [e[1:] for e in log]}
    # repeatability
    bad = [0, 0]
    for _ in range(30):
      y2, _ = ops.conv_forward(layer, x, None, True, 0.2)
      dx2 = ops.conv_dgrad(layer, gy, (h, w))
      bad[0] += int(not torch.equal(y2, y)); bad[1] += int(not torch.equal(dx2, dx))
    res['unstable'] = bad
    # timing
    for mode, fn in (('fwd', lambda: ops.conv_forward(layer, x, None, True, 0.2)),
                     ('dgrad', lambda: ops.conv_dgrad(layer, gy, (h, w)))):
      for _ in range(3):
        fn()
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(30):
        fn()
      e1.record(); torch.cuda.synchronize()
      us = e0.elapsed_time(e1) / 30 * 1e3
      res[mode + '_us'] = us
      res[mode + '_tf'] = 2.0 * b * y.shape[1] * y.shape[2] * cout * cin * k * k / us / 1e6
    out[name] = res
  torch.save(out, path)


if __name__ == '__main__':
  if len(sys.argv) > 2 and sys.argv[1] == 'dump':
    dump(sys.argv[2], sys.argv[3:])
  else:
    names = [a for a in sys.argv[1:] if a in CASES] or list(CASES)
    env = dict(os.environ, CSMRI_8P='1')
    subprocess.check_call([sys.executable, __file__, 'dump', '/tmp/g8p_a.pt'] + names, env=env)
    env['CSMRI_NO_8P'] = '1'
    subprocess.check_call([sys.executable, __file__, 'dump', '/tmp/g8p_b.pt'] + names, env=env)
    a, b = torch.load('/tmp/g8p_a.pt'), torch.load('/tmp/g8p_b.pt')
    for k in names:
      for what in ('y', 'dx'):
        ref = b[k][what]
        rel = float((a[k][what] - ref).norm() / ref.norm())
        print('%-10s %-2s rel_l2 vs old %.2e  unstable runs %s' % (k, what, rel, a[k]['unstable']))
      print('%-10s fwd   %7.1f us %7.1f TF  (old %7.1f us %7.1f TF)   %s | %s' %
            (k, a[k]['fwd_us'], a[k]['fwd_tf'], b[k]['fwd_us'], b[k]['fwd_tf'], a[k]['kern'][0], b[k]['kern'][0]))
      print('%-10s dgrad %7.1f us %7.1f TF  (old %7.1f us %7.1f TF)   %s | %s' %
            (k, a[k]['dgrad_us'], a[k]['dgrad_tf'], b[k]['dgrad_us'], b[k]['dgrad_tf'], a[k]['kern'][1:], b[k]['kern'][1:]))
