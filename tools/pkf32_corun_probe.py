"""Reproducer: fp32 VALU kernels next to MFMA kernels in two branches of one hipGraph.

  python tools/pkf32_corun_probe.py <A> <B>      (GPU box)
    A: dc | convdc | lastconvdc      work of the checked branch (data-consistency layers)
    B: conv | conv1 | mm | myelem | elem   work of the other branch
  env: SIZE, BS, PLACE=side_side|main_side|side_main, ORDER=ab|ba, DETAIL=1, SPEC=1

Finding (round 1): with the library built WITH packed-fp32 VALU instructions (v_pk_*_f32), every
DC layer that overlaps in time with one of our MFMA conv kernels returned wrong values in
individual lanes (quarter-waves 0 and 3 of the last butterfly stage), deterministically, for any
stream placement; next to torch.mm / elementwise kernels, or alone, it was exact.  Coherence
fences, cache-bypassing loads and LDS padding changed nothing; rebuilding with
-target-feature -packed-fp32-ops (csrc/Makefile) made every combination exact.  The same check is
tests/test_hip_ops.py::test_fp32_kernels_exact_next_to_mfma_kernels_in_one_graph."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import csmri_oracle as O
from csmri_hip import ops
from utils.config import Configuration
from models.utils import set_default_compute_dtype
from training import build_runner
import utils
from test_hip_path import gan_conf

set_default_compute_dtype('bf16')
size, bs = int(os.environ.get("SIZE", 256)), int(os.environ.get("BS", 8))
A, B = sys.argv[1], sys.argv[2]
batch = {k: v.cuda() for k, v in O.synth_batch(bs, size, size, acc=4, seed=60).items()}
conf = gan_conf(Configuration, 'bf16')
utils.set_random_seeds(conf.seed)
r = build_runner(conf, 'adversarial', '0', 'train')
net = r.gen.pretrained_model
inp, kspace, mask = r.train_model_input_fn(batch)
x_pad = ops.ToNHWC.apply(inp, net.dtype, 8)
x_c = ops.ToNHWC.apply(inp, torch.float32, 2)
k0 = ops.nchw_to_nhwc(kspace, torch.float32, 2)
m8 = ops.mask_to_u8(mask)
block = net.conv_blocks[0]
big = torch.randn(64 << 20, device='cuda')
big_img = torch.randn(bs * 4, 2, size, size, device='cuda')
mm_a = torch.randn(4096, 4096, device='cuda', dtype=torch.bfloat16) * 0.01
mm_b = torch.randn(4096, 4096, device='cuda', dtype=torch.bfloat16) * 0.01

def convs(x, upto):
  for i in range(upto):
    cp = block.layers[str(3 * i + 1)]
    last = i == block.num_convs - 1
    x = ops.ConvAct.apply(x, None, cp.weight, cp.bias, cp.layer, 1.0 if last else block.slope,
                          torch.float32 if last else None)
  return x

def branch_a():
  outs = []
  if A == 'dc':
    x = x_c
    for _ in range(6):
      x, _ = ops.dc_raw(x, k0, m8, None); outs.append(x)
  elif A == 'convdc':
    for _ in range(3):
      y = convs(x_pad, block.num_convs); outs.append(y)
      x, _ = ops.dc_raw(y, k0, m8, None); outs.append(x)
  elif A == 'lastconvdc':
    h = convs(x_pad, block.num_convs - 1); outs.append(h)
    for _ in range(4):
      cp = block.layers[str(3 * (block.num_convs - 1) + 1)]
      y = ops.ConvAct.apply(h, None, cp.weight, cp.bias, cp.layer, 1.0, torch.float32); outs.append(y)
      x, _ = ops.dc_raw(y, k0, m8, None); outs.append(x)
  return outs

def branch_b():
  if B == 'conv':
    x = x_pad
    for _ in range(4):
      x = convs(x_pad, block.num_convs - 1)
    return x
  if B == 'mm':
    y = mm_a
    for _ in range(30):
      y = torch.mm(y, mm_b)
    return y
  if B == 'conv1':
    cp = block.layers['1']
    for _ in range(12):
      x = ops.ConvAct.apply(x_pad, None, cp.weight, cp.bias, cp.layer, block.slope, None)
      BPTR.append((x.data_ptr(), x.numel() * x.element_size()))
    return x
  if B == 'myelem':
    for _ in range(60):
      y = ops.nchw_to_nhwc(big_img, torch.bfloat16, 8)
    return y
  if B == 'elem':
    y = big
    for _ in range(40):
      y = y * 1.0001 + 0.1
    return y
  return None

with torch.no_grad():
  ref = [t.clone() for t in branch_a()]
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
BPTR = []
PLACE = os.environ.get('PLACE', 'side_side')
def both():
  cur = torch.cuda.current_stream()
  sa = cur if PLACE.split('_')[0] == 'main' else s1
  sb = cur if PLACE.split('_')[1] == 'main' else s2
  s1.wait_stream(cur); s2.wait_stream(cur)
  with torch.no_grad():
    if os.environ.get('ORDER', 'ab') == 'ab':
      with torch.cuda.stream(sa):
        a = branch_a()
      with torch.cuda.stream(sb):
        b = branch_b()
    else:
      with torch.cuda.stream(sb):
        b = branch_b()
      with torch.cuda.stream(sa):
        a = branch_a()
  cur.wait_stream(s1); cur.wait_stream(s2)
  return a, b
for _ in range(3):
  both()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode='thread_local'):
  a, b = both()
del BPTR[:-12]
ar = [(t.data_ptr(), t.numel() * t.element_size()) for t in a]
print('A ranges', [(hex(p), n >> 20) for p, n in ar])
print('B ranges', sorted(set((hex(p), n >> 20) for p, n in BPTR)))
print('static: x_c', hex(x_c.data_ptr()), 'k0', hex(k0.data_ptr()), 'm8', hex(m8.data_ptr()), 'x_pad', hex(x_pad.data_ptr()))
for p, n in ar:
  for q, m in BPTR:
    if p < q + m and q < p + n:
      print('OVERLAP', hex(p), hex(q))
bad = [0] * len(ref)
for it in range(30):
  g.replay(); torch.cuda.synchronize()
  for i, (u, v) in enumerate(zip(a, ref)):
    bad[i] += int((u != v).any())
if os.environ.get('DETAIL'):
  for i, (u, v) in enumerate(zip(a, ref)):
    badm = (u != v)
    if badm.any():
      rows = badm.any(dim=3).any(dim=2)        # [B,H]
      cols = badm.any(dim=3).any(dim=1)        # [B,W]
      print(' out', i, 'frac bad', float(badm.float().mean()), 'maxabs', float((u - v).abs().max()),
            'bad rows per image', rows.sum(1).tolist(), 'bad cols per image', cols.sum(1).tolist(),
            'nan', int(torch.isnan(u).sum()))
      if i == 1 or i == 0:
        rr = rows[0].nonzero().flatten().tolist(); cc = cols[0].nonzero().flatten().tolist()
        print('   image0 bad rows', rr[:40], '... cols', cc[:40])
if os.environ.get('SPEC') and A == 'dc':
  def K(t):
    return torch.fft.fft2(torch.view_as_complex(t.contiguous()), norm='ortho')
  msk = m8.bool()
  kk0 = torch.view_as_complex(k0.contiguous())
  for i in (0, 1, 2):
    kb, kr = K(a[i]), K(ref[i])
    kin = K(x_c if i == 0 else ref[i - 1])
    kin_bad = K(x_c if i == 0 else a[i - 1])
    print(' out', i, 'sampled: |Kbad-k0-0| max', float((kb - kk0)[msk].abs().max()), '|Kref-k0| max', float((kr - kk0)[msk].abs().max()),
          ' unsampled: |Kbad-Kin_ref| max', float((kb - kin)[~msk].abs().max()), '|Kbad-Kin_bad|', float((kb - kin_bad)[~msk].abs().max()),
          '|Kref-Kin| max', float((kr - kin)[~msk].abs().max()), ' |Kbad| sampled max', float(kb[msk].abs().max()), 'k0 max', float(kk0.abs().max()))
if os.environ.get('SPEC') and A == 'dc':
  e = torch.view_as_complex((a[1] - ref[1]).contiguous())          # [B,H,W]
  def desc(name, t):
    m = t.abs()
    thr = m.max() * 1e-3
    nz = m > thr
    rows = nz.any(dim=2).sum(1).tolist(); cols = nz.any(dim=1).sum(1).tolist()
    print('  domain', name, 'max', float(m.max()), 'frac>1e-3max', float(nz.float().mean()), 'rows', rows, 'cols', cols)
    if name == 'hyb_y_kx' or name == 'hyb_ky_x':
      print('     image0 rows', nz[0].any(dim=1).nonzero().flatten().tolist()[:64])
      print('     image0 cols', nz[0].any(dim=0).nonzero().flatten().tolist()[:64])
  desc('image', e)
  desc('hyb_y_kx', torch.fft.fft(e, dim=2, norm='ortho'))
  desc('hyb_ky_x', torch.fft.fft(e, dim=1, norm='ortho'))
  desc('kspace', torch.fft.fft2(e, norm='ortho'))
print(PLACE, os.environ.get('ORDER', 'ab'), 'A=%s B=%s mismatching replays per output of A: %s' % (A, B, bad))
