"""Are torch's own elementwise kernels exact next to our MFMA conv kernels in one hipGraph?
(Follow-up of tools/pkf32_corun_probe.py: our library is built without packed-fp32 instructions,
torch's kernels are whatever the wheel ships.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
from csmri_hip import ops

g = torch.Generator().manual_seed(1)
b, size = 8, 256
wt = torch.randn(64, 64, 3, 3, generator=g) * 0.05
layer = ops.ConvLayer(torch.nn.Parameter(wt.cuda()), None, 1, (1, 1, 1, 1), 'zero', torch.bfloat16)
act = torch.randn(b, size, size, 64, generator=g).bfloat16().cuda()
xf = [torch.randn(b * size * size * 8, generator=g).cuda() for _ in range(3)]
xb = [t.bfloat16() for t in xf]

def torch_chain():
  outs = []
  t = xf[0]
  for i in range(6):
    t = t + xf[1 + i % 2]            # fp32 add (vectorized_elementwise_kernel<4, add<float>>)
    outs.append(t)
  u = xb[0]
  for i in range(6):
    u = u + xb[1 + i % 2]            # bf16 add
    outs.append(u)
  v = xf[0]
  for i in range(4):
    v = v * 1.0001 + xf[1]           # mul + add
    outs.append(v)
  outs.append(torch.stack([t.sum() for t in xf]))
  outs.append(torch.cat([xb[0][:4096], xb[1][:4096]]) * 0.5)
  return outs

def conv_chain():
  t = act
  for _ in range(10):
    t, _ = ops.conv_forward(layer, t, use_bias=False, act_slope=0.2)
  return t

with torch.no_grad():
  ref = [t.clone() for t in torch_chain()]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
  cur = torch.cuda.current_stream()
  s1.wait_stream(cur); s2.wait_stream(cur)
  with torch.no_grad():
    with torch.cuda.stream(s2):
      c = conv_chain()
    with torch.cuda.stream(s1):
      a = torch_chain()
  cur.wait_stream(s1); cur.wait_stream(s2)
  return a, c
for _ in range(2):
  both()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, capture_error_mode='thread_local'):
  a, c = both()
bad = [0] * len(ref)
for it in range(30):
  graph.replay(); torch.cuda.synchronize()
  for i, (u, v) in enumerate(zip(a, ref)):
    bad[i] += int(not torch.equal(u, v))
print('torch elementwise kernels next to MFMA convs: mismatching replays per output (30 replays):', bad)
