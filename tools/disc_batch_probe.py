"""Timing probe: two D passes at B=8 vs one at B=16 (fwd+bwd, hipGraph replay)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
import bench

runner, conf = bench.build_runner('bf16', 8)
disc = runner.disc
disc.train()
dev = torch.device('cuda', 0)
x8a = torch.randn(8, 256, 256, 8, device=dev).bfloat16()
x8b = torch.randn(8, 256, 256, 8, device=dev).bfloat16()
x16 = torch.cat([x8a, x8b], 0)


def two():
  for x in (x8a, x8b):
    o = disc(nhwc=x)
    o['logits'].sum().backward()


def one():
  o = disc(nhwc=x16)
  o['logits'].sum().backward()


def fwd_only():
  with torch.no_grad():
    disc(nhwc=x8a)


for fn in (two, one, fwd_only):
  s = torch.cuda.Stream()
  s.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(s):
    for _ in range(3):
      fn()
  torch.cuda.current_stream().wait_stream(s)
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    fn()
  for _ in range(3):
    g.replay()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20):
    g.replay()
  e1.record(); torch.cuda.synchronize()
  print(fn.__name__, '%.3f ms' % (e0.elapsed_time(e1) / 20))
