"""How does a replayed hipGraph order a side-stream node against the main chain?  (rocprofv3 --kernel-trace around it)
main: A1 -> chain of N long kernels; side: B depends only on A1 (event), captured at different points of the chain.
Prints nothing itself: tools/graph_dep_probe.sh reads the kernel trace and reports when B started relative to the chain."""
import sys
import torch

N = 200                                            # chain of small kernels (a few us each): the GPU stays nearly empty
x = torch.zeros(1 << 12, device='cuda')
y = torch.zeros(1 << 12, device='cuda')
z = torch.zeros(16 << 20, device='cuda')           # B: one ~30 us kernel
side = torch.cuda.Stream()


def build(pos, op):
  """B captured after `pos` kernels of the chain (0 = right behind A1, before the chain)."""
  g = torch.cuda.CUDAGraph()
  s = torch.cuda.Stream()
  with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
      x.fill_(1.0)                                  # A1
      ev = torch.cuda.Event()
      ev.record(s)
      for i in range(N):
        if i == pos:
          side.wait_event(ev)
          with torch.cuda.stream(side):
            op(z)                                   # B
        y.add_(x)                                   # chain
      if pos >= N:
        side.wait_event(ev)
        with torch.cuda.stream(side):
          op(z)
      s.wait_stream(side)
  return g


variants = [(0, lambda t: t.mul_(1.5)), (2, lambda t: t.sub_(1.0)), (N, lambda t: t.div_(2.0)),
            (100, lambda t: t.clamp_(min=0.0))]
graphs = [build(p, op) for p, op in variants]
torch.cuda.synchronize()
for rep in range(4):
  for g in graphs:
    g.replay()
    torch.cuda.synchronize()
