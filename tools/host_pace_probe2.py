"""Is the captured step host-paced?  Host time of graph.replay() vs GPU time of the graph, back-to-back replays vs
replays with a sync in between, and the runner's own train_epoch loop."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
import bench
from data.synthetic import synth_batch
runner, conf = bench.build_runner('c3', 'bf16', 8)
dev = torch.device('cuda', 0)
hb = [synth_batch(8, 256, 256, acc=4, seed=i) for i in range(2)]
runner.train_epoch(bench.PinnedHostLoader(hb, 3, dev, resident=True), 0)
runner.overlap_streams = True
b0 = {k: v.to(dev) for k, v in hb[0].items()}
runner.enable_graphs(b0)
runner.train_epoch(bench.PinnedHostLoader(hb, 5, dev, resident=True), 1)
torch.cuda.synchronize()
g = runner._graph['graphs'][0]
n = 30
# (a) back to back, host free-running
torch.cuda.synchronize(); t0 = time.perf_counter(); host = []
for _ in range(n):
  h0 = time.perf_counter(); g.replay(); host.append(time.perf_counter() - h0)
torch.cuda.synchronize(); ta = (time.perf_counter() - t0) / n
host.sort()
print('back-to-back replays: %.3f ms per step; host time inside replay(): median %.3f ms, max %.3f ms' % (ta * 1e3, host[n // 2] * 1e3, host[-1] * 1e3))
# (b) sync after every replay: the host cannot run ahead
t0 = time.perf_counter()
for _ in range(n):
  g.replay(); torch.cuda.synchronize()
tb = (time.perf_counter() - t0) / n
print('replay + synchronize each step: %.3f ms per step' % (tb * 1e3))
# (c) the training loop as bench.py runs it
t0 = time.perf_counter()
runner.train_epoch(bench.PinnedHostLoader(hb, n, dev, resident=True), 1, steps_per_train_summary=10 ** 9)
torch.cuda.synchronize()
print('train_epoch loop: %.3f ms per step' % ((time.perf_counter() - t0) / n * 1e3))
