"""Pure-GPU step time: replay the captured step graph back to back (no host-side per-step work)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
import bench
from data.synthetic import synth_batch

runner, conf = bench.build_runner('bf16', 8)
dev = torch.device('cuda', 0)
batches = [{k: v.to(dev) for k, v in synth_batch(8, 256, 256, acc=4, seed=i).items()} for i in range(2)]
runner.overlap_streams = True
runner.enable_graphs(batches[0])
runner.train_epoch(bench.DeviceLoader(batches, 5), 1)
torch.cuda.synchronize()
g = runner._graph['graphs'][0]
for _ in range(3):
  g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
  g.replay()
torch.cuda.synchronize()
print('pure replay ms/step %.3f' % ((time.perf_counter() - t0) / 20 * 1e3))
t0 = time.perf_counter()
runner.train_epoch(bench.DeviceLoader(batches, 20), 1, steps_per_train_summary=10 ** 9)
torch.cuda.synchronize()
print('train_epoch ms/step %.3f' % ((time.perf_counter() - t0) / 20 * 1e3))
