"""Is the host ahead of the GPU in the bench loop?  Host-side time of each step call (no sync)
next to the wall time per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'csmri-refinement_amd'))
import torch
import bench
from data.synthetic import synth_batch

runner, conf = bench.build_runner('bf16', 8)
batches = [{k: v.cuda() for k, v in synth_batch(8, 256, 256, acc=4, seed=conf.seed + 100000 * i).items()} for i in range(2)]
def request(loader, volatile=False):
  try:
    return next(runner.data_iter)
  except StopIteration:
    runner.data_iter = None
    return None
runner._request_data = request
runner._set_train()
runner.overlap_streams = True
runner.prefetch_pretrained = os.environ.get('PF', '1') == '1'
runner.enable_graphs(batches[0])
runner.train_epoch(bench.DeviceLoader(batches, 5), 1)
torch.cuda.synchronize()
runner._pf = None
runner.data_iter = iter(bench.DeviceLoader(batches, 31))
host = []
t_start = time.perf_counter()
for i in range(30):
  t0 = time.perf_counter()
  num, lm, data = runner._train_single_step(None)
  m = runner._compute_train_metrics(data)
  del data
  host.append((time.perf_counter() - t0) * 1e3)
t_issue = time.perf_counter() - t_start
torch.cuda.synchronize()
t_all = time.perf_counter() - t_start
print('host ms per step call:', [round(h, 2) for h in host])
print('all 30 steps issued after %.1f ms; GPU done after %.1f ms (%.3f ms/step)' % (t_issue * 1e3, t_all * 1e3, t_all * 1e3 / 30))
