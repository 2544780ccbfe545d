"""Does a pinned H2D copy on a side stream overlap with compute on the main stream on this box?"""
import time, torch
dev = torch.device('cuda', 0)
n = 134 * 1024 * 1024 // 4
host = torch.empty(n, dtype=torch.float32).pin_memory()
dst = torch.empty(n, dtype=torch.float32, device=dev)
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
copy_stream = torch.cuda.Stream()

def t(fn, reps=10):
  fn(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps):
    fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / reps * 1e3

def compute():
  for _ in range(6):
    torch.mm(a, a)
def copy():
  with torch.cuda.stream(copy_stream):
    dst.copy_(host, non_blocking=True)
def both():
  copy(); compute()
  torch.cuda.current_stream().wait_stream(copy_stream)
print('compute alone %.2f ms   copy alone %.2f ms (%.1f GB/s)   both %.2f ms' % (
    t(compute), t(lambda: (copy(), torch.cuda.current_stream().wait_stream(copy_stream))),
    0.134 * 1.048576 / t(lambda: (copy(), torch.cuda.current_stream().wait_stream(copy_stream))) * 1e3, t(both)))
import os
print('HSA_ENABLE_SDMA', os.environ.get('HSA_ENABLE_SDMA'))
