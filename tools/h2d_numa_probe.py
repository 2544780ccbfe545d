"""H2D rate of a 134 MB pinned batch (C2: 64 slices x 4 tensors) by the NUMA node the pinned pages were allocated on."""
import glob, os, time, torch
dev = torch.device('cuda', 0)
props = torch.cuda.get_device_properties(dev)
bdf = '%04x:%02x:%02x.0' % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
try:
  gnode = int(open('/sys/bus/pci/devices/%s/numa_node' % bdf).read())
except Exception as e:
  gnode = repr(e)
print('gpu', bdf, 'numa_node', gnode, 'cpus', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
nodes = sorted(int(p.rsplit('node', 1)[1]) for p in glob.glob('/sys/devices/system/node/node[0-9]*'))
n = 33554432 // 4
dst = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(4)]
st = torch.cuda.Stream()
old = os.sched_getaffinity(0)
for node in [None] + nodes:
  if node is not None:
    cpus = set()
    for part in open('/sys/devices/system/node/node%d/cpulist' % node).read().strip().split(','):
      a, _, b = part.partition('-')
      cpus.update(range(int(a), int(b or a) + 1))
    if not (cpus & old):
      continue
    os.sched_setaffinity(0, cpus & old)
  host = [torch.randn(n).pin_memory() for _ in range(4)]
  os.sched_setaffinity(0, old)
  def copy():
    with torch.cuda.stream(st):
      for d, h in zip(dst, host):
        d.copy_(h, non_blocking=True)
  copy(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(10):
    copy()
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 10
  print('pinned on node %s: 134 MB in %.2f ms = %.1f GB/s' % (node, dt * 1e3, 4 * n * 4 / dt / 1e9))
  del host
