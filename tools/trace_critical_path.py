"""Critical chain of one replayed step from a rocprofv3 kernel trace (csv): walk back from the step's last kernel, at every
kernel taking as its predecessor the kernel (on any queue) that ENDED last before it started -- the dependency that released
it (or, on the same queue, the kernel in front of it).  Prints the chain's composition by kernel name, the launch gaps on it,
and how much kernel time runs off the chain.
usage: python tools/trace_critical_path.py trace.csv"""
import bisect, collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '').split('(')[0][:56]) for r in rows)
adam = [i for i, e in enumerate(ev) if 'adam_dev' in e[2]]
lo, hi = adam[-3], adam[-1]                  # one step: behind the previous generator Adam .. this step's generator Adam
win = ev[lo + 1:hi + 1]
t0 = win[0][0]
by_end = sorted(win, key=lambda e: e[1])
ends = [e[1] for e in by_end]
cur = max(win, key=lambda e: e[1])
chain, gaps = [], 0
while True:
  chain.append(cur)
  i = bisect.bisect_right(ends, cur[0] + 200) - 1          # ended before (or within 0.2 us of) this start
  while i >= 0 and (by_end[i] is cur or by_end[i][0] >= cur[0]):
    i -= 1
  if i < 0:
    break
  pred = by_end[i]
  gaps += max(0, cur[0] - pred[1])
  cur = pred
chain.reverse()
tot = sum(e[1] - e[0] for e in chain)
wall = max(e[1] for e in win) - t0
allk = sum(e[1] - e[0] for e in win)
print('step %.3f ms wall, %d kernels, %.3f ms of kernel time' % (wall / 1e6, len(win), allk / 1e6))
print('critical chain: %d kernels, %.3f ms of kernel time + %.3f ms of gaps; off the chain: %d kernels, %.3f ms' %
      (len(chain), tot / 1e6, gaps / 1e6, len(win) - len(chain), (allk - tot) / 1e6))
names = collections.defaultdict(lambda: [0, 0])
for s, e, n in chain:
  names[n][0] += 1; names[n][1] += e - s
for n, (k, t) in sorted(names.items(), key=lambda kv: -kv[1][1])[:28]:
  print('  %-58s %3d %8.1f us' % (n, k, t / 1e3))
onchain = set(id(e) for e in chain)
off = collections.defaultdict(lambda: [0, 0])
for e in win:
  if id(e) not in onchain:
    off[e[2]][0] += 1; off[e[2]][1] += e[1] - e[0]
print('largest off-chain:')
for n, (k, t) in sorted(off.items(), key=lambda kv: -kv[1][1])[:12]:
  print('  %-58s %3d %8.1f us' % (n, k, t / 1e3))
