"""Concurrency analysis of a rocprofv3 kernel trace (csv): for the last few steady-state steps, how much wall time
has 0 / 1 / >= 2 kernels resident, and which kernels run ALONE (the serial fraction of the step).
usage: python tools/trace_concurrency.py trace.csv [n_steps_in_window]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:60], r['Queue_Id']) for r in rows]
ev.sort()
# steady state: take the last third of the trace, bounded by two occurrences of the generator Adam kernel
adam = [i for i, e in enumerate(ev) if 'adam_dev' in e[2]]
nwin = int(sys.argv[2]) if len(sys.argv) > 2 else 4
# two adam launches per step (D, G): window = last nwin steps
lo, hi = adam[-2 * nwin - 1], adam[-1]
win = ev[lo + 1:hi + 1]
t0, t1 = win[0][0], max(e[1] for e in win)
print('window: %d kernels, %.3f ms, %.3f ms per step' % (len(win), (t1 - t0) / 1e6, (t1 - t0) / 1e6 / nwin))
pts = []
for s, e, n, q in win:
  pts.append((s, 1, n)); pts.append((e, -1, n))
pts.sort()
active = collections.Counter(); cur = 0; last = t0
hist = collections.Counter(); alone = collections.Counter()
for t, d, n in pts:
  dt = t - last
  if dt > 0:
    hist[min(cur, 3)] += dt
    if cur == 1:
      alone[next(iter(k for k, v in active.items() if v > 0))] += dt
  last = t
  active[n] += d; cur += d
tot = float(t1 - t0)
for k in sorted(hist):
  print('  %s kernels resident: %6.3f ms/step  %5.1f %%' % ('>=3' if k == 3 else k, hist[k] / 1e6 / nwin, 100 * hist[k] / tot))
print('kernels running ALONE (ms/step):')
for n, v in alone.most_common(22):
  print('  %-60s %.3f' % (n, v / 1e6 / nwin))
busy = collections.Counter()
for s, e, n, q in win:
  busy[n] += e - s
print('kernel time by name (ms/step):')
for n, v in busy.most_common(14):
  print('  %-60s %.3f' % (n, v / 1e6 / nwin))
