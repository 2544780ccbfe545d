"""One steady-state step of a rocprofv3 kernel trace (csv) as a timeline: per kernel start / duration / queue, relative
to the step start, so that exposed tails and gaps are visible.
usage: python tools/trace_timeline.py trace.csv > timeline.txt"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '')[:int(__import__('os').environ.get('NAMELEN', '52'))].split('(')[0] if int(__import__('os').environ.get('NAMELEN', '52')) <= 52 else r['Kernel_Name'].replace('void ', '')[:int(__import__('os').environ.get('NAMELEN', '52'))], r['Queue_Id']) for r in rows]
ev.sort()
adam = [i for i, e in enumerate(ev) if 'adam_dev' in e[2]]
lo, hi = adam[-3], adam[-1]          # from the end of the previous step's generator Adam to this step's
win = ev[lo + 1:hi + 1]
t0 = win[0][0]
qs = sorted(set(e[3] for e in win))
print('step: %d kernels, %.3f ms; queues %s' % (len(win), (max(e[1] for e in win) - t0) / 1e6, qs))
prev_end = {}
for s, e, n, q in win:
  col = qs.index(q)
  gap = (s - prev_end.get(q, s)) / 1e3
  print('%9.1f us  %7.1f us  %s%-52s' % ((s - t0) / 1e3, (e - s) / 1e3, '    ' * col, n))
  prev_end[q] = e
