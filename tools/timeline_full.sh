#!/bin/bash
# usage (GPU box): tools/timeline_full.sh -> gpurun_out/timeline_full.txt (full kernel names) + gpurun_out/step_kernel_sums.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tl
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -o t -- python3 $R/bench.py --no-cpu-baseline --no-roofline --steps 12 --warmup 4 > /dev/null 2>&1
f=$(find $R/gpurun_out/tl -name "*kernel_trace.csv" | head -1)
NAMELEN=230 python3 $R/tools/trace_timeline.py "$f" > $R/gpurun_out/timeline_full.txt
python3 $R/tools/trace_timeline.py "$f" > $R/gpurun_out/timeline.txt
python3 $R/tools/trace_concurrency.py "$f" > $R/gpurun_out/concurrency.txt
python3 - "$f" > $R/gpurun_out/step_kernel_sums.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
adam = [i for i, e in enumerate(ev) if 'adam_dev' in e[2]]
lo, hi = adam[-3], adam[-1]
acc = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in ev[lo + 1:hi + 1]:
  acc[n][0] += 1; acc[n][1] += (e - s) / 1e3
tot = sum(v[1] for v in acc.values())
print('kernel time of one step: %.0f us in %d launches' % (tot, sum(v[0] for v in acc.values())))
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
  print('%8.1f us %4d  %s' % (t, c, n[:200]))
PY
rm -rf $R/gpurun_out/tl
