#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/gp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gp -o t -- python3 $R/tools/graph_dep_probe.py > /dev/null 2>&1
f=$(find $R/gpurun_out/gp -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $R/gpurun_out/graph_dep_probe.log
import csv, sys
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in csv.DictReader(open(sys.argv[1]))))
# split into replays at each fill kernel
groups, cur = [], None
for s, e, n, q in rows:
  if 'FillFunctor' in n:
    cur = []; groups.append(cur)
  if cur is not None:
    cur.append((s, e, n, q))
names = {'MulFunctor': 'B captured before the chain (first successor of A1)', 'sub': 'B captured after 2 chain kernels', 'Div': 'B captured after the whole chain', 'clamp': 'B captured after 10 chain kernels'}
for idx, g in enumerate(groups[-4:]):
  t0 = g[0][1]
  chain = [k for k in g if 'add' in k[2].lower() and 'Functor_add' in k[2] or 'CUDAFunctor_add' in k[2]]
  b = [k for k in g if k not in chain and k is not g[0]]
  if not chain or not b: continue
  tag = ['B right behind A1 (first successor)', 'B after 2 chain kernels', 'B after the whole chain', 'B after 100 chain kernels'][idx]
  print('%-58s chain %6.0f..%6.0f us (q%s, %d kernels)   B starts %7.1f us, ends %7.1f (q%s)' % (tag, (chain[0][0] - t0) / 1e3, (chain[-1][1] - t0) / 1e3, chain[0][3], len(chain), (b[0][0] - t0) / 1e3, (b[0][1] - t0) / 1e3, b[0][3]))
PY
rm -rf $R/gpurun_out/gp
