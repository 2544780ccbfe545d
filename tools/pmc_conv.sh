#!/bin/bash
# usage (GPU box): tools/pmc_conv.sh <tag> "<counters>" <bench_conv args...> -> gpurun_out/pmc_<tag>.txt
tag=$1; ctrs=$2; shift; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -o r01 -- python3 $R/tools/bench_conv.py "$@" > $R/gpurun_out/pmc_$tag.log 2>&1
python3 - $R/gpurun_out/pmc_$tag $R/gpurun_out/pmc_$tag.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in f:
  for r in csv.DictReader(open(fn)):
    k = r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[(k, r['Counter_Name'])] += 1
with open(sys.argv[2], 'w') as o:
  for k, d in agg.items():
    o.write(k + '\n')
    for c, v in d.items():
      o.write('   %-28s %14.0f per-dispatch (n=%d)\n' % (c, v / cnt[(k, c)], cnt[(k, c)]))
PY
rm -rf $R/gpurun_out/pmc_$tag
