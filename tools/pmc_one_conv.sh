#!/bin/bash
# usage (GPU box): tools/pmc_one_conv.sh <bench_conv case> <mode> -> memory-path counters of one conv launch shape
R=$GRAFT_REPO_ROOT
case_=$1; mode=$2
cd /tmp && export TMPDIR=/tmp
# (only the L2 set: the TA_* / TCP_* sets did not finish within 10 minutes on this pool)
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"; do
  rm -rf $R/gpurun_out/pmc1
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc1 -o c -- python3 $R/tools/bench_conv.py $case_ $mode > /dev/null 2>&1
  python3 - $R <<'PY'
import csv, glob, sys, collections
R = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in glob.glob(R + '/gpurun_out/pmc1/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(fn)):
    if ('conv' in r['Kernel_Name'] or 'gpipe' in r['Kernel_Name'] or 'wgrad' in r['Kernel_Name'] or 'wpatch' in r['Kernel_Name']) and 'reduce' not in r['Kernel_Name'] and 'scatter' not in r['Kernel_Name']:
      agg[r['Kernel_Name'][:48]][r['Counter_Name']] += float(r['Counter_Value']); n[(r['Kernel_Name'][:48], r['Counter_Name'])] += 1
for k, d in agg.items():
  for c, v in sorted(d.items()):
    print('%-48s %-40s per launch %16.0f' % (k, c, v / n[(k, c)]))
PY
done
rm -rf $R/gpurun_out/pmc1
