#!/bin/bash
# usage (GPU box): tools/pmc_bench.sh  -> gpurun_out/pmc_bench_traffic.json
# Two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over the bench command; per-kernel
# averages per dispatch.  FETCH_SIZE is doubled for gfx950 (MI355X_MICROARCH.md, HBM section).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcb_$c -o r01 -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-other-configs --no-input-ab --settle-s 0 --steps 6 --warmup 2 "$@" > $R/gpurun_out/pmcb_$c.log 2>&1
done
python3 - $R <<'PY'
import csv, glob, sys, json, collections
R = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
  for fn in glob.glob(R + '/gpurun_out/pmcb_%s/**/*counter_collection.csv' % c, recursive=True):
    for r in csv.DictReader(open(fn)):
      agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, d in agg.items():
  if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
    f = sum(d['FETCH_SIZE']) / len(d['FETCH_SIZE']) * 1024.0 * 2.0      # KiB -> B, gfx950 x2
    w = sum(d['WRITE_SIZE']) / len(d['WRITE_SIZE']) * 1024.0
    out[k] = {'fetch_bytes_per_launch': f, 'write_bytes_per_launch': w, 'launches': len(d['FETCH_SIZE'])}
json.dump(out, open(R + '/gpurun_out/pmc_bench_traffic.json', 'w'), indent=1)
PY
rm -rf $R/gpurun_out/pmcb_FETCH_SIZE $R/gpurun_out/pmcb_WRITE_SIZE
