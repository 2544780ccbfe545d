#!/bin/bash
# usage (GPU box): tools/pmc_insts.sh -> per-kernel instruction mix (VALU / MFMA / LDS / SALU per launch)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmci -o r01 -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-other-configs --settle-s 0 --no-overlap --steps 4 --warmup 2 > $R/gpurun_out/pmci.log 2>&1
python3 - $R <<'PY'
import csv, glob, sys, collections
R = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in glob.glob(R + '/gpurun_out/pmci/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(fn)):
    agg[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_INSTS_MFMA', 0)):
  m = d.get('SQ_INSTS_MFMA', 0)
  if m <= 0: continue
  print('%-66s valu/mfma %.2f lds/mfma %.2f salu/mfma %.2f wait_frac %.2f' % (
      k[:66], d.get('SQ_INSTS_VALU', 0) / m, d.get('SQ_INSTS_LDS', 0) / m, d.get('SQ_INSTS_SALU', 0) / m,
      d.get('SQ_WAIT_INST_ANY', 0) / max(d.get('SQ_WAVE_CYCLES', 1), 1)))
PY
rm -rf $R/gpurun_out/pmci
