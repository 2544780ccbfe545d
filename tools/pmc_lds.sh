#!/bin/bash
# usage (GPU box): tools/pmc_lds.sh -> per-kernel VALU/MFMA/LDS instruction mix and LDS bank-conflict share over the bench step
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  rm -rf $R/gpurun_out/pmcl_$((++i))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcl_$i -o l -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-other-configs --no-input-ab --settle-s 0 --no-overlap --steps 4 --warmup 2 > /dev/null 2>&1
done
python3 - $R <<'PY'
import csv, glob, sys, collections
R = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in glob.glob(R + '/gpurun_out/pmcl_*/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(fn)):
    agg[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value'])
    n[(r['Kernel_Name'], r['Counter_Name'])] += 1
rows = []
for k, d in agg.items():
  m = d.get('SQ_INSTS_MFMA', 0)
  if m <= 0: continue
  act = d.get('GRBM_GUI_ACTIVE', 0)
  rows.append((d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0), k, d.get('SQ_INSTS_VALU', 0) / m, d.get('SQ_INSTS_LDS', 0) / m,
               d.get('SQ_LDS_BANK_CONFLICT', 0) / max(d.get('SQ_LDS_IDX_ACTIVE', 1), 1),
               d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(act / 8.0 * 1024.0, 1), n[(k, 'GRBM_GUI_ACTIVE')]))
for busy, k, vm, lm, cf, util, cnt in sorted(rows, reverse=True):
  print('%-70s n=%4d valu/mfma %5.2f lds/mfma %5.2f lds-conflict %4.2f mfma-util %5.3f' % (k[:70], cnt, vm, lm, cf, util))
PY
rm -rf $R/gpurun_out/pmcl_*
