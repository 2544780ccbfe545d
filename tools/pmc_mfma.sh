#!/bin/bash
# usage (GPU box): tools/pmc_mfma.sh  -> gpurun_out/pmc_mfma_util.json
# One rocprofv3 --pmc pass over the bench command: MFMA-pipe busy cycles per kernel against the
# kernel's own duration.  SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs and
# GRBM_GUI_ACTIVE over its 8 XCDs (MI355X_MICROARCH.md), so
#   utilisation = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmcm -o r01 -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-other-configs --no-input-ab --settle-s 0 --no-overlap --steps 6 --warmup 2 "$@" > $R/gpurun_out/pmcm.log 2>&1
python3 - $R <<'PY'
import csv, glob, sys, json, collections
R = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(R + '/gpurun_out/pmcm/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(fn)):
    agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, d in agg.items():
  if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and 'GRBM_GUI_ACTIVE' in d and sum(d['SQ_VALU_MFMA_BUSY_CYCLES']) > 0:
    busy, act = sum(d['SQ_VALU_MFMA_BUSY_CYCLES']), sum(d['GRBM_GUI_ACTIVE'])
    out[k] = {'launches': len(d['GRBM_GUI_ACTIVE']), 'mfma_busy_cycles_per_launch': busy / len(d['GRBM_GUI_ACTIVE']),
              'gui_active_per_launch': act / len(d['GRBM_GUI_ACTIVE']),
              'mfma_utilisation': busy / (act / 8.0 * 1024.0)}
json.dump(out, open(R + '/gpurun_out/pmc_mfma_util.json', 'w'), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]['mfma_busy_cycles_per_launch'] * kv[1]['launches'])[:14]:
  print('%-62s n=%4d util %.3f' % (k[:62], v['launches'], v['mfma_utilisation']))
PY
rm -rf $R/gpurun_out/pmcm
