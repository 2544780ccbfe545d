#!/bin/bash
# usage (GPU box): tools/pmc_any.sh <kernel-name substring> <python script + args ...> -> SQ / LDS counters per launch of the matching kernels
R=$GRAFT_REPO_ROOT
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmca_$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmca_$i -o a -- python3 "$@" > $R/gpurun_out/pmcalog_$i.txt 2>&1
done
python3 - $R "$pat" <<'PY'
import csv, glob, sys, collections
R, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in glob.glob(R + '/gpurun_out/pmca_*/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(fn)):
    if pat in r['Kernel_Name']:
      k = r['Kernel_Name'][:40] + ' grid ' + r.get('Grid_Size', '?')
      agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k, d in sorted(agg.items()):
  print(k)
  for c, v in sorted(d.items()):
    print('   %-32s per launch %16.0f   (n=%d)' % (c, v / n[(k, c)], n[(k, c)]))
PY
rm -rf $R/gpurun_out/pmca_[0-9]
