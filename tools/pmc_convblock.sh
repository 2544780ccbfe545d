#!/bin/bash
# usage (GPU box): tools/pmc_convblock.sh -> instruction mix / busy counters of the fused conv-block kernel
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM"; do
  rm -rf $R/gpurun_out/pmccb
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmccb -o cb -- python3 $R/tools/bench_convblock.py > /dev/null 2>&1
  python3 - $R <<'PY'
import csv, glob, sys, collections
R = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in glob.glob(R + '/gpurun_out/pmccb/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(fn)):
    if 'convblock' in r['Kernel_Name'] and 'Lb0' not in r['Kernel_Name']:
      pass
    if 'convblock' in r['Kernel_Name']:
      agg[r['Kernel_Name'][:60]][r['Counter_Name']] += float(r['Counter_Value'])
      n[(r['Kernel_Name'][:60], r['Counter_Name'])] += 1
for k, d in agg.items():
  print(k)
  for c, v in sorted(d.items()):
    print('   %-28s %16.0f  per launch %14.0f' % (c, v, v / n[(k, c)]))
PY
done
rm -rf $R/gpurun_out/pmccb
