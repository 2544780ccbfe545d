#!/bin/bash
# same-box A/B of several builds of the tree: ab/<name> (git worktrees of earlier commits, library built in each) against
# the working tree ("new"), alternating, so that box-to-box clock differences (+-3 % on this pool) cancel; then one
# kernel-trace summary per build (per-kernel totals over 40 replayed steps)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab; mkdir -p $O
VARIANTS="$@"
rm -f $O/*_[0-9].json
for i in 1 2 3; do
  for v in $VARIANTS new; do
    d=$R/ab/$v; [ $v = new ] && d=$R
    (cd $d && python bench.py --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null) > $O/${v}_$i.json
  done
done
[ -n "$NOPROF" ] || for v in $VARIANTS new; do
  d=$R/ab/$v; [ $v = new ] && d=$R
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/prof_$v -o r01 -- python3 $d/bench.py --no-cpu-baseline --no-roofline --steps 40 --settle-s 0 --no-other-configs --no-input-ab > $O/prof_$v.log 2>&1
   python3 $R/tools/rocpd_stats.py $O/prof_$v/r01_results.db $O/prof_$v.csv; rm -rf $O/prof_$v)
done
python - $VARIANTS new <<'PY'
import json, glob, sys
for k in sys.argv[1:]:
  rows = [json.load(open(f)) for f in sorted(glob.glob('gpurun_out/ab/%s_*.json' % k))]
  print(k, [(r['value'], r['input_ab']['resident'], r['ms_per_step']) for r in rows])
PY
