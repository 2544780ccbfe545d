#!/bin/bash
# same-box A/B of runner-level switches (tools/bench_toggle.py CSMRI_RUNNER=...), three alternating rounds
# usage: tools/ab_runner.sh "name1:k=v,k=v" "name2:k=v" ...   (a baseline run "base" is added)
cd $GRAFT_REPO_ROOT
O=gpurun_out/abr; mkdir -p $O; rm -f $O/*.json
for i in 1 2 3; do
  for spec in "base:" "$@"; do
    name=${spec%%:*}; kv=${spec#*:}
    CSMRI_RUNNER=$kv python tools/bench_toggle.py --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null > $O/${name}_$i.json
  done
done
python - <<'PY'
import json, glob, os
rows = {}
for f in sorted(glob.glob('gpurun_out/abr/*.json')):
  k = os.path.basename(f).rsplit('_', 1)[0]
  r = json.load(open(f))
  rows.setdefault(k, []).append((r['value'], r['input_ab']['resident'], r['ms_per_step']))
for k, v in rows.items():
  print('%-30s %s' % (k, v))
PY
