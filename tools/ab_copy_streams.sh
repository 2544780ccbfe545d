#!/bin/bash
# C3 host-batch leg by number of copy streams, alternating on one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/abc; mkdir -p $O; rm -f $O/*.json
for i in 1 2 3; do for n in 1 2 3; do
  python bench.py --no-cpu-baseline --no-other-configs --no-roofline --copy-streams $n 2>/dev/null > $O/cs${n}_$i.json
done; done
python - <<'PY'
import json, glob, os
rows = {}
for f in sorted(glob.glob('gpurun_out/abc/*.json')):
  k = os.path.basename(f).rsplit('_', 1)[0]
  r = json.load(open(f))
  rows.setdefault(k, []).append((r['value'], r['input_ab']['resident'], r['ms_per_step']))
for k, v in rows.items():
  print('%-10s %s' % (k, v))
PY
