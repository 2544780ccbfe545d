#!/bin/bash
# same-box A/B of the working tree with single switches off (tools/bench_toggle.py), two alternating rounds
cd $GRAFT_REPO_ROOT
O=gpurun_out/abt; mkdir -p $O
for i in 1 2; do
  for off in NONE FANIN_TAPS FANIN_REFINE FANIN_LOGITS SEED_CONST PACK_BIAS BN_SMALL "FANIN_TAPS,FANIN_REFINE,FANIN_LOGITS,SEED_CONST,PACK_BIAS,BN_SMALL"; do
    o=$off; [ $off = NONE ] && o=""
    CSMRI_OFF=$o python tools/bench_toggle.py --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null > "$O/${off//,/+}_$i.json"
  done
  (cd ab/old && python bench.py --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null) > $O/OLD_$i.json
done
python - <<'PY'
import json, glob, os
rows = {}
for f in sorted(glob.glob('gpurun_out/abt/*.json')):
  k = os.path.basename(f).rsplit('_', 1)[0]
  r = json.load(open(f))
  rows.setdefault(k, []).append((r['ms_per_step'], r['input_ab']['resident']))
for k, v in rows.items():
  print('%-70s %s' % (k, v))
PY
