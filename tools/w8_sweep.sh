for v in 4 1 3 7; do echo "== CSMRI_GLDS_WAVES=$v"; CSMRI_GLDS_WAVES=$v python tools/bench_conv.py vgg4_2 vgg3_2 u128 disc3 disc5 vgg4_2b16 vgg5_2b16 vgg3_1b16 vgg2_1b16 fwd dgrad 2>&1 | grep -v amdgpu.ids; done
for v in 4 1 3 7; do CSMRI_GLDS_WAVES=$v python bench.py --steps 250 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench waves-mask $v', d['value'], d['ms_per_step'])"; done
