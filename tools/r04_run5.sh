#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04/weights_split
python -m pytest tests/test_hip_ops.py tests/test_hip_path.py tests/test_checkpoints_gpu.py tests/test_trajectory_gpu.py -m gpu -q -k "split or convblock or recnet or dc or c2_ or smoke or checkpoint or trajectory or f3 or F3 or layout" > gpurun_out/r04/gputest_6.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/gputest_6.log
python tools/trajectory.py --config c2 --steps 2400 --dtypes fp32,bf16 --ensemble 2 --variant "split-bf16 images between cascades" --save-weights gpurun_out/r04/weights_split --out gpurun_out/r04/traj_c2_2400_split.json > gpurun_out/r04/traj_c2_2400_split.log 2>&1
python tools/trajectory.py --config c2 --steps 600 --dtypes fp32,bf16 --ensemble 4 --variant "split-bf16 images between cascades" --out gpurun_out/r04/traj_c2_600_split.json > gpurun_out/r04/traj_c2_600_split.log 2>&1
python bench.py --config c2 --no-cpu-baseline --no-roofline --no-other-configs > gpurun_out/r04/bench_c2_split.json 2>/dev/null
tail -4 gpurun_out/r04/gputest_6.log
python - <<'PY'
import json
for f in ('traj_c2_2400_split','traj_c2_600_split'):
    d=json.load(open('gpurun_out/r04/%s.json'%f)); print(f, json.dumps(d['summary']['ensemble'].get('bf16_minus_fp32')))
    for k,r in d['runs'].items(): print('  ',k, round(r['final_psnr_heldout_eval'],4), r.get('final_psnr_heldout_eval_fp32_compute_of_these_weights'))
d=json.loads(open('gpurun_out/r04/bench_c2_split.json').read().strip().splitlines()[-1]); print('c2', d['value'], d['ms_per_step'], d['input_ab'])
PY
