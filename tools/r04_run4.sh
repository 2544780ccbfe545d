#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -m pytest tests/test_hip_path.py tests/test_distributed_gpu.py tests/test_bench_shapes.py -m gpu -q -x -k "disc_phase_grads or three_group or c2_recnet5 or fp8 or nccl or bench_layer or dispatch" > gpurun_out/r04/gputest_4.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/gputest_4.log
python tools/stamp_convblock_bwd.py > gpurun_out/r04/stamp_convblock_bwd.log 2>&1
B="--no-cpu-baseline --no-roofline --no-other-configs --no-input-ab --steps 300"
for rep in 1 2; do
python bench.py $B > gpurun_out/r04/ab_default_$rep.json 2>/dev/null
python bench.py $B --lookahead-last > gpurun_out/r04/ab_lookahead_last_$rep.json 2>/dev/null
python bench.py $B --finish-multi 1 > gpurun_out/r04/ab_finish_multi_$rep.json 2>/dev/null
python bench.py $B --finish-multi 1 --lookahead-last > gpurun_out/r04/ab_both_$rep.json 2>/dev/null
done
tail -5 gpurun_out/r04/gputest_4.log; cat gpurun_out/r04/stamp_convblock_bwd.log
for f in gpurun_out/r04/ab_*.json; do echo $f $(python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"); done
