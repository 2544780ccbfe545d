#!/bin/bash
# the round's closing pass on ONE box: full gpu suite, smoke, bench lines (default / driver-style / C2), rocprofv3 kernel
# stats (replayed step, single stream, C2), PMC traffic (C3, C2), MFMA utilisation, step timeline, same-box A/B against
# the round-5 build (ab/r05 if present)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06final; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc $?" >> $O/gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
# (the PMC passes first: bench.py quotes `roofline.traffic` from profiles/r06_pmc_bench_traffic*.json, which must come from
#  THIS build -- kernel instance names are the keys)
bash tools/pmc_bench.sh; cp gpurun_out/pmc_bench_traffic.json profiles/r06_pmc_bench_traffic.json; mv gpurun_out/pmc_bench_traffic.json $O/pmc_bench_traffic.json
bash tools/pmc_bench.sh --config c2; cp gpurun_out/pmc_bench_traffic.json profiles/r06_pmc_bench_traffic_c2.json; mv gpurun_out/pmc_bench_traffic.json $O/pmc_bench_traffic_c2.json
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1_20steps.json 2> $O/bench_n1_20steps.err ) 2> $O/bench_n1_20steps.time
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --config c2 --no-other-configs > $O/bench_c2.json 2>/dev/null
bash tools/prof.sh r06 --steps 40 --settle-s 0 --no-other-configs --no-input-ab; mv gpurun_out/prof_r06.csv $O/bench_n1_kernel_stats.csv
bash tools/prof.sh r06ss --steps 40 --settle-s 0 --no-other-configs --no-input-ab --no-overlap --no-graphs; mv gpurun_out/prof_r06ss.csv $O/bench_n1_kernel_stats_single_stream.csv
bash tools/prof.sh r06c2 --steps 40 --settle-s 0 --no-other-configs --no-input-ab --config c2; mv gpurun_out/prof_r06c2.csv $O/bench_c2_kernel_stats.csv
bash tools/pmc_mfma.sh > $O/pmc_mfma_util.txt 2>&1; mv gpurun_out/pmc_mfma_util.json $O/pmc_mfma_util.json
bash tools/timeline.sh; mv gpurun_out/timeline.txt $O/c3_step_timeline.txt; mv gpurun_out/concurrency.txt $O/c3_step_concurrency.txt; mv gpurun_out/critical_path.txt $O/c3_step_critical_chain.txt
[ -d ab/r05 ] && NOPROF=1 bash tools/ab_old_new.sh r05 > $O/same_box_ab.txt 2>&1
tail -3 $O/gputest.log; tail -2 $O/smoke.log; cat $O/bench_n1_20steps.time; tail -2 $O/same_box_ab.txt
