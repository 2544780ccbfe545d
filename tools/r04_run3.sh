#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -q > gpurun_out/r04/gputest_3.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04/gputest_3.log
bash tools/prof.sh r04a --steps 40 --settle-s 0 --no-other-configs
bash tools/prof.sh r04a_ss --steps 40 --settle-s 0 --no-other-configs --no-overlap --no-graphs
bash tools/timeline.sh
mv gpurun_out/timeline.txt gpurun_out/r04/timeline_a.txt; mv gpurun_out/concurrency.txt gpurun_out/r04/concurrency_a.txt; mv gpurun_out/critical_path.txt gpurun_out/r04/critical_path_a.txt
tail -6 gpurun_out/r04/gputest_3.log
