#!/bin/bash
# full GPU test suite (the dispatch-coverage test needs the committed bench line: run after bench), bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_full2; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json | head -c 300; echo
cp $O/bench.json profiles/r05_bench_n1.json
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; tail -5 $O/tests.log
