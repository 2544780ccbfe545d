#!/bin/bash
# full GPU test suite, then same-box A/B against ab/r04
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_full; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; tail -5 $O/tests.log
NOPROF=1 bash tools/ab_old_new.sh r04 2>&1 | tail -3
