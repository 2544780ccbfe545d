#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "small_ops or loss or feature or multi" 2>&1 | tail -3
bash tools/ab_old_new.sh prev
