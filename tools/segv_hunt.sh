#!/bin/bash
# Repeat the invocation that died with SIGSEGV once in round 5 (four legs in ONE process: c3, c3-fp32, c2, c5, c5-fp8)
# with a native backtrace handler preloaded (tools/segv_bt); keep stderr of every failing run.
# usage: tools/segv_hunt.sh TAG N [extra bench args]
cd $GRAFT_REPO_ROOT
TAG=${1:-a}; N=${2:-20}; shift; shift
O=gpurun_out/segv_hunt_$TAG; mkdir -p $O
export CSMRI_BENCH_DIAG=1
[ -f tools/segv_bt/segv_bt.so ] || gcc -O1 -g -fPIC -shared -o tools/segv_bt/segv_bt.so tools/segv_bt/segv_bt.c
fails=0
for i in $(seq 1 $N); do
  LD_PRELOAD=$PWD/tools/segv_bt/segv_bt.so python ${HUNT_BENCH:-bench.py} --steps 3 --warmup 2 --no-cpu-baseline --settle-s 0.2 --inprocess-legs "$@" > $O/out_$i.json 2> $O/err_$i.log
  rc=$?
  errs=$(grep -o '"error": "[^"]*"' $O/out_$i.json | head -3)
  echo "run $i rc $rc $errs" >> $O/summary.txt
  if [ $rc -ne 0 ]; then fails=$((fails+1)); else [ -z "$errs" ] && rm -f $O/err_$i.log $O/out_$i.json; fi
done
echo "fails $fails of $N ($TAG: $@)" | tee -a $O/summary.txt
