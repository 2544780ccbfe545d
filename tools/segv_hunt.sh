#!/bin/bash
# Repeat the invocation that died with SIGSEGV once in round 5 (four legs in ONE process: c3, c2, c5, c5-fp8) with
# faulthandler on; keep stderr of every failing run.  usage: tools/segv_hunt.sh N [extra bench args]
cd $GRAFT_REPO_ROOT
N=${1:-20}; shift
O=gpurun_out/segv_hunt; mkdir -p $O
export CSMRI_BENCH_DIAG=1 AMD_LOG_LEVEL=1
fails=0
for i in $(seq 1 $N); do
  python -X faulthandler bench.py --steps 3 --warmup 2 --no-cpu-baseline --settle-s 0.2 --inprocess-legs "$@" > $O/out_$i.json 2> $O/err_$i.log
  rc=$?
  diag=$(grep -c "DIAG" $O/err_$i.log)
  errs=$(grep -o '"error": "[^"]*"' $O/out_$i.json | head -3)
  echo "run $i rc $rc diag $diag $errs" | tee -a $O/summary.txt
  if [ $rc -ne 0 ]; then fails=$((fails+1)); else [ "$diag" = "0" ] && [ -z "$errs" ] && rm -f $O/err_$i.log $O/out_$i.json; fi
done
echo "fails $fails of $N" | tee -a $O/summary.txt
