#!/bin/bash
out=gpurun_out/$1; mkdir -p $out; : > $out/groups.txt
for cfg in "1 0" "2 1" "3 1" "4 1" "6 1" "8 1" "2 0"; do
  set -- $cfg
  echo "== groups=$1 stagger=$2" >> $out/groups.txt
  if [ "$2" = "1" ]; then export CKL_TRAIL_STAGGER=1; else unset CKL_TRAIL_STAGGER; fi
  CKL_TRAIL_GROUPS=$1 timeout -k 10 200 python tools/stage_diag.py 2>&1 | grep "iter" >> $out/groups.txt || exit 1
done
cat $out/groups.txt
