#!/bin/bash
# C2 decode stage timings of k_strip_fused under its two knobs (lag in slices per ticket counter, number of counters)
out=gpurun_out/$1; mkdir -p $out
for cfg in "12 8" "6 8" "20 8" "32 8" "12 1" "12 4" "3 8"; do
  set -- $cfg
  echo "== lag=$1 heads=$2" >> $out/sweep.txt
  CKL_FUSED_LAG=$1 CKL_FUSED_HEADS=$2 timeout -k 10 200 python tools/stage_diag.py 2>&1 | grep -A1 "iter 2" >> $out/sweep.txt || exit 1
done
cat $out/sweep.txt
