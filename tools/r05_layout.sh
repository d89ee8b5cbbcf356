#!/bin/bash
out=gpurun_out/r05_layout.txt
: > $out
for m in 0 1 2 3 7; do
  echo "== CKL_STRIP_LAYOUT=$m" >> $out
  CKL_STRIP_LAYOUT=$m python3 tools/stage_diag.py 2>&1 | grep "stages\|iter" | tail -4 >> $out
done
cat $out
