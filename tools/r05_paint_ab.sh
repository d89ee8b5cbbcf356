#!/bin/bash
out=gpurun_out/r05_paint_ab.txt
: > $out
for m in 0 0x4000 0x8000 0xE000 0x10000; do
  echo "== CKL_ABLATE=$m" >> $out
  CKL_TUNING_LIB=1 CKL_ABLATE=$m CKL_ABLATE_NOCHECK=1 python3 tools/stage_diag.py 2>&1 | grep stages | tail -2 >> $out
done
cat $out
