#!/bin/bash
out=gpurun_out/r05_pad.txt
: > $out
for pad in 0 3000 8000 16000; do
  echo "== CKL_STRIP_PAD=$pad" >> $out
  CKL_STRIP_PAD=$pad CKL_TUNING_LIB=1 python3 tools/stage_diag.py 2>&1 | grep -v amdgpu.ids | tail -2 >> $out
done
cat $out
