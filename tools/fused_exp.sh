#!/bin/bash
# timing experiments on k_strip_fused (tuning build, CKL_EXP bits; results may be wrong: no check)
out=gpurun_out/$1; mkdir -p $out; : > $out/exp.txt
for ex in 0 1 2 4 8 16 32 3 48 9; do
  echo "== CKL_EXP=$ex" >> $out/exp.txt
  CKL_TUNING_LIB=1 CKL_ABLATE_NOCHECK=1 CKL_EXP=$ex timeout -k 10 200 python tools/stage_diag.py 2>&1 | grep -A1 "iter 2" >> $out/exp.txt || echo failed >> $out/exp.txt
done
cat $out/exp.txt
