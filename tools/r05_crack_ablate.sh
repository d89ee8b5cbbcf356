#!/bin/bash
# k_crack_match at C2 with parts of its record pass left out (tuning build, results wrong): what the scattered record stores / the cursor atomics cost
out=gpurun_out/r05_crack_ablate.txt
: > $out
for v in 0 0x400000 0x800000 0x1000000 0x1400000; do
  echo "== CKL_ABLATE=$v" >> $out
  CKL_ABLATE=$v CKL_ABLATE_NOCHECK=1 CKL_TUNING_LIB=1 python3 tools/stage_diag.py 2>&1 | grep -v amdgpu.ids | grep stages | tail -1 >> $out
done
cat $out
