#!/bin/bash
# A/B of the strip kernels at C2 (on the GPU box, from the repo root): stage times of the shipped build per variant,
# then the per-workgroup phase stamps of the tuning build.
out=gpurun_out/r05_strip_ab.txt
: > $out
for v in "CKL_STRIP_V1=1" "CKL_STRIP_VARIANT=0" "CKL_STRIP_VARIANT=1"; do
  echo "== $v (tuning build, no stamps)" >> $out
  env $v CKL_TUNING_LIB=1 python3 tools/stage_diag.py 2>&1 | grep -v amdgpu.ids >> $out
done
echo "== shipped build" >> $out
python3 tools/stage_diag.py 2>&1 | grep -v amdgpu.ids >> $out
for v in "CKL_STRIP_VARIANT=0" "CKL_STRIP_VARIANT=1"; do
  echo "== $v stamps" >> $out
  env $v CKL_TUNING_LIB=1 CKL_STRIP_DIAG=1 python3 tools/stage_diag.py 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
