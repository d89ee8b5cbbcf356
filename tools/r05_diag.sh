#!/bin/bash
# stage times of the shipped build, then the tuning build's in-kernel stamps (strip kernels, k_crack_match) at C2
out=gpurun_out/r05_diag.txt
: > $out
echo "== shipped build" >> $out
python3 tools/stage_diag.py 2>&1 | grep -v amdgpu.ids >> $out
echo "== strip stamps" >> $out
CKL_TUNING_LIB=1 CKL_STRIP_DIAG=1 python3 tools/stage_diag.py 2>&1 | grep -v amdgpu.ids | grep "diag" | tail -2 >> $out
echo "== crack stamps" >> $out
CKL_TUNING_LIB=1 CKL_CRACK_DIAG=1 python3 tools/stage_diag.py 2>&1 | grep -v amdgpu.ids | grep "diag" | tail -1 >> $out
cat $out
