#!/bin/bash
# per-stage decode timings at C2 under CKL_ABLATE bit masks (tuning aid; some masks give wrong results)
for m in "$@"; do
  echo "== CKL_ABLATE=$m"
  CKL_ABLATE=$m CKL_ABLATE_NOCHECK=1 python tools/stage_diag.py 2>&1 | grep stages | tail -1
done
