#!/bin/bash
# usage: tools/ab_lib.sh NAME [stage_diag args]   (on the GPU box, from the repo root)
# stage times at C2 (or the given shape) of libcrackle_amd_NAME.so (an earlier build) and of the current library, alternating, in one call
name=$1; shift
out=gpurun_out/ab_${name}.txt
: > $out
for r in 1 2; do
  echo "== $name (run $r)" >> $out
  CKL_LIB_AB=$name python3 tools/stage_diag.py "$@" 2>&1 | grep -v amdgpu.ids | tail -4 >> $out || exit 1
  echo "== current (run $r)" >> $out
  python3 tools/stage_diag.py "$@" 2>&1 | grep -v amdgpu.ids | tail -4 >> $out || exit 1
done
cat $out
