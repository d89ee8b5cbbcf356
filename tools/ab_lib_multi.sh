#!/bin/bash
# usage: tools/ab_lib_multi.sh NAME   (on the GPU box, from the repo root)
# like ab_lib.sh, over the decode shapes that matter: C2, C2 with a markov model, a C4 slab with and without one, a C3 slab
name=$1
out=gpurun_out/abm_${name}.txt
: > $out
run() {
  echo "== $* : $name | current" >> $out
  CKL_LIB_AB=$name python3 tools/stage_diag.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1 >> $out || exit 1
  python3 tools/stage_diag.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1 >> $out || exit 1
}
run 1024 1024 512
run 1024 1024 512 --markov 5
run 2048 2048 32 --markov 5
run 2048 2048 32
run 512 512 128
cat $out
