#!/bin/bash
out=gpurun_out/r05_chunks.txt
: > $out
for c in 1 2 4; do for w in 2 0; do
  echo "== CKL_DECODE_CHUNKS=$c CKL_PAINT_WGS=$w" >> $out
  CKL_DECODE_CHUNKS=$c CKL_PAINT_WGS=$w python3 tools/stage_diag.py 2>&1 | grep "iter" | tail -2 >> $out
done; done
cat $out
