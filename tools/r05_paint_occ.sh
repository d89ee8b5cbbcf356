#!/bin/bash
out=gpurun_out/r05_paint_occ.txt
: > $out
for w in 0 1 2 3 4; do
  echo "== CKL_PAINT_WGS=$w" >> $out
  CKL_PAINT_WGS=$w python3 tools/stage_diag.py 2>&1 | grep stages | tail -2 >> $out
done
echo "== C3 slab u64, wgs 0 then 2" >> $out
CKL_PAINT_WGS=0 python3 tools/stage_diag.py 1024 1024 128 2>&1 | grep stages | tail -1 >> $out
python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --shape 1024x1024x128 --dtype uint64 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['decode_stage_ms'], d['roundtrip_ok'])" >> $out
CKL_PAINT_WGS=0 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --shape 1024x1024x128 --dtype uint64 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['decode_stage_ms'], d['roundtrip_ok'])" >> $out
cat $out
