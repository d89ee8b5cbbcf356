#!/bin/bash
# stage timings: C2 (u32), C3 slab (u64), C1; plus the opt-in fused path for correctness
out=gpurun_out/$1; mkdir -p $out; : > $out/paint.txt
echo "== C2 u32" >> $out/paint.txt
timeout -k 10 200 python tools/stage_diag.py 2>&1 | grep -A1 "iter 2" >> $out/paint.txt || exit 1
for cfg in "--shape 1024x1024x128 --dtype uint64" "--shape 512x512x128" "--shape 2048x2048x32 --markov 5"; do
  echo "== $cfg" >> $out/paint.txt
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 $cfg 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(f'{d[\"value\"]/1e9:.1f} GVx/s encode {d[\"encode_ms\"]:.2f} ms decode {d[\"decode_ms\"]:.2f} ms ok={d[\"roundtrip_ok\"]} stages={d[\"roofline\"][\"decode_stage_ms\"]}')" >> $out/paint.txt || exit 1
done
cat $out/paint.txt
