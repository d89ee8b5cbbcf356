#!/bin/bash
# bench.py on the other BASELINE.json configurations (one GPU's slab each) -> gpurun_out/other_configs.txt
out=gpurun_out/other_configs.txt
: > $out
run() {
  name=$1; shift
  python bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(f'$name: {d[\"value\"]/1e9:.1f} GVx/s encode {d[\"encode_ms\"]:.2f} ms decode {d[\"decode_ms\"]:.2f} ms (device pipeline {d[\"decode_device_pipeline_ms\"]:.3f} ms = {100*d[\"roofline\"][\"frac\"]:.1f} % of 8 TB/s; encode pipeline {100*d[\"roofline_encode\"][\"frac\"]:.1f} %) ok={d[\"roundtrip_ok\"]} walk={d[\"encode_dfs_kernel_ms\"]:.2f} stages={ {k: round(v, 3) for k, v in d[\"roofline\"][\"decode_stage_ms\"].items()} }')" >> $out
}
run "C2 markov 5" --markov 5
run "C1 512x512x128 u32" --shape 512x512x128
run "C3 slab 1024x1024x128 u64" --shape 1024x1024x128 --dtype uint64
run "C4 slab 2048x2048x32 u32 markov 5" --shape 2048x2048x32 --markov 5
run "C4 slab 2048x2048x32 u32 pins" --shape 2048x2048x32 --pins 1 --steps 2 --warmup 1
# the reference's adversarial inputs (benchmarks/README.md:108-114, 193-227), reported for honesty
run "noise2000 1024x1024x64 u32 (PERMISSIBLE)" --shape 1024x1024x64 --data noise2000 --steps 3 --warmup 1
run "binary noise 1024x1024x64 u32" --shape 1024x1024x64 --data binary --steps 3 --warmup 1
cat $out
