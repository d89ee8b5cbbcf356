#!/bin/bash
# over-segmented volumes (the reference's watershed benchmark, benchmarks/README.md:284-318): which kernels answer, how fast
out=gpurun_out/r05_dense.txt
: > $out
run() {
  name=$1; shift
  python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>gpurun_out/r05_dense.err | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(f'$name: {d[\"value\"]/1e9:.1f} GVx/s encode {d[\"encode_ms\"]:.2f} ms decode {d[\"decode_ms\"]:.2f} ms (device pipeline {d[\"decode_device_pipeline_ms\"]:.3f} ms = {100*d[\"roofline\"][\"frac\"]:.1f} %) ok={d[\"roundtrip_ok\"]} walk={d[\"encode_dfs_kernel_ms\"]:.2f} ratio={d[\"compression_ratio_pct\"]:.2f}% stages={ {k: round(v, 3) for k, v in d[\"roofline\"][\"decode_stage_ms\"].items()} }')" >> $out 2>&1
}
run "cell 16x16x4 u64 1024x1024x128" --shape 1024x1024x128 --dtype uint64 --cell 16x16x4
run "cell 12x12x4 u64 1024x1024x128" --shape 1024x1024x128 --dtype uint64 --cell 12x12x4
run "cell 8x8x4 u64 1024x1024x128" --shape 1024x1024x128 --dtype uint64 --cell 8x8x4
run "cell 8x8x4 u32 1024x1024x512" --shape 1024x1024x512 --cell 8x8x4
cat $out
