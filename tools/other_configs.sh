#!/bin/bash
# bench.py on the other BASELINE.json configurations (one GPU's slab each), on the reference's adversarial inputs and on
# over-segmented volumes (its watershed benchmark), the last two groups with the compiled reference timed on the box's
# host cores beside them -> gpurun_out/other_configs.txt
out=gpurun_out/other_configs.txt
: > $out
run() {
  name=$1; shift
  python3 bench.py --steps 5 --warmup 2 "$@" 2>gpurun_out/other_configs.err | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
cpu = d.get('cpu_baseline')
cpu_s = f' | CPU reference ({cpu[\"cores\"]} threads, {cpu[\"kind\"]}): {cpu[\"value\"]/1e9:.3f} GVx/s (encode {cpu[\"encode_voxels_per_s\"]/1e9:.3f}, decode {cpu[\"decode_voxels_per_s\"]/1e9:.3f}) -> GPU/CPU {d[\"value\"]/cpu[\"value\"]:.1f}x (encode {d[\"encode_voxels_per_s\"]/cpu[\"encode_voxels_per_s\"]:.1f}x, decode {d[\"decode_voxels_per_s\"]/cpu[\"decode_voxels_per_s\"]:.1f}x), same bytes: {cpu.get(\"hip_bytes_equal_cpu_bytes_on_sample\")}' if cpu else ''
print(f'$name: {d[\"value\"]/1e9:.1f} GVx/s encode {d[\"encode_ms\"]:.2f} ms decode {d[\"decode_ms\"]:.2f} ms (device pipeline {d[\"decode_device_pipeline_ms\"]:.3f} ms = {100*d[\"roofline\"][\"frac\"]:.1f} % of 8 TB/s; encode pipeline {100*d[\"roofline_encode\"][\"frac\"]:.1f} %) ok={d[\"roundtrip_ok\"]} walk={d[\"encode_dfs_kernel_ms\"]:.2f} ratio={d[\"compression_ratio_pct\"]:.2f}% stages={ {k: round(v, 3) for k, v in d[\"roofline\"][\"decode_stage_ms\"].items()} }' + cpu_s)" >> $out 2>&1
}
run "C2 markov 5" --markov 5 --no-cpu-baseline
run "C1 512x512x128 u32" --shape 512x512x128 --no-cpu-baseline
run "C3 slab 1024x1024x128 u64" --shape 1024x1024x128 --dtype uint64 --no-cpu-baseline
run "C4 slab 2048x2048x32 u32 markov 5" --shape 2048x2048x32 --markov 5 --no-cpu-baseline
run "C4 slab 2048x2048x32 u32 pins" --shape 2048x2048x32 --pins 1 --steps 2 --warmup 1 --no-cpu-baseline
# the reference's adversarial inputs (benchmarks/README.md:108-114, 193-227), with the reference itself beside them
run "noise2000 1024x1024x64 u32 (PERMISSIBLE)" --shape 1024x1024x64 --data noise2000 --steps 3 --warmup 1
run "binary noise 1024x1024x64 u32" --shape 1024x1024x64 --data binary --steps 3 --warmup 1
# over-segmented labels (the reference's watershed benchmark, benchmarks/README.md:284-318: uint64, ~16 k segments per slice at cell 8x8)
run "watershed-like cell 16x16x4 u64 1024x1024x128" --shape 1024x1024x128 --dtype uint64 --cell 16x16x4 --steps 3 --warmup 1
run "watershed-like cell 12x12x4 u64 1024x1024x128" --shape 1024x1024x128 --dtype uint64 --cell 12x12x4 --steps 3 --warmup 1
run "watershed-like cell 8x8x4 u64 1024x1024x128" --shape 1024x1024x128 --dtype uint64 --cell 8x8x4 --steps 3 --warmup 1
run "watershed-like cell 8x8x4 u32 1024x1024x512" --shape 1024x1024x512 --cell 8x8x4 --steps 3 --warmup 1 --no-cpu-baseline
cat $out
