#!/bin/bash
# usage: tools/ab_env.sh REPS "ENV1=.. ENV2=.." "ENV.." ...   (bench.py under each environment, REPS times, interleaved)
reps=$1; shift
for r in $(seq 1 $reps); do
  i=0
  for cfg in "$@"; do
    env $cfg python bench.py > gpurun_out/ab_${i}_${r}.json 2> gpurun_out/ab_${i}_${r}.err || exit 1
    python - "$cfg" gpurun_out/ab_${i}_${r}.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
st = d["roofline"]["decode_stage_ms"]
stages = " ".join(f"{k}={v:.3f}" for k, v in st.items())
print(f"{sys.argv[1]:40s} GVx/s={d['value']/1e9:6.2f} enc={d['encode_ms']:.3f} dec={d['decode_ms']:.3f} dfs={d['encode_dfs_kernel_ms']:.3f} ok={d['roundtrip_ok']} {stages}", flush=True)
PY
    i=$((i+1))
  done
done
