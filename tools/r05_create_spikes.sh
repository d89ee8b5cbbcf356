#!/bin/bash
# how often does a bench process see a decoder set-up of several milliseconds (one step in ~20), with the runtime's default
# number of hardware queues and with more: usage tools/r05_create_spikes.sh RUNS [ENV=VALUE ...]
runs=$1; shift
bad=0
for i in $(seq 1 $runs); do
  m=$(env "$@" python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = d['per_step_ms']['decoder_create'] + d['per_step_ms']['encode'] 
print(round(max(d['per_step_ms']['decoder_create']), 2), round(max(d['per_step_ms']['encode']), 2), round(max(d['per_step_ms']['decode']), 2), round(d['value'] / 1e9, 1))")
  echo "$m"
done
