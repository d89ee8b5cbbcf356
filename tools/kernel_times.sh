#!/bin/bash
# usage: tools/kernel_times.sh TAG [bench args...]   (on the GPU box, from the repo root)
# one rocprofv3 kernel-trace of bench.py; prints the ckl kernels' average durations -> gpurun_out/<TAG>_kt.txt
tag=$1; shift
root=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/${tag}_kt
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_kt -o s -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $root/gpurun_out/${tag}_kt.log 2>&1
cd $root
f=$(find gpurun_out/${tag}_kt -name "*kernel_stats.csv" | head -1)
python3 - "$f" > gpurun_out/${tag}_kt.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'ckl' in r['Name']:
        print(f"{r['Name'][:60]:60s} calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
rm -rf gpurun_out/${tag}_kt
cat gpurun_out/${tag}_kt.txt
