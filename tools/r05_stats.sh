#!/bin/bash
root=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/r05s
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r05s -o s -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/r05s.log 2>&1
f=$(find /tmp/r05s -name "*kernel_stats.csv" | head -1)
cp $f $root/gpurun_out/r05_kernel_stats_wip.csv
python3 $root/tools/kstat.py $f k_ ckl > $root/gpurun_out/r05_kstat.txt
cat $root/gpurun_out/r05_kstat.txt
