#!/bin/bash
# usage: tools/timeline.sh TAG   (on the GPU box, from the repo root): kernel timeline of the last encode of a bench run
tag=$1
root=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/${tag}_tl
rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/${tag}_tl -o t -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $root/gpurun_out/${tag}_tl.log 2>&1
cd $root
f=$(find gpurun_out/${tag}_tl -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py "$f" > gpurun_out/${tag}_timeline.txt
rm -rf gpurun_out/${tag}_tl
