#!/bin/bash
# usage: tools/kernel_stats.sh TAG  -> gpurun_out/TAG_kernels.txt: average duration of every ckl kernel of bench.py (3 launches)
tag=$1
root=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/${tag}_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_stats -o s -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline $CKL_BENCH_ARGS > $root/gpurun_out/${tag}_stats.log 2>&1
cd $root
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/{tag}_stats/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "ckl" in r["Name"]]
with open(f"gpurun_out/{tag}_kernels.txt", "w") as out:
  for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    out.write(f"{r['Name'].split('(')[0][:58]:58s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs']) / 1e3:9.1f}\n")
PY
rm -rf gpurun_out/${tag}_stats
cat gpurun_out/${tag}_kernels.txt
