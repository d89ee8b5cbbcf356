#!/bin/bash
# usage: tools/icache_profile.sh TAG   (on the GPU box, from the repo root)
# instruction-fetch counters of the decode kernels at C2 (one rocprofv3 --pmc pass, program directly after `--`)
tag=$1
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $root/gpurun_out/${tag}_avail.txt 2>&1
grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQC_INST[A-Z_]*" $root/gpurun_out/${tag}_avail.txt | sort -u > $root/gpurun_out/${tag}_icache_names.txt
cat $root/gpurun_out/${tag}_icache_names.txt
g="SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAIT_INST_ANY"
d=$root/gpurun_out/${tag}_ic
rm -rf $d
rocprofv3 --kernel-trace --pmc $g --kernel-include-regex ckl --output-format csv -d $d -o q -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $root/gpurun_out/${tag}_ic.log 2>&1 || echo "pass failed"
cd $root
python3 - <<PY
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/${tag}_ic/**/*counter_collection.csv", recursive=True):
  for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    if not any(s in k for s in ("k_crack_match", "k_strip_ccl2", "k_slice_resolve", "k_paint_strips")): continue
    rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
with open("gpurun_out/${tag}_icache.txt", "w") as o:
  for k, v in rows.items():
    line = k + " launches=%d  " % n[k] + "  ".join("%s=%.4g" % (c, x / max(1, n[k])) for c, x in sorted(v.items()))
    print(line); o.write(line + "\n")
PY
