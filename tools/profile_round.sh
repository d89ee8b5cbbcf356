#!/bin/bash
# usage: tools/profile_round.sh TAG [ROUND]   (on the GPU box, from the repo root; ROUND defaults to r05)
# kernel stats, the two PMC passes, a bench line of the same build (with roofline.traffic from those passes) and the
# same line through the sharded path as a process group of one -> gpurun_out/<TAG>_*  (copy what is to be judged
# into profiles/ afterwards: gpurun_out/ is scratch)
set -e
tag=$1
round=${2:-r05}
root=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/${tag}_stats $root/gpurun_out/${tag}_fetch $root/gpurun_out/${tag}_write
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_stats -o s -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $root/gpurun_out/${tag}_stats.log 2>&1
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex ckl --output-format csv -d $root/gpurun_out/${tag}_fetch -o f -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $root/gpurun_out/${tag}_fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex ckl --output-format csv -d $root/gpurun_out/${tag}_write -o w -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $root/gpurun_out/${tag}_write.log 2>&1
echo "write done"
cd $root
python3 tools/pmc_summary.py gpurun_out/${tag}_fetch gpurun_out/${tag}_write "1024x1024x512 uint32 markov 0" gpurun_out/${tag}_pmc_traffic.json > gpurun_out/${tag}_pmc.txt
cp gpurun_out/${tag}_pmc_traffic.json profiles/${round}_pmc_traffic.json   # bench.py below reads roofline.traffic from it (same build: lib_sha16)
# the raw traces are tens of MiB: keep the summaries only
mkdir -p gpurun_out/${tag}
find gpurun_out/${tag}_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}/kernel_stats.csv \;
rm -rf gpurun_out/${tag}_stats gpurun_out/${tag}_fetch gpurun_out/${tag}_write
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 600 gpurun_out/${tag}_bench.json
# the same line through the sharded path as a process group of one over RCCL (what N ranks pay per rank)
CKL_BENCH_REHEARSAL=group1 python3 bench.py --no-cpu-baseline > gpurun_out/${tag}_bench_group1.json 2> gpurun_out/${tag}_bench_group1.err || echo "group-of-one run failed"
tail -c 300 gpurun_out/${tag}_bench_group1.json
