#!/bin/bash
# usage: tools/sq_profile.sh TAG    (on the GPU box, from the repo root)
# SQ counter passes (rocprofv3 --pmc, one group per pass, program directly after `--`) of
# bench.py's C2 workload for the ckl kernels -> gpurun_out/<TAG>_sq_counters.json
# (folded by tools/sq_summary.py).  Groups of <= 8 SQ counters: MI355X_MICROARCH.md, PMC slots.
tag=$1
root=$PWD
cd /tmp && export TMPDIR=/tmp
groups=(
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM"
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU"
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_THREAD_CYCLES_VALU"
)
i=0
for g in "${groups[@]}"; do
  d=$root/gpurun_out/${tag}_sq$i
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $g --kernel-include-regex ckl --output-format csv -d $d -o q -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $root/gpurun_out/${tag}_sq$i.log 2>&1 \
    || echo "group $i failed (see gpurun_out/${tag}_sq$i.log)"
  echo "sq group $i done"
  i=$((i+1))
done
cd $root
python3 tools/sq_summary.py gpurun_out/${tag}_sq_counters.json gpurun_out/${tag}_sq0 gpurun_out/${tag}_sq1 gpurun_out/${tag}_sq2 > gpurun_out/${tag}_sq.txt
rm -rf gpurun_out/${tag}_sq0 gpurun_out/${tag}_sq1 gpurun_out/${tag}_sq2
cat gpurun_out/${tag}_sq.txt
