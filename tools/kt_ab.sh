#!/bin/bash
# usage: tools/kt_ab.sh NAME kernel [kernel ...]   (on the GPU box, from the repo root)
# average durations of the named kernels in bench.py's C2 step: libcrackle_amd_NAME.so | the current library (two rocprofv3 kernel traces)
name=$1; shift
export CKL_LIB_AB=$name; tools/kernel_times.sh ab_$name > /dev/null 2>&1; unset CKL_LIB_AB
tools/kernel_times.sh ab_cur > /dev/null 2>&1
for k in "$@"; do echo "$k: $(grep $k gpurun_out/ab_${name}_kt.txt | head -1 | sed "s/.*avg//") | $(grep $k gpurun_out/ab_cur_kt.txt | head -1 | sed "s/.*avg//")"; done
