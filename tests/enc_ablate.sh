#!/bin/bash
# per-kernel time for several phase-skip masks (debug): usage: enc_ablate.sh ENVVAR kernelname masks...
export TMPDIR=/tmp
var=$1; kn=$2; shift 2
for m in "$@"; do
  env $var=$m timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ea_$m -o k -- python3 tests/gpu_kern_times.py > /dev/null 2>&1
  echo -n "skip=$m  "; python tests/prof_summary.py gpurun_out/ea_$m/k_kernel_stats.csv 6 | grep $kn | cut -c60-
done
