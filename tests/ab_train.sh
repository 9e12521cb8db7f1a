#!/bin/bash
# Same-box A/B on the training step: tests/ab_train.sh <variant>... (build/lib_<variant>.so); kernel summary of 20 steps + steps/s
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in "$@"; do
  cp build/lib_$v.so diffusionnag_amd/libdnag.so || exit 1
  echo "== $v"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abt_$v -o k -- python3 bench.py --workload train --steps 20 > gpurun_out/abt_$v.log 2>&1 || { tail -5 gpurun_out/abt_$v.log; exit 1; }
  python3 tests/prof_summary.py gpurun_out/abt_$v/k_kernel_stats.csv 6 2>&1
  timeout -k 10 200 python3 bench.py --workload train > gpurun_out/abt_$v.bench.log 2>&1 || exit 1
  grep -o "\"value\": [0-9.]*" gpurun_out/abt_$v.bench.log
done
