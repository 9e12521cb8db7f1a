#!/bin/bash
# Same-box A/B on the OFA guided step: tests/ab_ofa.sh <variant>... (build/lib_<variant>.so); kernel summary of 100 PC steps
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in "$@"; do
  cp build/lib_$v.so diffusionnag_amd/libdnag.so || exit 1
  echo "== $v"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abo_$v -o k -- python3 bench.py --workload ofa --steps 1 --warmup 0 --sde-steps 100 --no-cpu-baseline > gpurun_out/abo_$v.log 2>&1 || { tail -5 gpurun_out/abo_$v.log; exit 1; }
  python3 tests/prof_summary.py gpurun_out/abo_$v/k_kernel_stats.csv 2>&1 | head -6
done
