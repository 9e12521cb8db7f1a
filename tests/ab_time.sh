#!/bin/bash
# Same-box kernel times of library builds WITHOUT the correctness check (ablation builds compute wrong values):
# tests/ab_time.sh <variant>... ; each variant is build/lib_<variant>.so.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in "$@"; do
  cp build/lib_$v.so diffusionnag_amd/libdnag.so || exit 1
  echo "== $v"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abt_$v -o k -- python3 tests/gpu_kern_times.py > gpurun_out/abt_$v.log 2>&1 || { tail -5 gpurun_out/abt_$v.log; exit 1; }
  python3 tests/prof_summary.py gpurun_out/abt_$v/k_kernel_stats.csv 2>&1 | grep -E "predictor|encoder"
done
