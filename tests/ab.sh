#!/bin/bash
# Same-box A/B of library builds: tests/ab.sh <variant>... ; each variant is build/lib_<variant>.so.
# Per variant: one correctness spot check (cate golden test), then kernel times of tests/gpu_kern_times.py under rocprofv3.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in "$@"; do
  cp build/lib_$v.so diffusionnag_amd/libdnag.so || exit 1
  echo "== $v"
  timeout -k 10 200 python -u -m pytest tests/test_gpu_parity.py -q -m gpu -k "cate_forward or first_steps or predictor" --timeout 120 --timeout-method thread > gpurun_out/ab_$v.test.log 2>&1
  rc=$?; tail -1 gpurun_out/ab_$v.test.log
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$v -o k -- python3 tests/gpu_kern_times.py > gpurun_out/ab_$v.log 2>&1 || { tail -5 gpurun_out/ab_$v.log; exit 1; }
  python3 tests/prof_summary.py gpurun_out/ab_$v/k_kernel_stats.csv 2>&1 | head -12
done
