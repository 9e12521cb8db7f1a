#!/bin/bash
# Same-box A/B of library builds on the whole sampling step: tests/ab_bench.sh <variant>... (build/lib_<variant>.so);
# per variant the kernel summary of `bench.py --steps 1 --warmup 1` under rocprofv3 and the bench line's value.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in "$@"; do
  cp build/lib_$v.so diffusionnag_amd/libdnag.so || exit 1
  echo "== $v"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abb_$v -o k -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/abb_$v.log 2>&1 || { tail -5 gpurun_out/abb_$v.log; exit 1; }
  python3 tests/prof_summary.py gpurun_out/abb_$v/k_kernel_stats.csv 2>&1 | head -6
  python3 -c "
import json,sys
for l in open('gpurun_out/abb_$v.log'):
    if l.startswith('{'): d=json.loads(l); print('samples/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],1))
"
done
