#!/bin/bash
# One GPU-box visit: new tests, whole GPU suite, bench lines, rocprof summary.  Each step has its own time
# limit; a step that ends by signal / abort / timeout stops the script (an ordinary test failure does not).
mkdir -p gpurun_out
run() {   # run <seconds> <log> <cmd...>
    local secs=$1 log=$2; shift 2
    timeout -k 10 "$secs" "$@" > "gpurun_out/$log" 2>&1
    local rc=$?
    echo "[$log] rc=$rc"
    if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "stopping: $log ended with $rc"; tail -5 "gpurun_out/$log"; exit $rc; fi
    return 0
}
for step in "$@"; do
case $step in
  new)    run 400 new_tests.log python -u -m pytest tests/test_gpu_range.py tests/test_gpu_shard.py tests/test_zz_no_alloc.py -v -s -m gpu --timeout 120 --timeout-method thread; tail -40 gpurun_out/new_tests.log ;;
  suite)  run 900 gpu_tests.log python -u -m pytest tests -m gpu -x -v --timeout 120 --timeout-method thread; tail -15 gpurun_out/gpu_tests.log ;;
  bench)  run 300 bench_default.log python bench.py; tail -1 gpurun_out/bench_default.log ;;
  train)  run 200 bench_train.log python bench.py --workload train; tail -1 gpurun_out/bench_train.log ;;
  ofa)    run 300 bench_ofa.log python bench.py --workload ofa --no-cpu-baseline; tail -1 gpurun_out/bench_ofa.log ;;
  prof)   cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
          run 300 prof.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra
          tail -1 gpurun_out/prof.log ;;
  profofa) run 300 prof_ofa.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ofa -o ofa -- python3 bench.py --workload ofa --steps 1 --warmup 0 --sde-steps 100 --no-cpu-baseline
          tail -1 gpurun_out/prof_ofa.log ;;
  proftrain) run 200 prof_train.log rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -o train -- python3 bench.py --workload train --steps 20
          tail -1 gpurun_out/prof_train.log ;;
  pmc)    # HBM/fabric traffic of the per-evaluation kernels: separate passes, as the guide's HBM section prescribes
          for c in FETCH_SIZE WRITE_SIZE; do
            run 150 pmc_$c.log timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -o k -- python3 tests/gpu_kern_times.py
            python3 tests/pmc_summary.py gpurun_out/pmc_$c/k_counter_collection.csv > gpurun_out/pmc_$c.txt 2>&1; cat gpurun_out/pmc_$c.txt
          done ;;
  smoke)  run 200 smoke.log python __graft_entry__.py smoke; tail -2 gpurun_out/smoke.log ;;
esac
done
