#!/bin/bash
# second SQ pass: co-execution of MFMA and VALU, instruction mix, LDS / VMEM activity
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_sq2 -o k -- python3 tests/gpu_kern_times.py > gpurun_out/pmc_sq2.log 2>&1 || { tail -5 gpurun_out/pmc_sq2.log; exit 1; }
python3 tests/pmc_summary.py gpurun_out/pmc_sq2/k_counter_collection.csv > gpurun_out/pmc_sq2.txt 2>&1; cat gpurun_out/pmc_sq2.txt
