#!/bin/bash
# SQ counters of the per-evaluation kernels (one pass, 8 SQ counters) and the clock under load (GRBM pass); run on the GPU box
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_sq -o k -- python3 tests/gpu_kern_times.py > gpurun_out/pmc_sq.log 2>&1 || { tail -5 gpurun_out/pmc_sq.log; exit 1; }
python3 tests/pmc_summary.py gpurun_out/pmc_sq/k_counter_collection.csv > gpurun_out/pmc_sq.txt 2>&1; cat gpurun_out/pmc_sq.txt
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_grbm -o k -- python3 tests/gpu_kern_times.py > gpurun_out/pmc_grbm.log 2>&1 || { tail -5 gpurun_out/pmc_grbm.log; exit 1; }
python3 tests/pmc_summary.py gpurun_out/pmc_grbm/k_counter_collection.csv > gpurun_out/pmc_grbm.txt 2>&1; cat gpurun_out/pmc_grbm.txt
