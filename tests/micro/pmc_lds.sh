set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_lds -o k -- $R/tests/micro/gemm16_bench > $R/gpurun_out/pmc_lds.log 2>&1
python3 $R/tests/pmc_summary.py $R/gpurun_out/pmc_lds/k_counter_collection.csv
