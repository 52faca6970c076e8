# HBM traffic of the dominant kernels: two PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python tools/pmc_traffic.py > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python tools/pmc_traffic.py > gpurun_out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- python tools/pmc_traffic.py > gpurun_out/pmc_mfma.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_mfma > gpurun_out/pmc_mfma_summary.txt
python tools/pmc_summary.py gpurun_out/pmc_fetch > gpurun_out/pmc_fetch_summary.txt
python tools/pmc_summary.py gpurun_out/pmc_write > gpurun_out/pmc_write_summary.txt
tail -2 gpurun_out/pmc_fetch.log
