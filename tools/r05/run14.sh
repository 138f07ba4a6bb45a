#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/r05/pmc_busy_gn_all -- python $GRAFT_REPO_ROOT/bench.py --fast --steps 1 --warmup 1 --cpu-images 0 --no-extras --fast-steps 0 --profile none --profile-steps 1 --raster-gn -1 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/r05/pmc_busy_gn_4 -- python $GRAFT_REPO_ROOT/bench.py --fast --steps 1 --warmup 1 --cpu-images 0 --no-extras --fast-steps 0 --profile none --profile-steps 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for a in gn_all gn_4; do python tools/pmc_summary.py gpurun_out/r05/pmc_busy_$a "" > gpurun_out/r05/pmc_busy_${a}_summary.txt 2>&1; echo "== $a"; grep -A4 "KERNEL.*gemm_pp6_kernel<T_F16, [67]" gpurun_out/r05/pmc_busy_${a}_summary.txt | head -12; grep "DURATION.*gemm_pp6_kernel<T_F16, [67]" gpurun_out/r05/pmc_busy_${a}_summary.txt; done
find gpurun_out/r05 -name "*.csv" -size +5M -delete
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4
