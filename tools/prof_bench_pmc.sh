#!/bin/bash
# MFMA-pipe utilisation and effective clock of every kernel of the benchmark step (one rocprofv3 counter pass).
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_bench_pmc
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/pmc -- python $REPO/bench.py --steps 1 --warmup 1 --cpu-images 0 --no-extras --fast-steps 0 --profile none --profile-steps 1 > $OUT/pmc.log 2>&1
tail -1 $OUT/pmc.log | cut -c1-120
python $REPO/tools/pmc_summary.py $OUT/pmc "" > $OUT/summary.txt
grep -A9 "gemm_pp_kernel<T_F16, [567]\|gemm_pp6_kernel<T_F16\|attention8" $OUT/summary.txt | head -80
find $OUT -name "*.csv" -size +5M -delete
