#!/bin/bash
# PMC passes over the exact mode (tools/exact_prof.py): memory-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and MFMA-busy
# cycles of its kernels.  Counters are collected in runs of their own with --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-52}
OUT=$REPO/gpurun_out/prof_exact_pmc
mkdir -p $OUT
CMD="python $REPO/tools/exact_prof.py $N 1"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES -d $OUT/busy -- $CMD > $OUT/busy.log 2>&1
for p in fetch write busy; do python $REPO/tools/pmc_summary.py $OUT/$p "" > $OUT/${p}_summary.txt; done
grep -A6 "attention_f32\|split_x3_kernel<true>\|ln_x3\|gemm_pp_kernel<T_F16, 4" $OUT/busy_summary.txt | head -60
grep -A2 "attention_f32\|split_x3_kernel<true>\|ln_x3_kernel\|gemm_pp_kernel<T_F16, 4" $OUT/fetch_summary.txt | head -30
grep -A2 "attention_f32\|split_x3_kernel<true>\|ln_x3_kernel\|gemm_pp_kernel<T_F16, 4" $OUT/write_summary.txt | head -30
find $OUT -name "*.csv" -size +2M -delete
