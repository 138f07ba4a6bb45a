#!/bin/bash
# MFMA-busy / effective clock / LDS conflicts of the three GEMM mainloops side by side: variant 36 (8-wave ping-pong), variant 64
# (one wave per SIMD) and hipBLASLt's kernel, same shapes, one rocprofv3 counter pass each.
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_w4
mkdir -p $OUT
PMC="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
rocprofv3 --kernel-trace --output-format csv --pmc $PMC -d $OUT/ours -- python $REPO/tools/epi_probe.py 36,64 > $OUT/ours.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc $PMC -d $OUT/blaslt -- python $REPO/tools/blaslt_probe.py --iters 4 > $OUT/blaslt.log 2>&1
python $REPO/tools/pmc_summary.py $OUT/ours gemm > $OUT/summary_ours.txt
python $REPO/tools/pmc_summary.py $OUT/blaslt Cijk > $OUT/summary_blaslt.txt
cat $OUT/summary_ours.txt $OUT/summary_blaslt.txt | grep -v "^STATS\|^   \"" | head -150
find $OUT -name "*.csv" -size +3M -delete
