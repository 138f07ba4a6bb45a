#!/bin/bash
# v6 (variant 11, 32x32x16) against v8 (variant 21, 16x16x32): clocks / pipe busy, and the LDS side (bank conflicts must stay 0).
# Tools build (it holds both kernels).  One PMC pass each per variant, written to gpurun_out/prof_attn8/.
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_attn8
mkdir -p $OUT
export PIGEON_HIP_LIB=$REPO/pigeon_amd/libpigeon_hip_dev.so
CMD="python $REPO/tools/attn_bench.py --iters 2 --rounds 2"
for v in 11 21; do
  PIGEON_ATTN_VARIANT=$v rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA -d $OUT/v${v}_pmc1 -- $CMD > $OUT/v${v}_pmc1.log 2>&1
  PIGEON_ATTN_VARIANT=$v rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/v${v}_pmc2 -- $CMD > $OUT/v${v}_pmc2.log 2>&1
  for p in pmc1 pmc2; do echo "== variant $v $p"; python $REPO/tools/pmc_summary.py $OUT/v${v}_$p attention; done
done > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -size +5M -delete
cat $OUT/summary.txt
