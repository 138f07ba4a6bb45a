#!/bin/bash
# clock / pipe-busy comparison of two attention variants (one PMC pass each + the instruction-mix pass for the first)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_attn2
mkdir -p $OUT
CMD="python $REPO/tools/attn_bench.py --iters 2 --rounds 2"
for v in 11 13; do
  PIGEON_ATTN_VARIANT=$v rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/v${v}_pmc1 -- $CMD > $OUT/v${v}_pmc1.log 2>&1
  PIGEON_ATTN_VARIANT=$v rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA -d $OUT/v${v}_pmc2 -- $CMD > $OUT/v${v}_pmc2.log 2>&1
  for p in pmc1 pmc2; do echo "== v$v $p"; python $REPO/tools/pmc_summary.py $OUT/v${v}_$p attention5 | tee $OUT/v${v}_${p}_summary.txt; done
done
find $OUT -name "*.csv" -size +5M -delete
