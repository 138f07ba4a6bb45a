#!/bin/bash
# rocprofv3 counter passes over the attention kernel.  usage: tools/prof_attn.sh <tag>
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-attn}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python $REPO/tools/attn_bench.py --iters 2 --rounds 2"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc1 -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA -d $OUT/pmc2 -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM -d $OUT/pmc3 -- $CMD > $OUT/pmc3.log 2>&1
for p in pmc1 pmc2 pmc3; do echo "== $p"; tail -1 $OUT/$p.log; python $REPO/tools/pmc_summary.py $OUT/$p attention > $OUT/${p}_summary.txt; cat $OUT/${p}_summary.txt; done
find $OUT -name "*.csv" -size +5M -delete
