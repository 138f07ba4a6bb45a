#!/bin/bash
# rocprofv3 SQ counter pass + kernel durations for GEMM variants.  usage: tools/prof_pp.sh <shapes> <variants> <tag> [images]
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
SHAPES=${1:-fc2}; VARS=${2:-33}; TAG=${3:-pp}; IMAGES=${4:-512}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python $REPO/tools/gemm_prof.py --shape $SHAPES --variants $VARS --iters 2 --rounds 2 --images $IMAGES"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc1 -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc2 -- $CMD > $OUT/pmc2.log 2>&1
for p in pmc1 pmc2; do echo "== $p"; tail -2 $OUT/$p.log; python $REPO/tools/pmc_summary.py $OUT/$p gemm > $OUT/${p}_summary.txt; cat $OUT/${p}_summary.txt; done
find $OUT -name "*.csv" -size +5M -delete
du -sh $OUT
