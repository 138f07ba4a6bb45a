#!/bin/bash
# rocprofv3 counter passes over one GEMM shape.  usage: tools/prof_gemm.sh <shape> <variant> <tag>
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
SHAPE=${1:-qkv}; VAR=${2:-1}; TAG=${3:-gemm}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python $REPO/tools/gemm_prof.py --shape $SHAPE --variants $VAR --iters 4"
rocprofv3 -L > $OUT/counters_list.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc1 -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc2 -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE TCC_REQ_sum -d $OUT/pmc3 -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_EA0_WRREQ_sum -d $OUT/pmc4 -- $CMD > $OUT/pmc4.log 2>&1
for p in pmc1 pmc2 pmc3 pmc4; do echo "== $p"; tail -3 $OUT/$p.log; python $REPO/tools/pmc_summary.py $OUT/$p gemm; done
python $REPO/tools/pmc_summary.py $OUT/trace gemm | tail -20
# keep the merge small
du -sh $OUT
