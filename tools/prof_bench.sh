#!/bin/bash
# rocprofv3 kernel-trace summary (+ HBM traffic counters for the GEMM kernels) of the benchmark command.
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}
OUT=$REPO/gpurun_out/prof_bench_$TAG
mkdir -p $OUT
CMD="python $REPO/bench.py --steps 2 --warmup 1 --cpu-images 0 --no-extras --fast-steps 0 --profile-steps 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log
CMD2="python $REPO/bench.py --steps 1 --warmup 1 --cpu-images 0 --no-extras --fast-steps 0 --profile none --profile-steps 1"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -- $CMD2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -- $CMD2 > $OUT/pmc_write.log 2>&1
STATS=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $STATS $OUT/kernel_stats.csv
python $REPO/tools/pmc_summary.py $OUT/pmc_fetch "" > $OUT/pmc_fetch_summary.txt
python $REPO/tools/pmc_summary.py $OUT/pmc_write "" > $OUT/pmc_write_summary.txt
head -30 $OUT/kernel_stats.csv | cut -c1-220
python $REPO/tools/make_traffic.py $OUT $OUT/traffic.json 295424 | head -50
find $OUT -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
