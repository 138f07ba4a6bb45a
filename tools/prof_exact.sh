#!/bin/bash
# rocprofv3 kernel-trace summary of the exact mode on N images (default 52 = 13 panoramas, a typical re-encode set).
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-52}
OUT=$REPO/gpurun_out/prof_exact
mkdir -p $OUT
python $REPO/tools/exact_prof.py $N 3 | tee $OUT/timing.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $REPO/tools/exact_prof.py $N 2 > $OUT/trace.log 2>&1
STATS=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $STATS $OUT/kernel_stats.csv
head -14 $OUT/kernel_stats.csv | cut -c1-200
find $OUT -name "*kernel_trace.csv" -size +20M -delete
rm -rf $OUT/trace
