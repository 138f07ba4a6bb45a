#!/bin/bash
# Per-kernel durations of the four GEMM forms at 512 images with the tail split on (default) and off: how long gemm_tail.hip
# takes against the persistent kernels' round it removes.
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_tail
mkdir -p $OUT
for t in 768 0; do
  PIGEON_GEMM_TAIL_ROWS=$t rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$t -- python $REPO/tools/gemm_pp_check.py --variants 56 --skip-check --time --images 512 --rounds 2 > $OUT/t$t.log 2>&1
  f=$(find $OUT/t$t -name "*kernel_stats.csv" | head -1)
  echo "== PIGEON_GEMM_TAIL_ROWS=$t"; grep TIME $OUT/t$t.log
  python - "$f" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "gemm" in n:
        print(f'{n[:95]:95s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}  max {float(r["MaxNs"])/1e3:9.1f}')
P
done 2>&1 | tee $OUT/summary.txt
find $OUT -name "*.csv" -size +3M -delete
