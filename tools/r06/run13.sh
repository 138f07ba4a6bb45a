#!/bin/bash
# round 6, GPU session 13: rocprofv3 kernel stats of the encoder on ONE panorama (4 images): which kernels a serving request runs
mkdir -p gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06/prof_latency
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $GRAFT_REPO_ROOT/tools/latency_probe.py 4 > $OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
STATS=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $STATS gpurun_out/r06/latency4_kernel_stats.csv
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/r06/latency4_kernel_stats.csv')))
for r in rows[:16]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:8.1f} pct {r['Percentage']}")
PY
rm -rf $OUT
