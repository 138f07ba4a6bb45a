#!/bin/bash
# round 6, GPU session 4: kernel-level profile of the bench command (what does a step's certainty / queueing cost; is any ATen kernel left
# between a step's launches?), of the exact pass at 44 images, and the handshake-vs-barrier skeleton probe
mkdir -p gpurun_out/r06
hipcc --offload-arch=gfx950 -O3 tools/handshake_probe.hip -o /tmp/handshake_probe 2>/dev/null
timeout 300 /tmp/handshake_probe > gpurun_out/r06/handshake_probe.txt 2>&1; cat gpurun_out/r06/handshake_probe.txt
bash tools/prof_exact.sh 44 > gpurun_out/r06/prof_exact44.log 2>&1; tail -16 gpurun_out/r06/prof_exact44.log | cut -c1-180
cp gpurun_out/prof_exact/kernel_stats.csv gpurun_out/r06/exact44_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06/prof_step
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 2 --cpu-images 0 --no-extras --fast-steps 0 --profile none --profile-steps 1 > $OUT/trace.log 2>&1
cd $GRAFT_REPO_ROOT
STATS=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $STATS gpurun_out/r06/step_kernel_stats.csv
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/r06/step_kernel_stats.csv')))
for r in rows[:45]:
    print(f"{r['Name'][:110]:110s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
PY
rm -rf $OUT/trace
