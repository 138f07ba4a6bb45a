#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_precise.py tests/test_gpu_certainty.py tests/test_gpu_top1.py -q -m gpu -s 2>&1 | tail -60 > gpurun_out/r05/t_new4.txt
tail -6 gpurun_out/r05/t_new4.txt; grep -h "all_heads\|trained_like\|product on" gpurun_out/r05/t_new4.txt | head -6
rm -f gpurun_out/r05/exact_small_batches_parts.txt
for n in 4 8 16 32 52; do timeout 200 python tools/exact_prof.py $n 3 2>/dev/null | tail -1 >> gpurun_out/r05/exact_small_batches_parts.txt; done
cat gpurun_out/r05/exact_small_batches_parts.txt
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05/exact8_trace -- python $GRAFT_REPO_ROOT/tools/exact_prof.py 8 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
S=$(find gpurun_out/r05/exact8_trace -name "*kernel_stats.csv" | head -1); cp $S gpurun_out/r05/exact8_kernel_stats.csv; head -16 gpurun_out/r05/exact8_kernel_stats.csv | cut -c1-150
find gpurun_out/r05/exact8_trace -name "*.csv" -size +5M -delete
timeout 600 python bench.py --steps 12 --warmup 2 --no-extras --cpu-images 0 > gpurun_out/r05/bench4.json 2> gpurun_out/r05/bench4.err; tail -c 300 gpurun_out/r05/bench4.err; python - <<'P'
import json
try:
    d=json.loads(open('gpurun_out/r05/bench4.json').read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ['value','ms_per_step','certain_frac','exact_cost_vs_fast','mfma_frac_end_to_end']})
    c=d.get('certainty',{}); print({k:c.get(k) for k in ['reencoded_panoramas_per_step','uncertain_after_step','uncertain_by_cause']})
    print(d.get('fast_mode'))
except Exception as e: print('parse fail',e)
P
