#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_precise.py tests/test_gpu_certainty.py tests/test_gpu_top1.py -q -m gpu -s 2>&1 | tail -120 > gpurun_out/r05/t_new3.txt
tail -8 gpurun_out/r05/t_new3.txt; grep -h "all_heads\|trained_like" gpurun_out/r05/t_new3.txt | head -4
rm -f gpurun_out/r05/exact_small_batches_ksplit.txt
for n in 4 8 16 32 52; do timeout 200 python tools/exact_prof.py $n 3 2>/dev/null | tail -1 >> gpurun_out/r05/exact_small_batches_ksplit.txt; done
cat gpurun_out/r05/exact_small_batches_ksplit.txt
timeout 600 python bench.py --steps 12 --warmup 2 --no-extras --cpu-images 0 > gpurun_out/r05/bench3.json 2> gpurun_out/r05/bench3.err; tail -c 300 gpurun_out/r05/bench3.err; python - <<'P'
import json
try:
    d=json.loads(open('gpurun_out/r05/bench3.json').read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ['value','ms_per_step','certain_frac','exact_cost_vs_fast','mfma_frac_end_to_end']})
    c=d.get('certainty',{}); print({k:c.get(k) for k in ['reencoded_panoramas_per_step','uncertain_after_step','uncertain_by_cause']})
    print(d.get('fast_mode'))
except Exception as e: print('parse fail',e)
P
