#!/bin/bash
# GPU call 2: the new tests (no -x), bench, exact tier timing at small batches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_certainty.py tests/test_gpu_top1.py tests/test_gpu_precise.py -q -m gpu -s 2>&1 | tail -220 > gpurun_out/r05/t_new2.txt
tail -12 gpurun_out/r05/t_new2.txt
for n in 4 8 16 32; do timeout 200 python tools/exact_prof.py $n 3 2>/dev/null | tail -1 >> gpurun_out/r05/exact_small_batches.txt; done
cat gpurun_out/r05/exact_small_batches.txt
timeout 600 python bench.py --steps 8 --warmup 2 > gpurun_out/r05/bench2.json 2> gpurun_out/r05/bench2.err; tail -c 400 gpurun_out/r05/bench2.err; python - <<'P'
import json
try:
    d=json.loads(open('gpurun_out/r05/bench2.json').read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ['value','ms_per_step','certain_frac','exact_cost_vs_fast','mfma_frac_end_to_end']})
    c=d.get('certainty',{}); print({k:c.get(k) for k in ['calibration','reencoded_panoramas_per_step','uncertain_after_step','uncertain_by_cause']})
    print(d.get('fast_mode')); print(d.get('per_rank_split_ms'))
    for k in ['parity_vs_oracle_sample','parity_vs_reference_module_gpu_fp32']:
        r=d.get(k,{})
        print(k,{kk:r.get(kk) for kk in ['embedding_rel_err','flips','geocell_argmax_equal','refined_mismatch_unconditional','certain','flips_among_certain','error']})
        print('   fast:', {kk:(r.get('fast_mode') or {}).get(kk) for kk in ['flips','refined_mismatch_unconditional']})
except Exception as e: print('parse fail',e)
P
