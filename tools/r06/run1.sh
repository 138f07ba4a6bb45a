#!/bin/bash
# round 6, GPU session 1: the deferred exact tier's kernels + engine (new tests), the suites that touch certain_forward, a short bench
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_requeue.py -q -m gpu -x 2>&1 | tail -25 > gpurun_out/r06/t_requeue.txt
cat gpurun_out/r06/t_requeue.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r06/gpu_suite_first.txt
cat gpurun_out/r06/gpu_suite_first.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --cpu-images 0 > gpurun_out/r06/bench_quick.json 2> gpurun_out/r06/bench_quick.err
tail -c 3000 gpurun_out/r06/bench_quick.err
python - <<'PY'
import json
try:
    r = json.loads(open('gpurun_out/r06/bench_quick.json').read().strip().splitlines()[-1])
    print({k: r.get(k) for k in ('value', 'ms_per_step', 'exact_cost_vs_fast', 'mfma_frac_end_to_end')})
    print(r.get('fast_mode', {}).get('value'), r.get('certainty', {}).get('reencoded_panoramas_per_step'), r.get('certainty', {}).get('reencoded_share'))
    print(r.get('exact_pass_schedule'))
    print(r.get('per_rank_split_ms'))
except Exception as e:
    print('bench parse failed', e)
PY
