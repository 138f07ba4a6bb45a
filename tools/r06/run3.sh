#!/bin/bash
# round 6, GPU session 3: refine_certainty checks EVERY alternative prototype / member (certainty + contract tests), quick bench,
# the 2-product tier probe (default + spread tower), the reference-module audit on 16 batches (tool check)
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_certainty.py tests/test_gpu_top1.py tests/test_gpu_requeue.py -q -m gpu 2>&1 | tail -12 > gpurun_out/r06/t_run3.txt
cat gpurun_out/r06/t_run3.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --cpu-images 0 > gpurun_out/r06/bench_quick3.json 2> gpurun_out/r06/bench_quick3.err
python - <<'PY'
import json
try:
    r = json.loads(open('gpurun_out/r06/bench_quick3.json').read().strip().splitlines()[-1])
    print({k: r.get(k) for k in ('value', 'ms_per_step', 'exact_cost_vs_fast', 'mfma_frac_end_to_end')})
    c = r['certainty']
    print(r.get('fast_mode', {}).get('value'), c['reencoded_panoramas_per_step'], c['reencoded_share'], c['uncertain_by_cause'], c['uncertain_after_step'])
    print(r.get('exact_pass_schedule'))
    print(r.get('per_rank_split_ms'))
    print(r.get('roofline_refine'))
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/r06/bench_quick3.err').read()[-3000:])
PY
timeout 600 python tools/mid_tier_probe.py 128 default > gpurun_out/r06/mid_tier_probe_default.txt 2>&1; cat gpurun_out/r06/mid_tier_probe_default.txt | tail -12
timeout 600 python tools/mid_tier_probe.py 128 spread > gpurun_out/r06/mid_tier_probe_spread.txt 2>&1; cat gpurun_out/r06/mid_tier_probe_spread.txt | tail -12
timeout 900 python tools/certainty_audit_ref.py 16 default > gpurun_out/r06/certainty_audit_ref_2048.txt 2>&1; tail -8 gpurun_out/r06/certainty_audit_ref_2048.txt
