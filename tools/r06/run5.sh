#!/bin/bash
# round 6, GPU session 5: the driver's bench command on the current tree (all extras: spread-tower leg, parity legs, CPU baseline),
# the reference-module audit on 64 batches (8192 panoramas) of the default tower and 32 of the spread tower, full GPU suite
mkdir -p gpurun_out/r06
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_line_a.json 2> gpurun_out/r06/bench_line_a.err
python - <<'PY'
import json
try:
    r = json.loads(open('gpurun_out/r06/bench_line_a.json').read().strip().splitlines()[-1])
    print({k: r.get(k) for k in ('value', 'ms_per_step', 'exact_cost_vs_fast', 'mfma_frac_end_to_end')})
    c = r['certainty']
    print(r.get('fast_mode', {}).get('value'), c['reencoded_share'], c['uncertain_by_cause'], c['uncertain_after_step'])
    print('oracle sample:', {k: r['parity_vs_oracle_sample'].get(k) for k in ('geocell_argmax_equal', 'refined_mismatch_unconditional', 'embedding_rel_err', 'flips')})
    p = r['parity_vs_reference_module_gpu_fp32']
    print('ref module:', {k: p.get(k) for k in ('n_panoramas', 'geocell_argmax_equal', 'refined_mismatch_unconditional', 'embedding_rel_err', 'flips', 'certain', 'flips_among_certain', 'error')})
    print('fast mode vs ref:', {k: p.get('fast_mode', {}).get(k) for k in ('flips', 'refined_mismatch_unconditional')})
    for o in r['other_configs']:
        print({k: (v if not isinstance(v, (dict, list)) else '...') for k, v in o.items()})
    print(r['roofline'])
    print(r['cpu_baseline'].get('value'), r['h2d_inclusive'].get('value'))
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/r06/bench_line_a.err').read()[-3000:])
PY
timeout 1500 python tools/certainty_audit_ref.py 64 default > gpurun_out/r06/certainty_audit_ref_8192.txt 2>&1; tail -7 gpurun_out/r06/certainty_audit_ref_8192.txt
timeout 1200 python tools/certainty_audit_ref.py 32 spread > gpurun_out/r06/certainty_audit_ref_spread_4096.txt 2>&1; tail -7 gpurun_out/r06/certainty_audit_ref_spread_4096.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r06/gpu_suite_2.txt; cat gpurun_out/r06/gpu_suite_2.txt
