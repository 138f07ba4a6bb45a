#!/bin/bash
# round 6, GPU session 36: the driver's bench command (final bench.py: the fast-mode leg times 10 batches), the GPU suite and smoke() on the final tree
mkdir -p gpurun_out/r06
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_line_run36.json 2> gpurun_out/r06/bench_line_run36.err
python - <<'PY'
import json
try:
    r = json.loads(open('gpurun_out/r06/bench_line_run36.json').read().strip().splitlines()[-1])
    print({k: r.get(k) for k in ('value', 'ms_per_step', 'exact_cost_vs_fast', 'mfma_frac_end_to_end')}, r['fast_mode']['value'], r['roofline']['frac'])
    print(json.dumps(r['h2d_inclusive'])[:900])
    print({k: r['parity_vs_reference_module_gpu_fp32'].get(k) for k in ('n_panoramas', 'geocell_argmax_equal', 'refined_mismatch_unconditional', 'certain', 'error')})
    print({k: r['parity_vs_oracle_sample'].get(k) for k in ('geocell_argmax_equal', 'refined_mismatch_unconditional', 'flips')})
    def walk(o, path=''):
        if isinstance(o, dict):
            for k, v in o.items():
                if k in ('error', 'trace') and v: print('ERROR', path + '/' + k, str(v)[:300])
                walk(v, path + '/' + k)
        elif isinstance(o, list):
            for i, v in enumerate(o): walk(v, path + f'[{i}]')
    walk(r)
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/r06/bench_line_run36.err').read()[-3000:])
PY
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r06/gpu_suite_run36.txt; cat gpurun_out/r06/gpu_suite_run36.txt
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | grep -v amdgpu.ids | tail -2
