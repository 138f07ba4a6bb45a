#!/bin/bash
# round 6, GPU session 38: the GPU suite, the driver's bench command and smoke() on the tree with the de-biased embeddings (SuperGuessr and CLIPEmbedding)
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -40 > $O/gpu_suite_run38.txt; tail -12 $O/gpu_suite_run38.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_run38.json 2> $O/bench_line_run38.err
python - <<'PY'
import json
try:
    txt = open('gpurun_out/r06/bench_line_run38.json').read().strip().splitlines()
    print('stdout lines:', len(txt))
    r = json.loads(txt[-1])
    print({k: r.get(k) for k in ('value', 'ms_per_step', 'exact_cost_vs_fast', 'mfma_frac_end_to_end')}, r['fast_mode']['value'], r['roofline']['frac'])
    print(json.dumps(r['h2d_inclusive'])[:600])
    print({k: r['parity_vs_reference_module_gpu_fp32'].get(k) for k in ('n_panoramas', 'geocell_argmax_equal', 'refined_mismatch_unconditional', 'certain', 'error')})
    print({k: r['parity_vs_oracle_sample'].get(k) for k in ('geocell_argmax_equal', 'refined_mismatch_unconditional', 'flips')})
    print(json.dumps(r['certainty'])[-900:])
    def walk(o, path=''):
        if isinstance(o, dict):
            for k, v in o.items():
                if k in ('error', 'trace') and v: print('ERROR', path + '/' + k, str(v)[:300])
                walk(v, path + '/' + k)
        elif isinstance(o, list):
            for i, v in enumerate(o): walk(v, path + f'[{i}]')
    walk(r)
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/r06/bench_line_run38.err').read()[-3000:])
PY
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | grep -v amdgpu.ids | tail -3
