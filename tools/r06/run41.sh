#!/bin/bash
# round 6, GPU session 41 (final tree): gemm_mid's operand DMAs without the arithmetic (tools/dma_rate_probe.hip); rocprofv3 kernel
# stats + FETCH / WRITE passes of the bench command; the GPU suite, the driver's bench command and smoke()
mkdir -p gpurun_out/r06
O=gpurun_out/r06
tools/bin/dma_rate_probe > $O/dma_rate_probe.txt 2>&1; cat $O/dma_rate_probe.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -30 > $O/gpu_suite_run41.txt; tail -4 $O/gpu_suite_run41.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_run41.json 2> $O/bench_line_run41.err
python - <<'PY'
import json
try:
    txt = open('gpurun_out/r06/bench_line_run41.json').read().strip().splitlines()
    print('stdout lines:', len(txt))
    r = json.loads(txt[-1])
    print({k: r.get(k) for k in ('value', 'ms_per_step', 'exact_cost_vs_fast', 'mfma_frac_end_to_end')}, r['fast_mode']['value'], r['roofline']['frac'])
    print({k: r['parity_vs_reference_module_gpu_fp32'].get(k) for k in ('n_panoramas', 'geocell_argmax_equal', 'refined_mismatch_unconditional', 'certain', 'error')})
    print({k: r['parity_vs_oracle_sample'].get(k) for k in ('geocell_argmax_equal', 'refined_mismatch_unconditional', 'flips')})
    print([ (c.get('workload','')[:40], c.get('value'), c.get('error')) for c in r.get('other_configs', [])])
    def walk(o, path=''):
        if isinstance(o, dict):
            for k, v in o.items():
                if k in ('error', 'trace') and v: print('ERROR', path + '/' + k, str(v)[:300])
                walk(v, path + '/' + k)
        elif isinstance(o, list):
            for i, v in enumerate(o): walk(v, path + f'[{i}]')
    walk(r)
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/r06/bench_line_run41.err').read()[-3000:])
PY
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | grep -v amdgpu.ids | tail -2
bash tools/prof_bench.sh r06c 2>&1 | tail -45
