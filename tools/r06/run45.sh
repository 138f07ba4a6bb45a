#!/bin/bash
# round 6, GPU session 45: the tree with gemm_mid.hip's producer wave and the refitted routing -- GPU suite, encoder latency 1 .. 64 images,
# a serving request, the driver's bench command, smoke()
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -30 > $O/gpu_suite_run45.txt; tail -4 $O/gpu_suite_run45.txt
timeout 300 python tools/latency_probe.py 1 2 4 8 12 16 20 24 28 32 48 64 2>&1 | grep -v amdgpu.ids | cut -c1-90 | tee $O/latency_run45.txt
timeout 300 python tools/serve_latency.py > $O/serve_latency_run45.txt 2>&1; grep -v amdgpu.ids $O/serve_latency_run45.txt | tail -2
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_run45.json 2> $O/bench_line_run45.err
python - <<'PY'
import json
try:
    txt = open('gpurun_out/r06/bench_line_run45.json').read().strip().splitlines()
    print('stdout lines:', len(txt))
    r = json.loads(txt[-1])
    print({k: r.get(k) for k in ('value', 'ms_per_step', 'exact_cost_vs_fast', 'mfma_frac_end_to_end')}, r['fast_mode']['value'], r['roofline']['frac'], r['per_rank_split_ms']['compute'], r['per_rank_split_ms']['exact_passes_per_step'])
    print({k: r['parity_vs_reference_module_gpu_fp32'].get(k) for k in ('n_panoramas', 'geocell_argmax_equal', 'refined_mismatch_unconditional', 'certain', 'error')})
    print({k: r['parity_vs_oracle_sample'].get(k) for k in ('geocell_argmax_equal', 'refined_mismatch_unconditional', 'flips')})
    print([f["ms"] for f in r["exact_pass_schedule"]["this_rank"]] if "exact_pass_schedule" in r else None)
    def walk(o, path=''):
        if isinstance(o, dict):
            for k, v in o.items():
                if k in ('error', 'trace') and v: print('ERROR', path + '/' + k, str(v)[:300])
                walk(v, path + '/' + k)
        elif isinstance(o, list):
            for i, v in enumerate(o): walk(v, path + f'[{i}]')
    walk(r)
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/r06/bench_line_run45.err').read()[-3000:])
PY
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | grep -v amdgpu.ids | tail -2
