#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_precise.py -q -m gpu -s -k "attention_f32 or abi" 2>&1 | grep -i "attention_f32 \[\|passed\|failed" | head
timeout 900 python bench.py > gpurun_out/r05/bench_line.json 2> gpurun_out/r05/bench_line.err
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r05/bench_driver_shape.json 2>/dev/null
python - <<'P'
import json
for f in ['bench_line','bench_driver_shape']:
    d=json.loads(open(f'gpurun_out/r05/{f}.json').read().strip().splitlines()[-1])
    print(f,{k:d.get(k) for k in ['value','ms_per_step','exact_cost_vs_fast','mfma_frac_end_to_end']}, d['roofline']['frac'], d['roofline'].get('frac_rocprof'), (d.get('fast_mode') or {}).get('value'))
    r=d['roofline_refine']; print('  refine', {k:r[k] for k in ['achieved','frac','avg_ms']})
    for k in ['parity_vs_oracle_sample','parity_vs_reference_module_gpu_fp32']:
        r=d.get(k,{}); print('  ',k,{kk:r.get(kk) for kk in ['flips','geocell_argmax_equal','refined_mismatch_unconditional','error']})
P
