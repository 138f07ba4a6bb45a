#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/r05/t_full_final.txt
tail -3 gpurun_out/r05/t_full_final.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05/smoke.txt 2>&1; tail -2 gpurun_out/r05/smoke.txt
timeout 900 python bench.py > gpurun_out/r05/bench_line.json 2> gpurun_out/r05/bench_line.err
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r05/bench_driver_shape.json 2>/dev/null
python - <<'P'
import json
for f in ['bench_line','bench_driver_shape']:
    try:
        d=json.loads(open(f'gpurun_out/r05/{f}.json').read().strip().splitlines()[-1])
        print(f,{k:d.get(k) for k in ['value','ms_per_step','exact_cost_vs_fast','mfma_frac_end_to_end']}, d['roofline']['frac'], d['roofline'].get('frac_rocprof'), (d.get('fast_mode') or {}).get('value'))
        print('  ', d['certainty']['reencoded_panoramas_per_step'], d['certainty']['uncertain_by_cause'])
        for k in ['parity_vs_oracle_sample','parity_vs_reference_module_gpu_fp32']:
            r=d.get(k,{}); print('  ',k,{kk:r.get(kk) for kk in ['embedding_rel_err','flips','geocell_argmax_equal','refined_mismatch_unconditional','certain','error']}, {kk:(r.get('fast_mode') or {}).get(kk) for kk in ['flips','refined_mismatch_unconditional']})
        print('  cpu', (d.get('cpu_baseline') or {}).get('value'), 'h2d', (d.get('h2d_inclusive') or {}).get('value'), 'refine', (d.get('roofline_refine') or {}).get('frac'))
    except Exception as e: print(f,'parse fail',e)
P
bash tools/prof_bench.sh r05 2>&1 | tail -3
bash tools/prof_bench_pmc.sh 2>&1 | grep "DURATION" | head -8
