#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r05/t_full_final.txt
tail -6 gpurun_out/r05/t_full_final.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05/smoke.txt 2>&1; tail -3 gpurun_out/r05/smoke.txt
timeout 600 python bench.py --steps 20 --warmup 3 2>/dev/null > gpurun_out/r05/bench_driver_shape.json; python - <<'P'
import json
d=json.loads(open('gpurun_out/r05/bench_driver_shape.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','exact_cost_vs_fast','mfma_frac_end_to_end']}, d['roofline']['frac'], d['roofline'].get('frac_rocprof'))
for k in ['parity_vs_oracle_sample','parity_vs_reference_module_gpu_fp32']:
    r=d.get(k,{}); print(k,{kk:r.get(kk) for kk in ['flips','geocell_argmax_equal','refined_mismatch_unconditional','certain','error']})
print(d.get('cpu_baseline',{}).get('value'), d.get('rccl'))
P
