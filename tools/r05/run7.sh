#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_top1.py -q -m gpu -s 2>&1 | grep -v "^Initialized\|^$" | tail -40 > gpurun_out/r05/t_contract7.txt
grep -h "error model check\|passed\|failed" gpurun_out/r05/t_contract7.txt
timeout 900 python bench.py > gpurun_out/r05/bench_line.json 2> gpurun_out/r05/bench_line.err; tail -c 200 gpurun_out/r05/bench_line.err
timeout 600 python bench.py --steps 20 --warmup 3 --no-extras --cpu-images 0 > gpurun_out/r05/bench_line_20_steps.json 2>/dev/null
python - <<'P'
import json
for f in ['bench_line','bench_line_20_steps']:
    try:
        d=json.loads(open(f'gpurun_out/r05/{f}.json').read().strip().splitlines()[-1])
        print(f,{k:d.get(k) for k in ['value','ms_per_step','certain_frac','exact_cost_vs_fast','mfma_frac_end_to_end']}, d['roofline']['frac'], d['roofline'].get('frac_rocprof'), (d.get('fast_mode') or {}).get('value'))
        print('  ', d['certainty']['reencoded_panoramas_per_step'], d['certainty']['uncertain_by_cause'], d['certainty']['calibration'])
    except Exception as e: print(f,'parse fail',e)
P
bash tools/prof_bench.sh r05 2>&1 | tail -12
bash tools/prof_bench_pmc.sh 2>&1 | grep "DURATION" | head -12
