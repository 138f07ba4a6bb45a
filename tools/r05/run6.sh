#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_precise.py tests/test_gpu_entrypoints.py::test_run_py_evaluate_exact_top1 tests/test_gpu_parity.py::test_full_size_step_properties -q -m gpu 2>&1 | tail -8 > gpurun_out/r05/t_fix6.txt
tail -4 gpurun_out/r05/t_fix6.txt
timeout 900 python bench.py > gpurun_out/r05/bench_line.json 2> gpurun_out/r05/bench_line.err; tail -c 300 gpurun_out/r05/bench_line.err
timeout 600 python bench.py --steps 20 --warmup 3 --no-extras --cpu-images 0 > gpurun_out/r05/bench_line_20_steps.json 2>/dev/null
python - <<'P'
import json
for f in ['bench_line','bench_line_20_steps']:
    try:
        d=json.loads(open(f'gpurun_out/r05/{f}.json').read().strip().splitlines()[-1])
        print(f,{k:d.get(k) for k in ['value','ms_per_step','certain_frac','exact_cost_vs_fast','mfma_frac_end_to_end']}, d['roofline']['frac'], (d.get('fast_mode') or {}).get('value'))
        print('  ', d['certainty']['reencoded_panoramas_per_step'], d['certainty']['uncertain_by_cause'])
    except Exception as e: print(f,'parse fail',e)
P
bash tools/prof_bench.sh r05 2>&1 | tail -30
# raster A/B on the counters: FETCH_SIZE with every XCD round walking all N tiles
cd /tmp
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/r05/pmc_fetch_gn_all -- python $GRAFT_REPO_ROOT/bench.py --fast --steps 1 --warmup 1 --cpu-images 0 --no-extras --fast-steps 0 --profile none --profile-steps 1 --raster-gn -1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/r05/pmc_fetch_gn_all "" > gpurun_out/r05/pmc_fetch_gn_all_summary.txt 2>&1
grep -A3 "gemm_pp6_kernel<T_F16, [67]" gpurun_out/r05/pmc_fetch_gn_all_summary.txt | head -20
find gpurun_out/r05 -name "*.csv" -size +5M -delete
bash tools/prof_bench_pmc.sh 2>&1 | tail -40
