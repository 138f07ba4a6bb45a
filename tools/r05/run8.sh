#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 600 python tools/dump_fast_exact.py 4 default 2>&1 | tail -1
timeout 600 python tools/dump_fast_exact.py 2 spread 2>&1 | tail -1
cd /tmp
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/r05/pmc_fetch_gn_all -- python $GRAFT_REPO_ROOT/bench.py --fast --steps 1 --warmup 1 --cpu-images 0 --no-extras --fast-steps 0 --profile none --profile-steps 1 --raster-gn -1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/r05/pmc_fetch_gn_all "" > gpurun_out/r05/pmc_fetch_gn_all_summary.txt 2>&1
grep -A1 "KERNEL.*gemm_pp6_kernel<T_F16, [67]" gpurun_out/r05/pmc_fetch_gn_all_summary.txt | head -8
find gpurun_out/r05 -name "*.csv" -size +5M -delete
timeout 600 python bench.py --weights spread --no-refine --steps 8 --warmup 2 --cpu-images 0 > gpurun_out/r05/bench_line_spread_weights.json 2>/dev/null
python - <<'P'
import json
try:
    d=json.loads(open('gpurun_out/r05/bench_line_spread_weights.json').read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ['value','ms_per_step','certain_frac','exact_cost_vs_fast']}, (d.get('fast_mode') or {}).get('value'))
    print(d['certainty']['reencoded_panoramas_per_step'], d['certainty']['uncertain_by_cause'], d['certainty']['calibration'])
    r=d.get('parity_vs_reference_module_gpu_fp32',{}); print({k:r.get(k) for k in ['embedding_rel_err','embedding_rel_err_worst_image','flips','geocell_argmax_equal','certain','error']}, (r.get('fast_mode') or {}).get('flips'))
except Exception as e: print('parse fail', e)
P
