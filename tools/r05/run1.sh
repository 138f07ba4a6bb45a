#!/bin/bash
# GPU call 1: new tests first, smoke, a default bench line, raster A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_certainty.py tests/test_gpu_top1.py tests/test_gpu_precise.py -q -m gpu -x -s 2>&1 | tail -150 > gpurun_out/r05/t_new.txt
tail -5 gpurun_out/r05/t_new.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05/smoke.txt 2>&1; tail -4 gpurun_out/r05/smoke.txt
timeout 600 python bench.py --steps 8 --warmup 2 > gpurun_out/r05/bench1.json 2> gpurun_out/r05/bench1.err; tail -c 600 gpurun_out/r05/bench1.err; python - <<'P'
import json
try:
    d=json.loads(open('gpurun_out/r05/bench1.json').read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ['value','ms_per_step','certain_frac','exact_cost_vs_fast','mfma_frac_end_to_end']})
    print(d.get('certainty'))
    print(d.get('fast_mode'))
    for k in ['parity_vs_oracle_sample','parity_vs_reference_module_gpu_fp32']:
        r=d.get(k,{})
        print(k,{kk:r.get(kk) for kk in ['embedding_rel_err','flips','geocell_argmax_equal','refined_mismatch_unconditional','certain','flips_among_certain','fast_mode','error']})
except Exception as e: print('parse fail',e)
P
for i in 1 2; do
for gn in 0 -1; do
python bench.py --fast --no-extras --cpu-images 0 --fast-steps 0 --steps 10 --warmup 3 --raster-gn $gn 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('gn=$gn', round(d['value'],1), {n:round(k[n]['avg_ms'],4) for n in ('gemm_qkv','gemm_fc1','gemm_fc2','gemm_out','attention')})" >> gpurun_out/r05/raster_ab.txt
done; done
cat gpurun_out/r05/raster_ab.txt
