#!/bin/bash
# round 6, GPU session 8: the tree as committed -- whole GPU suite, smoke(), the driver's bench command, --fast / --no-refine sanity,
# rocprofv3 kernel stats + FETCH / WRITE / MFMA-busy passes of the bench command (profiles/r06)
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/r06/gpu_suite_final.txt; cat gpurun_out/r06/gpu_suite_final.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r06/smoke.txt; cat gpurun_out/r06/smoke.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_line.json 2> gpurun_out/r06/bench_line.err
python - <<'PY'
import json
try:
    r = json.loads(open('gpurun_out/r06/bench_line.json').read().strip().splitlines()[-1])
    print({k: r.get(k) for k in ('value', 'ms_per_step', 'exact_cost_vs_fast', 'mfma_frac_end_to_end')}, r['fast_mode']['value'], r['roofline']['frac'])
    print(r['certainty']['reencoded_share'], r['certainty']['uncertain_after_step'], r['exact_pass_schedule']['this_rank'])
    print({k: r['parity_vs_reference_module_gpu_fp32'].get(k) for k in ('n_panoramas', 'geocell_argmax_equal', 'refined_mismatch_unconditional', 'certain', 'error')})
    print([ (o.get('value'), o.get('exact_cost_vs_fast'), o.get('error')) for o in r['other_configs']])
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/r06/bench_line.err').read()[-3000:])
PY
timeout 300 python bench.py --steps 6 --warmup 2 --no-extras --cpu-images 0 --fast > gpurun_out/r06/bench_fast.json 2> gpurun_out/r06/bench_fast.err; python -c "
import json; r=json.loads(open('gpurun_out/r06/bench_fast.json').read().strip().splitlines()[-1]); print('--fast', r['value'], r.get('exact_mode',{}).get('value'))"
timeout 300 python bench.py --steps 6 --warmup 2 --no-extras --cpu-images 0 --no-refine > gpurun_out/r06/bench_norefine.json 2> gpurun_out/r06/bench_norefine.err; python -c "
import json; r=json.loads(open('gpurun_out/r06/bench_norefine.json').read().strip().splitlines()[-1]); print('--no-refine', r['value'], r['certainty']['reencoded_share'])"
bash tools/prof_bench.sh r06 > gpurun_out/r06/prof_bench.log 2>&1; tail -30 gpurun_out/r06/prof_bench.log | cut -c1-200
bash tools/prof_bench_pmc.sh > gpurun_out/r06/prof_bench_pmc.log 2>&1; tail -45 gpurun_out/r06/prof_bench_pmc.log | cut -c1-160
cp gpurun_out/prof_bench_pmc/summary.txt gpurun_out/r06/bench_pmc_mfma_busy.txt 2>/dev/null
cp gpurun_out/prof_bench_r06/kernel_stats.csv gpurun_out/r06/bench_kernel_stats.csv 2>/dev/null
cp gpurun_out/prof_bench_r06/traffic.json gpurun_out/r06/traffic.json 2>/dev/null
cp gpurun_out/prof_bench_r06/pmc_fetch_summary.txt gpurun_out/r06/bench_pmc_fetch_size.txt 2>/dev/null
cp gpurun_out/prof_bench_r06/pmc_write_summary.txt gpurun_out/r06/bench_pmc_write_size.txt 2>/dev/null
rm -rf gpurun_out/prof_bench_r06/trace gpurun_out/prof_bench_r06/pmc_fetch gpurun_out/prof_bench_r06/pmc_write gpurun_out/prof_bench_pmc/pmc
