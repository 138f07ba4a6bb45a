#!/bin/bash
# round 6, GPU session 9: equal chunks of <= 128 images in the exact pass (precise tests + a 130-image pass), serving latency per request
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_precise.py tests/test_gpu_requeue.py tests/test_gpu_top1.py -q -m gpu 2>&1 | tail -5 > gpurun_out/r06/t_run9.txt; cat gpurun_out/r06/t_run9.txt
for n in 68 130; do timeout 300 python tools/exact_prof.py $n 3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06/exact_chunks.txt; done; cat gpurun_out/r06/exact_chunks.txt
for on in 0 1; do PIGEON_GEMM_MID=$on timeout 600 python tools/serve_latency.py 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r06/serve_latency.txt; done; cat gpurun_out/r06/serve_latency.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --cpu-images 0 > gpurun_out/r06/bench_quick9.json 2> gpurun_out/r06/bench_quick9.err
python -c "
import json; r=json.loads(open('gpurun_out/r06/bench_quick9.json').read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['exact_cost_vs_fast'], r['exact_pass_schedule']['this_rank'])"
