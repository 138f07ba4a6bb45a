#!/bin/bash
# round 6, GPU session 32: the engine after the median-trigger rule (one rank: unchanged by construction) -- engine tests, quick bench form
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_requeue.py tests/test_gpu_entrypoints.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -3
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --cpu-images 0 2> /dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench quick', round(d['value'],1), round(d['ms_per_step'],2), 'fast', round(d['fast_mode']['value'],1), 'cost', round(d['exact_cost_vs_fast'],4), [(f['slots_run'], f['ms']) for f in d['exact_pass_schedule']['this_rank']])"
