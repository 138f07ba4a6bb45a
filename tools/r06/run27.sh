#!/bin/bash
# round 6, GPU session 27: three-way routing of small / middle batches (384 x 256 / 256 x 256 / gemm_mid) -- parity, the sweep with the
# new picks, encoder latency 1 .. 64 images with the routing off / on, the quick bench form (must not move: 512 images are above the range)
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or vit_batch or routing" 2>&1 | grep -v amdgpu.ids | tail -4
timeout 600 python tools/gemm_mid_sweep.py --three 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/gemm_three_sweep_routed.txt; cat gpurun_out/r06/gemm_three_sweep_routed.txt
for v in 0 1 0 1; do
  PIGEON_GEMM_MID=$v timeout 600 python tools/latency_probe.py 1 2 4 8 12 16 24 32 48 64 2>&1 | grep -v amdgpu.ids | sed "s/^/PIGEON_GEMM_MID=$v /"
done | tee gpurun_out/r06/latency_route.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --cpu-images 0 2> /dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench quick', round(d['value'],1), round(d['ms_per_step'],2), 'fast', round(d['fast_mode']['value'],1), 'cost', round(d['exact_cost_vs_fast'],4), [f['ms'] for f in d['exact_pass_schedule']['this_rank']])"
