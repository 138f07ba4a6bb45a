#!/bin/bash
# round 6, GPU session 44: the three-kernel sweep with gemm_mid.hip's producer wave (now the default build) -- what pg_gemm_launch's routing
# model is refitted to; a serving request; the exact encoder at small batches
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 600 python tools/gemm_mid_sweep.py --three > $O/gemm_three_sweep_producer.txt 2>&1; grep -v amdgpu.ids $O/gemm_three_sweep_producer.txt | cut -c1-330
timeout 300 python tools/serve_latency.py > $O/serve_latency_producer.txt 2>&1; grep -v amdgpu.ids $O/serve_latency_producer.txt | tail -2
for n in 4 8 16; do timeout 200 python tools/exact_prof.py $n 5 2>&1 | grep -v amdgpu.ids | cut -c1-200; done > $O/exact_small_producer.txt; cat $O/exact_small_producer.txt
