#!/bin/bash
# round 6, GPU session 39: gemm_mid.hip with the mainloop software-pipelined inside the wave (two fragment sets, DMAs three K tiles ahead,
# MFMAs in VGPR form): bits (all seven epilogues, tails, routing), the three-kernel sweep, encoder latency, the exact encoder at small batches
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gemm_mid or tail_split or small_batch_routing or batch_invariance" 2>&1 | tail -6 > $O/t_run39.txt; cat $O/t_run39.txt
timeout 600 python tools/gemm_mid_sweep.py --three > $O/gemm_three_sweep_pipelined.txt 2>&1; grep -v amdgpu.ids $O/gemm_three_sweep_pipelined.txt
timeout 300 python tools/latency_probe.py 1 2 4 8 12 16 20 24 28 32 48 64 > $O/latency_pipelined.txt 2>&1; grep -v amdgpu.ids $O/latency_pipelined.txt
PIGEON_GEMM_MID=2 timeout 300 python tools/latency_probe.py 1 4 8 16 28 > $O/latency_pipelined_midonly.txt 2>&1; grep -v amdgpu.ids $O/latency_pipelined_midonly.txt
for n in 4 8 16 28; do timeout 200 python tools/exact_prof.py $n 5 2>&1 | grep -v amdgpu.ids | cut -c1-200; done > $O/exact_small_pipelined.txt; cat $O/exact_small_pipelined.txt
timeout 300 python tools/serve_latency.py > $O/serve_latency_pipelined.txt 2>&1; grep -v amdgpu.ids $O/serve_latency_pipelined.txt | tail -4
