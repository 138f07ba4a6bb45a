#!/bin/bash
# round 6, GPU session 14: token_mean with 16 loads in flight (bit-identical order): parity subset + latency
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu 2>&1 | tail -4 > gpurun_out/r06/t_run14.txt; cat gpurun_out/r06/t_run14.txt
timeout 300 python tools/latency_probe.py 1 4 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/latency_run14.txt; cat gpurun_out/r06/latency_run14.txt
