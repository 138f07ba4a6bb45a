#!/bin/bash
# round 6, GPU session 29: the routing with its margins at 0.90 -- parity subset, in-situ A/B, serving request latency
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or vit_batch or routing" 2>&1 | grep -v amdgpu.ids | tail -3
bash tools/r06/run28.sh
timeout 600 python tools/serve_latency.py 2>&1 | grep -v amdgpu.ids | tail -1
