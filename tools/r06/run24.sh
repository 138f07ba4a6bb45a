#!/bin/bash
# round 6, GPU session 24: pinned staging in gpu_preprocess's PIL path -- parity tests, then the serving request's latency
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_preprocess.py tests/test_gpu_entrypoints.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -4
timeout 600 python tools/serve_latency.py > gpurun_out/r06/serve_latency_staged.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06/serve_latency_staged.txt | tail -3
timeout 600 python tools/serve_latency.py --profile 2>&1 | grep -v amdgpu.ids | grep -i "gpu_preprocess\|listcomp\|tobytes\|convert\|stream time\|predict_panorama\|synchronize" | cut -c1-180
