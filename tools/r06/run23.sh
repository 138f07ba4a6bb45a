#!/bin/bash
# round 6, GPU session 23: where a serving request's 1.9 ms outside the encoder go (host profile)
mkdir -p gpurun_out/r06
timeout 900 python tools/serve_latency.py --profile > gpurun_out/r06/serve_profile.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06/serve_profile.txt | cut -c1-200 | head -80
