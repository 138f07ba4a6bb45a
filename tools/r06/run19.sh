#!/bin/bash
# round 6, GPU session 19: numerics probe -- the exact tier's correction products in block-scaled fp8 (emulated in float64)
mkdir -p gpurun_out/r06
timeout 1500 python tools/fp8_corr_probe.py 64 default > gpurun_out/r06/fp8_corr_probe_default.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06/fp8_corr_probe_default.txt | tail -12
timeout 1500 python tools/fp8_corr_probe.py 64 spread > gpurun_out/r06/fp8_corr_probe_spread.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06/fp8_corr_probe_spread.txt | tail -12
