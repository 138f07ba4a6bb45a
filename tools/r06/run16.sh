#!/bin/bash
# round 6, GPU session 16: exact encoder time per image against the batch size (tile rounds of the N = 1024 GEMMs)
mkdir -p gpurun_out/r06
timeout 900 python tools/exact_sweep.py > gpurun_out/r06/exact_sweep.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06/exact_sweep.txt | tail -24
