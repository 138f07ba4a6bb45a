#!/bin/bash
# round 6, GPU session 26: small-batch GEMM sweep with a third arm -- the 256 x 256 persistent kernel where variant 56 means 384 x 256
mkdir -p gpurun_out/r06
timeout 900 python tools/gemm_mid_sweep.py --three > gpurun_out/r06/gemm_three_sweep.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06/gemm_three_sweep.txt
