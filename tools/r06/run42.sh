#!/bin/bash
# round 6, GPU session 42: gemm_mid.hip's ring depth (3 stages = the product; 4 and 5 as build variants: -DPG_MID_STAGES): bits, the
# sweep's gemm_mid column, encoder latency at 1 / 4 / 8 / 16 images
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gemm_mid or tail_split or small_batch_routing" 2>&1 | tail -3
for v in "" _s4 _s5; do
  export PIGEON_HIP_LIB=$PWD/pigeon_amd/libpigeon_hip$v.so
  echo "== libpigeon_hip$v.so"
  timeout 400 python tools/gemm_mid_sweep.py > $O/gemm_mid_sweep_stages$v.txt 2>&1; grep -v amdgpu.ids $O/gemm_mid_sweep_stages$v.txt | cut -c1-260
  timeout 200 python tools/latency_probe.py 1 4 8 16 2>&1 | grep -v amdgpu.ids | cut -c1-90 | tee $O/latency_stages$v.txt
done
