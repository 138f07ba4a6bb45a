#!/bin/bash
# round 6, GPU session 43: gemm_mid.hip with a fifth wave that only issues the operand DMAs (build variant -DPG_MID_PRODUCER=1) against the
# product's form: bits, the sweep's gemm_mid column, encoder latency, a serving request, the exact encoder at small batches
mkdir -p gpurun_out/r06
O=gpurun_out/r06
for v in _prod ""; do
  export PIGEON_HIP_LIB=$PWD/pigeon_amd/libpigeon_hip$v.so
  echo "== libpigeon_hip$v.so"
  timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "gemm_mid or tail_split or small_batch_routing or batch_invariance" 2>&1 | tail -3
  timeout 400 python tools/gemm_mid_sweep.py > $O/gemm_mid_sweep_producer$v.txt 2>&1; grep -v amdgpu.ids $O/gemm_mid_sweep_producer$v.txt | cut -c1-260
  timeout 200 python tools/latency_probe.py 1 4 8 16 28 2>&1 | grep -v amdgpu.ids | cut -c1-90 | tee $O/latency_producer$v.txt
done
