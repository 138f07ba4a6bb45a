#!/bin/bash
# round 6, GPU session 6: the 128 x 128 small-batch GEMM kernel (gemm_mid.hip): bit-identity with the persistent kernels, per-shape sweep
# against them, encoder latency at 1 .. 64 images with it on / off, the batch-invariance tests
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "gemm_mid or gemm_tail or batch_invariance or chunking" 2>&1 | tail -12 > gpurun_out/r06/t_mid.txt; cat gpurun_out/r06/t_mid.txt
timeout 600 python tools/gemm_mid_sweep.py > gpurun_out/r06/gemm_mid_sweep.txt 2>&1; cat gpurun_out/r06/gemm_mid_sweep.txt
for on in 0 1; do
  echo "PIGEON_GEMM_MID=$on" >> gpurun_out/r06/latency_mid.txt
  PIGEON_GEMM_MID=$on timeout 300 python tools/latency_probe.py 1 4 8 16 32 64 >> gpurun_out/r06/latency_mid.txt 2>&1
done
cat gpurun_out/r06/latency_mid.txt
for on in 0 1; do
  echo "PIGEON_GEMM_MID=$on" >> gpurun_out/r06/exact_small_mid.txt
  for n in 4 8 16; do PIGEON_GEMM_MID=$on timeout 200 python tools/exact_prof.py $n 5 >> gpurun_out/r06/exact_small_mid.txt 2>&1; done
done
cat gpurun_out/r06/exact_small_mid.txt
