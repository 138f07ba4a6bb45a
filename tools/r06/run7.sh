#!/bin/bash
# round 6, GPU session 7: gemm_mid cost model re-fitted (0.6 us per K tile), the exact tier's small batches through it, clean sweep
# (persistent kernels forced vs mid forced vs what the model picks), latency on / off, precise + parity suites, the 8192 audit with details
mkdir -p gpurun_out/r06
rm -f gpurun_out/r06/latency_mid.txt gpurun_out/r06/exact_small_mid.txt
timeout 600 python tools/gemm_mid_sweep.py > gpurun_out/r06/gemm_mid_sweep.txt 2>&1; cat gpurun_out/r06/gemm_mid_sweep.txt
for on in 0 1; do
  echo "PIGEON_GEMM_MID=$on" >> gpurun_out/r06/latency_mid.txt
  PIGEON_GEMM_MID=$on timeout 300 python tools/latency_probe.py 1 4 8 12 16 24 32 64 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06/latency_mid.txt
done
cat gpurun_out/r06/latency_mid.txt
for on in 0 1; do
  echo "PIGEON_GEMM_MID=$on" >> gpurun_out/r06/exact_small_mid.txt
  for n in 4 8 16 44; do PIGEON_GEMM_MID=$on timeout 200 python tools/exact_prof.py $n 5 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06/exact_small_mid.txt; done
done
cat gpurun_out/r06/exact_small_mid.txt
timeout 1500 python -m pytest tests/test_gpu_precise.py tests/test_gpu_parity.py tests/test_gpu_requeue.py -q -m gpu 2>&1 | tail -8 > gpurun_out/r06/t_run7.txt; cat gpurun_out/r06/t_run7.txt
timeout 1500 python tools/certainty_audit_ref.py 64 default > gpurun_out/r06/certainty_audit_ref_8192.txt 2>&1; tail -8 gpurun_out/r06/certainty_audit_ref_8192.txt
