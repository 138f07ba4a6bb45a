#!/bin/bash
# round 6, GPU session 28: in-situ A/B of the routing -- PIGEON_GEMM_MID=2 (gemm_mid.hip the only alternative: the first half of the round)
# against =1 (the 256 x 256 kernel as a third candidate), encoder latency 1 .. 64 images, alternating on one box
mkdir -p gpurun_out/r06
for v in 2 1 2 1; do
  PIGEON_GEMM_MID=$v timeout 600 python tools/latency_probe.py 2 4 8 12 16 20 24 28 32 40 48 56 64 2>&1 | grep -v amdgpu.ids | sed "s/^/PIGEON_GEMM_MID=$v /"
done | tee gpurun_out/r06/latency_route_ab.txt | awk '{print $1, $4, $6}' | paste - - - - - - - - - - - - - | head -8
