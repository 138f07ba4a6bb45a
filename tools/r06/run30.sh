#!/bin/bash
# round 6, GPU session 30: the routing model beyond its fitted range (64 .. 192 images), in situ, alternating
mkdir -p gpurun_out/r06
for v in 40000 200000 40000 200000; do
  PIGEON_GEMM_ROUTE_MAX_ROWS=$v timeout 600 python tools/latency_probe.py 64 72 80 96 112 128 160 192 2>&1 | grep -v amdgpu.ids | sed "s/^/MAX_ROWS=$v /"
done | tee gpurun_out/r06/latency_route_beyond.txt | awk '{print $1, $4, $6}' | paste - - - - - - - - | head -8
