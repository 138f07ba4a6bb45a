#!/bin/bash
# round 6, GPU session 46: the exact tier's small-batch routing with gemm_mid's new K-tile time (env PIGEON_EXACT_MID_US: 0.6 = the first
# session's constant, 0.44 = with the producer wave): the exact encoder at 4 .. 24 images, alternating
mkdir -p gpurun_out/r06
O=gpurun_out/r06
for rep in 1 2; do for c in 0.6 0.44 0.5; do for n in 4 8 12 16 20 24; do
  echo -n "PIGEON_EXACT_MID_US=$c "; PIGEON_EXACT_MID_US=$c timeout 120 python tools/exact_prof.py $n 5 2>&1 | grep -v amdgpu.ids | cut -c1-70
done; done; done | tee $O/exact_route_ab.txt
