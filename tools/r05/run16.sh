#!/bin/bash
# round 5, GPU session 16: the raster knob of the 384 x 256 GEMM swept over every group size that divides a column-tile count
# (QKV 12 tiles, fc1 16, fc2 4; tools/gemm_ab.py: time per launch of 295 424 rows + output hashes), two rounds, alternating
mkdir -p gpurun_out/r05
out=gpurun_out/r05/raster_sweep.txt
: > $out
for rnd in 1 2; do
  for gn in 0 1 2 3 6 8 -1; do
    echo -n "gn=$gn  " >> $out
    PIGEON_GEMM_RASTER_GN=$gn timeout 120 python tools/gemm_ab.py 2>&1 | tail -1 >> $out
  done
done
cat $out
