#!/bin/bash
# Is a persistent GEMM's residual epilogue limited per CU or by something the CUs share (HBM, fabric)?  Cap the persistent grid
# (PIGEON_GEMM_BLOCKS) and read a tile's epilogue time from the in-kernel stamps (tools build): with 64 of 256 CUs running, each
# has 4x the HBM bandwidth to itself.  Usage (GPU box): bash tools/epi_cap_sweep.sh > gpurun_out/epi_cap_sweep.txt
cd "$(dirname "$0")/.."
export PIGEON_HIP_LIB=$PWD/pigeon_amd/libpigeon_hip_dev.so EPI_TIMELINE_SUMMARY_ONLY=1
for cap in 0 128 64 32 0; do
  if [ "$cap" = 0 ]; then unset PIGEON_GEMM_BLOCKS; else export PIGEON_GEMM_BLOCKS=$cap; fi
  timeout 300 python tools/epi_timeline.py out fc2 fc1 2>&1 | grep -E "SUMMARY|Error|error"
done
# the XCD start stagger with the stamps beside it: do the phases persist, and does a de-phased epilogue get shorter?
unset PIGEON_GEMM_BLOCKS
for f in 0.5 1.0; do
  echo "== PIGEON_GEMM_STAGGER=$f"
  PIGEON_GEMM_STAGGER=$f timeout 300 python tools/epi_timeline.py out fc2 2>&1 | grep -E "SUMMARY|start of block|Error|error"
done
