#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_precise.py -q -m gpu -s 2>&1 | grep -i "attention_f32 vs\|vit2:\|vit24:\|passed\|failed\|Error\|assert" | head -20
rm -f gpurun_out/r05/exact_small_batches_x3attn.txt
for n in 4 8 16 52; do timeout 200 python tools/exact_prof.py $n 3 2>/dev/null | tail -1 >> gpurun_out/r05/exact_small_batches_x3attn.txt; done
cat gpurun_out/r05/exact_small_batches_x3attn.txt
PIGEON_EXACT_ATTN=f32 timeout 200 python tools/exact_prof.py 8 3 2>/dev/null | tail -1
