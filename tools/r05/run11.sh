#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_precise.py -q -m gpu -s 2>&1 | grep -i "attention_f32 vs\|vit2:\|vit24:\|passed\|failed\|Error\|assert" | head -20
rm -f gpurun_out/r05/exact_small_batches_x3attn_3w.txt
for n in 4 8 16 52; do timeout 200 python tools/exact_prof.py $n 3 2>/dev/null | tail -1 >> gpurun_out/r05/exact_small_batches_x3attn_3w.txt; done
cat gpurun_out/r05/exact_small_batches_x3attn_3w.txt
timeout 600 python bench.py --steps 12 --warmup 2 --no-extras --cpu-images 0 2>/dev/null > gpurun_out/r05/bench11.json; python - <<'P'
import json
d=json.loads(open('gpurun_out/r05/bench11.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','exact_cost_vs_fast','mfma_frac_end_to_end']}, (d.get('fast_mode') or {}).get('value'))
P
