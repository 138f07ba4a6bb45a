#!/bin/bash
# round 6, GPU session 18: the tail rows of the K = 1024 GEMMs through gemm_mid.hip (QKV and out-projection are now cut as well) -- parity,
# then an A/B of the quick bench form on one box (PIGEON_GEMM_MID=0 = the old routing: fc1 / fc2 tails through gemm_tail.hip, no cut elsewhere)
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or vit_batch" 2>&1 | grep -v amdgpu.ids | tail -4
for arm in "mid1:1" "mid0:0" "mid1b:1" "mid0b:0"; do
  name=${arm%%:*}; v=${arm#*:}
  PIGEON_GEMM_MID=$v timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --cpu-images 0 2> gpurun_out/r06/tailmid_${name}.err | tail -1 > gpurun_out/r06/tailmid_${name}.json
  python - "$name" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r06/tailmid_{name}.json"))
    print(name, round(d["value"], 1), "img/s", round(d["ms_per_step"], 2), "ms", "fast", round(d.get("fast_mode", {}).get("value", 0), 1), "cost", round(d.get("exact_cost_vs_fast", 0), 4),
          "roofline", d["roofline"].get("frac"), d["roofline"].get("kernel_ms", d["roofline"].get("avg_ms")))
except Exception as e:
    print(name, "failed", e)
PY
done
