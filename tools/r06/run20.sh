#!/bin/bash
# round 6, GPU session 20: the exact pass's activation splits inside their producers (attention_x3<true>, EPI_GELU_X3) -- parity, then
# time per image fused / unfused at the pass sizes that matter, then the quick bench form
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_precise.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -5
for f in 1 0 1 0; do
  echo "PIGEON_EXACT_FUSION=$f"
  PIGEON_EXACT_FUSION=$f timeout 600 python tools/exact_sweep.py 4 8 28 56 84 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r06/exact_fusion_ab.txt
for arm in "fuse1:1" "fuse0:0" "fuse1b:1"; do
  name=${arm%%:*}; v=${arm#*:}
  PIGEON_EXACT_FUSION=$v timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --cpu-images 0 2> gpurun_out/r06/fusion_${name}.err | tail -1 > gpurun_out/r06/fusion_${name}.json
  python - "$name" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r06/fusion_{name}.json"))
    sch = d.get("exact_pass_schedule", {})
    print(name, round(d["value"], 1), "img/s", round(d["ms_per_step"], 2), "ms", "passes", [(f.get("slots_run"), f.get("ms")) for f in sch.get("this_rank", [])][:8],
          "fast", round(d.get("fast_mode", {}).get("value", 0), 1), "cost", round(d.get("exact_cost_vs_fast", 0), 4), "parity", d.get("parity_vs_reference_module_gpu_fp32", {}).get("geocell_argmax_equal"))
except Exception as e:
    print(name, "failed", e)
PY
done
