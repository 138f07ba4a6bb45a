#!/bin/bash
# round 6, GPU session 17: exact passes in whole rounds of the CUs (pass quantum) -- A/B on one box, quick form of the bench command
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_gpu_requeue.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -3
for arm in "q7:--pass-quantum -1" "old:--pass-quantum 0 --min-flush 10" "q14:--pass-quantum 14 --min-flush 14" "q7b:--pass-quantum -1" "oldb:--pass-quantum 0 --min-flush 10"; do
  name=${arm%%:*}; flags=${arm#*:}
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --cpu-images 0 $flags 2> gpurun_out/r06/quantum_${name}.err | tail -1 > gpurun_out/r06/quantum_${name}.json
  python - "$name" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r06/quantum_{name}.json"))
    sch = d.get("exact_pass_schedule", {})
    print(name, round(d["value"], 1), "img/s", round(d["ms_per_step"], 2), "ms", "passes", [(f.get("slots_run"), f.get("ms")) for f in sch.get("this_rank", [])][:10],
          "min_flush", sch.get("min_flush"), "quantum", sch.get("pass_quantum"), "exact/step", d["per_rank_split_ms"]["exact_passes_per_step"], "fast", round(d.get("fast_mode", {}).get("value", 0), 1), "cost", round(d.get("exact_cost_vs_fast", 0), 4))
except Exception as e:
    print(name, "failed", e)
PY
done
