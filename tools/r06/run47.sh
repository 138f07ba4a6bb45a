#!/bin/bash
# round 6, GPU session 47: last tree (the exact tier's routing constant behind a knob, default unchanged): GPU suite, smoke(), quick form of the bench command
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -30 > $O/gpu_suite_run47.txt; tail -3 $O/gpu_suite_run47.txt
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python bench.py --no-extras --cpu-images 0 > $O/bench_quick_run47.json 2> $O/bench_quick_run47.err
python - <<'PY'
import json
txt = open('gpurun_out/r06/bench_quick_run47.json').read().strip().splitlines()
r = json.loads(txt[-1]); print('stdout lines:', len(txt), {k: r.get(k) for k in ('value', 'ms_per_step', 'exact_cost_vs_fast')}, r['per_rank_split_ms']['compute'], r['per_rank_split_ms']['exact_passes_per_step'], r['parity_vs_oracle_sample'].get('geocell_argmax_equal') if 'parity_vs_oracle_sample' in r else None)
PY
