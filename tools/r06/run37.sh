#!/bin/bash
# round 6, GPU session 37: the calibrated systematic part of the 16-bit encoder's error SUBTRACTED from the embeddings (pg_embedding_debias,
# pigeon_amd/certainty.py `debias`) -- kernel test, the contract tests on the real reference's fixtures, the audit against the reference
# module on both towers, and the bench's quick form with the switch off / on (same box); the fp8 / fp6 / fp4 MFMA probe
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_certainty.py tests/test_gpu_top1.py -q -m gpu -x 2>&1 | tail -25 > $O/t_run37.txt; tail -8 $O/t_run37.txt
for d in 0 1; do
  PIGEON_DEBIAS=$d timeout 600 python bench.py --no-extras --cpu-images 0 > $O/debias_$d.json 2> $O/debias_$d.err
  python - <<PY
import json
try:
    r = json.loads(open('$O/debias_$d.json').read().strip().splitlines()[-1])
    c = r['certainty']
    print('PIGEON_DEBIAS=$d', {k: r.get(k) for k in ('value', 'ms_per_step', 'exact_cost_vs_fast')}, 'reencoded_share', c['reencoded_share'], 'uncertain_after', sum(c['uncertain_after_step']), c['uncertain_by_cause'], 'passes/step', r['per_rank_split_ms']['exact_passes_per_step'], 'compute', r['per_rank_split_ms']['compute'])
    print('   ', c['rule'][-420:])
except Exception as e:
    print('parse failed', e); print(open('$O/debias_$d.err').read()[-2000:])
PY
done
timeout 900 python tools/certainty_audit_ref.py 64 default > $O/certainty_audit_ref_debias_8192.txt 2>&1; grep -v amdgpu.ids $O/certainty_audit_ref_debias_8192.txt | cut -c1-600 | tail -8
timeout 600 python tools/certainty_audit_ref.py 32 spread > $O/certainty_audit_ref_debias_spread_4096.txt 2>&1; grep -v amdgpu.ids $O/certainty_audit_ref_debias_spread_4096.txt | cut -c1-600 | tail -8
tools/bin/mfma_fp8_probe > $O/mfma_fp8_probe.txt 2>&1; cat $O/mfma_fp8_probe.txt
