mkdir -p gpurun_out/r3
python tools/gemm_pp_check.py --variants 56 --timeout 200 > gpurun_out/r3/pp6_resid_check2.txt 2>&1
for rep in 1 2 3; do
for m in 0 1; do PIGEON_GEMM_RESID6=$m python bench.py --no-extras --cpu-images 0 --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('resid6=$m', round(d['value'],1), 'img/s out', round(k['gemm_out']['avg_ms'],4), 'fc2', round(k['gemm_fc2']['avg_ms'],4), 'fc1', round(k['gemm_fc1']['avg_ms'],4))"; done
done > gpurun_out/r3/pp6_resid_bench_ab2.txt 2>&1
(python -m pytest tests -m gpu -q 2>&1 | tail -12) > gpurun_out/r3/gputest6.txt
tail -3 gpurun_out/r3/pp6_resid_check2.txt; cat gpurun_out/r3/pp6_resid_bench_ab2.txt; grep -E "passed|failed|FAILED" gpurun_out/r3/gputest6.txt
