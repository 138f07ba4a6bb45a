mkdir -p gpurun_out/r3
python tools/gemm_pp_check.py --variants 56 --timeout 200 > gpurun_out/r3/pp6_resid_check.txt 2>&1
python tools/epi_probe.py 36,56 > gpurun_out/r3/pp6_resid_probe.txt 2>&1
for rep in 1 2; do
for m in 0 1 3; do PIGEON_GEMM_RESID6=$m python bench.py --no-extras --cpu-images 0 --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('resid6=$m', round(d['value'],1), 'img/s out', round(k['gemm_out']['avg_ms'],4), 'fc2', round(k['gemm_fc2']['avg_ms'],4), 'fc1', round(k['gemm_fc1']['avg_ms'],4))"; done
PIGEON_GEMM_RESID6=3 PIGEON_GEMM_TAIL_MIN_K=1024 python bench.py --no-extras --cpu-images 0 --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('resid6=3 tailK1024', round(d['value'],1), 'img/s out', round(k['gemm_out']['avg_ms'],4), 'fc2', round(k['gemm_fc2']['avg_ms'],4), 'fc1', round(k['gemm_fc1']['avg_ms'],4))"
done > gpurun_out/r3/pp6_resid_bench_ab.txt 2>&1
cat gpurun_out/r3/pp6_resid_check.txt | tail -8; grep "row stats" gpurun_out/r3/pp6_resid_probe.txt; cat gpurun_out/r3/pp6_resid_bench_ab.txt
