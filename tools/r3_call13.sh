mkdir -p gpurun_out/r3
for rep in 1 2; do
for cfg in "1 2048" "3 2048" "3 1024"; do set -- $cfg; PIGEON_GEMM_RESID6=$1 PIGEON_GEMM_TAIL_MIN_K=$2 python bench.py --no-extras --cpu-images 0 --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('resid6=$1 tailK=$2', round(d['value'],1), 'img/s out', round(k['gemm_out']['avg_ms'],4), 'fc2', round(k['gemm_fc2']['avg_ms'],4))"; done
done > gpurun_out/r3/pp6_resid_bench_ab3.txt 2>&1
cat gpurun_out/r3/pp6_resid_bench_ab3.txt
