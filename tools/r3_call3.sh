mkdir -p gpurun_out/r3
(python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/r3/gputest2.txt
python __graft_entry__.py smoke > gpurun_out/r3/smoke2.txt 2>&1
export PIGEON_HIP_LIB=$PWD/pigeon_amd/libpigeon_hip_dev.so
for v in 11 16 11 16; do PIGEON_ATTN_VARIANT=$v python tools/attn_bench.py --images 512 --iters 10 --rounds 7; done > gpurun_out/r3/attn_ab.txt 2>&1
for v in 11 16 11 16; do PIGEON_ATTN_VARIANT=$v python bench.py --no-extras --cpu-images 0 --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $v', round(d['value'],1), 'img/s', 'attn', round(d['kernels']['attention']['avg_ms'],4))"; done > gpurun_out/r3/attn_bench_ab.txt 2>&1
tail -12 gpurun_out/r3/gputest2.txt; tail -4 gpurun_out/r3/smoke2.txt; cat gpurun_out/r3/attn_ab.txt gpurun_out/r3/attn_bench_ab.txt
