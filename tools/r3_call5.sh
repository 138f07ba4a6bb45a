mkdir -p gpurun_out/r3
export PIGEON_HIP_LIB=$PWD/pigeon_amd/libpigeon_hip_dev.so
for v in 11 16 18 19 20 11 16 18 19; do PIGEON_ATTN_VARIANT=$v timeout 120 python tools/attn_bench.py --images 512 --iters 10 --rounds 7 2>&1 | grep -E "ATTN|Error|error" | tail -3; done > gpurun_out/r3/attn_ab2.txt 2>&1
unset PIGEON_HIP_LIB
for p in all dominant none all dominant none; do python bench.py --no-extras --cpu-images 0 --steps 10 --profile $p 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('profile $p', round(d['value'],1), 'img/s', 'fc1', round(d['kernels']['gemm_fc1']['avg_ms'],4), 'frac', round(d['roofline']['frac'],4))"; done > gpurun_out/r3/profile_ab.txt 2>&1
cat gpurun_out/r3/attn_ab2.txt gpurun_out/r3/profile_ab.txt
