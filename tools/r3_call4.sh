mkdir -p gpurun_out/r3
(python -m pytest tests -m gpu -q 2>&1 | tail -60) > gpurun_out/r3/gputest3.txt
export PIGEON_HIP_LIB=$PWD/pigeon_amd/libpigeon_hip_dev.so
for v in 11 16 18 19 20 11 16 18 19; do PIGEON_ATTN_VARIANT=$v timeout 120 python tools/attn_bench.py --images 512 --iters 10 --rounds 7 2>&1 | grep ATTN; done > gpurun_out/r3/attn_ab2.txt 2>&1
grep -E "passed|failed|FAILED|flip|pipeline24|pixels" gpurun_out/r3/gputest3.txt | tail -30; cat gpurun_out/r3/attn_ab2.txt
