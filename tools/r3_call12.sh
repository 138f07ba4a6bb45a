mkdir -p gpurun_out/r3
export PIGEON_HIP_LIB=$PWD/pigeon_amd/libpigeon_hip_dev.so
for sch in 0 1; do PIGEON_W4_SCHED=$sch python tools/gemm_pp_check.py --variants 64 --timeout 200 2>&1 | tail -2; PIGEON_W4_SCHED=$sch python tools/epi_probe.py 64 2>&1 | grep -E "RMW"; done > gpurun_out/r3/w4_sched.txt 2>&1
python tools/epi_probe.py 36 2>&1 | grep -E "RMW" >> gpurun_out/r3/w4_sched.txt
cat gpurun_out/r3/w4_sched.txt
