mkdir -p gpurun_out/r3
export PIGEON_HIP_LIB=$PWD/pigeon_amd/libpigeon_hip_dev.so
export PIGEON_GEMM_RESID6=0
for rep in 1 2; do
python tools/epi_probe.py 36 2>&1 | grep "row stats" | sed 's/^/normal  /'
PIGEON_EPI_ABL=1 python tools/epi_probe.py 36 2>&1 | grep "row stats" | sed 's/^/8B-elem /'
done > gpurun_out/r3/epi_split_ablation.txt 2>&1
cat gpurun_out/r3/epi_split_ablation.txt
