#!/bin/bash
# round 6, GPU session 2: the three failures of session 1 fixed (empty batches, the test's threshold), certainty restatement with the
# new bounds, embed guard on the all-heads-high-gain tower; rocprof kernel stats of the bench command (is any ATen kernel left inside a step?)
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_requeue.py tests/test_gpu_certainty.py "tests/test_gpu_parity.py::test_empty_and_single_sample_batches" \
  "tests/test_gpu_precise.py::test_exact_mode_edges" "tests/test_gpu_precise.py::test_stress_towers_vs_reference_module" tests/test_gpu_top1.py -q -m gpu 2>&1 | tail -30 > gpurun_out/r06/t_run2.txt
cat gpurun_out/r06/t_run2.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r06/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extras --cpu-images 0 --fast-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/r06/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r06/bench_prof.err
cd $GRAFT_REPO_ROOT
find gpurun_out/r06/prof_bench -name "*kernel_stats.csv" | head -3
f=$(find gpurun_out/r06/prof_bench -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp $f gpurun_out/r06/bench_kernel_stats.csv; cut -c1-160 $f | head -60; fi
find gpurun_out/r06/prof_bench -type f ! -name "*stats*" -delete
