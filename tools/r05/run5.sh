#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export TMPDIR=/tmp
rm -f gpurun_out/r05/exact_small_batches_model.txt
for n in 4 8 16 32 52; do timeout 200 python tools/exact_prof.py $n 3 2>/dev/null | tail -1 >> gpurun_out/r05/exact_small_batches_model.txt; done
cat gpurun_out/r05/exact_small_batches_model.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25 > gpurun_out/r05/t_full.txt
tail -12 gpurun_out/r05/t_full.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05/smoke.txt 2>&1; tail -3 gpurun_out/r05/smoke.txt
