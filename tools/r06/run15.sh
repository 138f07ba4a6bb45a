#!/bin/bash
# round 6, GPU session 15: the product entry point at scale -- run.py evaluate on 600 synthetic panoramas (3 batches of 256, 24 layers)
mkdir -p gpurun_out/r06
( time timeout 900 python run.py evaluate none --synthetic 600 --layers 24 ) > gpurun_out/r06/run_py_evaluate_600.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06/run_py_evaluate_600.txt | tail -12 | cut -c1-600
( time timeout 900 python run.py embed random --synthetic 1024 --layers 24 --out-dir /tmp/emb_out --yfcc ) > gpurun_out/r06/run_py_embed_1024.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06/run_py_embed_1024.txt | tail -8 | cut -c1-400
