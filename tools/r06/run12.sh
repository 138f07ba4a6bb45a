#!/bin/bash
# round 6, GPU session 12: the engine's random stress test, then the whole GPU suite on the final tree
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_requeue.py -q -m gpu 2>&1 | tail -15 > gpurun_out/r06/t_run12.txt; cat gpurun_out/r06/t_run12.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r06/gpu_suite_final.txt; cat gpurun_out/r06/gpu_suite_final.txt
