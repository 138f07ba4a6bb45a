#!/bin/bash
# round 5, GPU session 15: certainty audit on the spread tower (4096 panoramas) + the whole GPU suite on the final tree
mkdir -p gpurun_out/r05
timeout 600 python tools/certainty_audit.py 32 spread > gpurun_out/r05/certainty_audit_spread_4096.txt 2>&1
tail -4 gpurun_out/r05/certainty_audit_spread_4096.txt
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r05/gpu_suite_final2.txt
cat gpurun_out/r05/gpu_suite_final2.txt
