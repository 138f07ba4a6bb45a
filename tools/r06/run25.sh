#!/bin/bash
# round 6, GPU session 25: the reference-module audit again on the final tree (passes of 7 panoramas, fused exact pass): 256 batches of the
# default tower, 32 of the spread tower
mkdir -p gpurun_out/r06
timeout 3000 python tools/certainty_audit_ref.py 256 default > gpurun_out/r06/certainty_audit_ref_32768_final.txt 2>&1; tail -9 gpurun_out/r06/certainty_audit_ref_32768_final.txt
timeout 1500 python tools/certainty_audit_ref.py 32 spread > gpurun_out/r06/certainty_audit_ref_spread_4096_final.txt 2>&1; tail -9 gpurun_out/r06/certainty_audit_ref_spread_4096_final.txt
