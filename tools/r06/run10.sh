#!/bin/bash
# round 6, GPU session 10: the reference-module audit on 256 batches (32 768 panoramas) of the default tower
mkdir -p gpurun_out/r06
timeout 3000 python tools/certainty_audit_ref.py 256 default > gpurun_out/r06/certainty_audit_ref_32768.txt 2>&1; tail -9 gpurun_out/r06/certainty_audit_ref_32768.txt
