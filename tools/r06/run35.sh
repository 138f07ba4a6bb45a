#!/bin/bash
# round 6, GPU session 35: the reference-module audit on 128 batches (16 384 panoramas) of the SPREAD tower (the closest stand-in for a trained CLIP)
mkdir -p gpurun_out/r06
timeout 3000 python tools/certainty_audit_ref.py 128 spread > gpurun_out/r06/certainty_audit_ref_spread_16384.txt 2>&1; tail -8 gpurun_out/r06/certainty_audit_ref_spread_16384.txt
