#!/bin/bash
# round 6, GPU session 40: the audits against the reference module at the first session's scale, on the tree with the de-biased
# embeddings (32 768 panoramas of the default tower, 16 384 of the spread tower); the three-kernel GEMM sweep once more (is the slow
# LN-fold column of session 39 the box or the tree?)
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 400 python tools/gemm_mid_sweep.py --three > $O/gemm_three_sweep_run40.txt 2>&1; grep -v amdgpu.ids $O/gemm_three_sweep_run40.txt | cut -c1-330
timeout 1500 python tools/certainty_audit_ref.py 256 default > $O/certainty_audit_ref_debias_32768.txt 2>&1; grep -v amdgpu.ids $O/certainty_audit_ref_debias_32768.txt | cut -c1-600 | tail -8
timeout 900 python tools/certainty_audit_ref.py 128 spread > $O/certainty_audit_ref_debias_spread_16384.txt 2>&1; grep -v amdgpu.ids $O/certainty_audit_ref_debias_spread_16384.txt | cut -c1-600 | tail -8
