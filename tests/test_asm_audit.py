"""Static guarantees about the compiled gfx950 kernels (no GPU): the hot kernels carry no waterfall loops and (next to) no spills.

hipcc turns a buffer descriptor whose record count went through `min / max` (-> v_med3_i32, a VALU instruction) into a VGPR
descriptor and wraps every load through it in a readfirstlane / saveexec loop; in round 2 that cost 2 % of the benchmark and showed
in no profile (tools/asm_audit.py).  This test reads the ISA of the built library so it cannot come back."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HOT = ("gemm_pp6_kernel", "gemm_pp_kernel", "gemm_tail_kernel", "gemm_mid_kernel", "gemm_bf16_kernel", "attention8_kernel")


def test_hot_kernels_have_no_waterfall_loops_and_no_spills(hip_lib):
    import asm_audit
    res = asm_audit.audit(os.path.join(ROOT, "pigeon_amd", "libpigeon_hip.so"))
    hot = {k: v for k, v in res.items() if any(h in k for h in HOT)}
    assert len(hot) >= 30, f"only {len(hot)} hot kernels found in the library: bundle parsing broken?"
    for name, v in hot.items():
        assert v["waterfall"] == 0, f"{name}: {v['waterfall']} waterfall loop(s) -- a descriptor or soffset lives in VGPRs"
        assert v["scratch"] <= 16, f"{name}: {v['scratch']} bytes of scratch per lane (spills)"
        assert v["vgpr"] > 0 and v["instructions"] > 100
    # the persistent kernels are built for one (pp / pp6 / tail) or two (w4: 256 + 256 AGPRs) waves per SIMD
    for name, v in hot.items():
        if "gemm_pp6_kernel" in name or "gemm_pp_kernel" in name:
            assert v["vgpr"] <= 256 and v["agpr"] == 0, name
        if "gemm_mid_kernel" in name:                     # round 6: the small-batch kernel (4 waves, 64 accumulators each)
            assert v["vgpr"] <= 128 and v["scratch"] == 0, (name, v)


def test_mfma_result_hazard_detector_on_synthetic_isa():
    """The detector itself: an asm-issued MFMA whose destination is read too early is reported; the accumulate chain, an
    `s_nop 7` pad and unrelated instructions are not."""
    import asm_audit
    dis = """
0000000000001000 <bad_kernel>:
	v_mfma_f32_16x16x32_f16 v[0:3], v[128:131], v[160:163], v[0:3]
	v_mfma_f32_16x16x32_f16 v[0:3], v[132:135], v[164:167], v[0:3]
	s_nop 1
	v_add_f32_e32 v9, v2, v8
	s_endpgm
0000000000002000 <good_kernel>:
	v_mfma_f32_16x16x32_f16 v[0:3], v[128:131], v[160:163], 0
	v_mfma_f32_16x16x32_f16 v[0:3], v[132:135], v[164:167], v[0:3]
	v_mfma_f32_16x16x32_f16 v[4:7], v[132:135], v[164:167], v[4:7]
	v_add_f32_e32 v9, v10, v8
	s_nop 7
	ds_write_b128 v20, v[0:3]
	s_endpgm
0000000000003000 <bad_overwrite>:
	v_mfma_f32_32x32x16_f16 v[0:15], v[128:131], v[160:163], v[0:15]
	s_nop 7
	v_mov_b32_e32 v15, 0
	s_endpgm
"""
    hz = asm_audit.mfma_hazards(dis)
    assert set(hz) == {"bad_kernel", "bad_overwrite"}, hz
    assert hz["bad_kernel"][0][2] == 3 and "v_add_f32" in hz["bad_kernel"][0][1]          # first MFMA: chained MFMA + s_nop 1 = 3 states where 8 are needed
    assert hz["bad_overwrite"][0][2] == 8                                                  # 8 states where the 8-pass op needs 12


def test_asm_mfma_results_are_never_touched_early(hip_lib):
    """ADVICE r02 (common.h:69): the persistent GEMMs issue MFMAs as inline asm, where hipcc pads no hazards -- the hand-placed
    `s_nop 7; s_nop 7` / barriers are checked in the final ISA of every kernel of the product library."""
    import asm_audit
    lib = os.path.join(ROOT, "pigeon_amd", "libpigeon_hip.so")
    hz = {k: v for k, v in asm_audit.audit_hazards(lib).items()
          if any(n in k for n in ("gemm_pp_kernel", "gemm_pp6_kernel", "gemm_tail_kernel"))}     # the kernels with asm-issued MFMAs
    assert not hz, {k: v[:2] for k, v in hz.items()}                     # (builtin MFMAs are padded by hipcc itself)
    # and the scan really saw the MFMA kernels (K = 128 instantiations included: every gemm_pp / pp6 / tail kernel)
    n = 0
    for co in asm_audit.code_objects(lib):
        n += co.count(b"gemm_pp6_kernel") + co.count(b"gemm_tail_kernel")
    assert n > 0
