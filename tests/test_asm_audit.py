"""Static guarantees about the compiled gfx950 kernels (no GPU): the hot kernels carry no waterfall loops and (next to) no spills.

hipcc turns a buffer descriptor whose record count went through `min / max` (-> v_med3_i32, a VALU instruction) into a VGPR
descriptor and wraps every load through it in a readfirstlane / saveexec loop; in round 2 that cost 2 % of the benchmark and showed
in no profile (tools/asm_audit.py).  This test reads the ISA of the built library so it cannot come back."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HOT = ("gemm_pp6_kernel", "gemm_pp_kernel", "gemm_w4_kernel", "gemm_tail_kernel", "gemm_bf16_kernel", "attention5_kernel")


def test_hot_kernels_have_no_waterfall_loops_and_no_spills(hip_lib):
    import asm_audit
    res = asm_audit.audit(os.path.join(ROOT, "pigeon_amd", "libpigeon_hip.so"))
    hot = {k: v for k, v in res.items() if any(h in k for h in HOT)}
    assert len(hot) >= 40, f"only {len(hot)} hot kernels found in the library: bundle parsing broken?"
    for name, v in hot.items():
        assert v["waterfall"] == 0, f"{name}: {v['waterfall']} waterfall loop(s) -- a descriptor or soffset lives in VGPRs"
        assert v["scratch"] <= 16, f"{name}: {v['scratch']} bytes of scratch per lane (spills)"
        assert v["vgpr"] > 0 and v["instructions"] > 100
    # the persistent kernels are built for one (pp / pp6 / tail) or two (w4: 256 + 256 AGPRs) waves per SIMD
    for name, v in hot.items():
        if "gemm_pp6_kernel" in name or "gemm_pp_kernel" in name:
            assert v["vgpr"] <= 256 and v["agpr"] == 0, name
