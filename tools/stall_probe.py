"""Where does a wave of the persistent 384 x 256 GEMM spend a K tile?  Tools build only (python -m pigeon_amd.build --dev):
gemm_pp6.hip accumulates, for waves 0 (leader group) and 4 (follower group) of block 0, the wall-clock ticks (100 MHz) of the four
sections of every ping-pong phase: LOAD (fragment ds_reads, the next K tile's DMA issue, the lgkmcnt / vmcnt waits), the barrier in
front of the MFMAs, the 24 MFMAs, the barrier behind them.
   PIGEON_HIP_LIB=pigeon_amd/libpigeon_hip_dev.so python tools/stall_probe.py [fc2 fc1 qkv]"""
import ctypes as C
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PIGEON_HIP_LIB", os.path.join(ROOT, "pigeon_amd", "libpigeon_hip_dev.so"))
from pigeon_amd import _lib, hip_ops

_lib.require_gpu()
lib = C.CDLL(os.environ["PIGEON_HIP_LIB"])
dev, dt = "cuda", torch.float16
M = 512 * 577
g = torch.Generator(device=dev).manual_seed(1)
FORMS = {"fc2": (1024, 4096, "resid_stat"), "qkv": (3072, 1024, "qkv_ln"), "fc1": (4096, 1024, "gelu_ln")}
for name in (sys.argv[1:] or list(FORMS)):
    N, K, kind = FORMS[name]
    A = torch.randn((M, K), generator=g, device=dev).to(dt)
    W = (torch.randn((N, K), generator=g, device=dev) * 0.03).to(dt)
    bias = torch.zeros(N, device=dev); cs = torch.zeros(N, device=dev); rs = torch.ones((M, 2), device=dev)
    X = torch.zeros((M + 384, N), device=dev)[:M] if kind == "resid_stat" else None   # + the slack the residual epilogue may read

    def run():
        if kind == "resid_stat":
            hip_ops.gemm16_resid_stat(A, W, bias, X, variant=56)
        else:
            hip_ops.gemm16_ln(A, W, bias, cs, rs, _lib.EPI_QKV_LN if kind == "qkv_ln" else _lib.EPI_GELU_LN, qscale=0.125, qcols=1024, variant=56)
    for _ in range(4):
        run()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    vm = (C.c_ulonglong * 32)()
    lib.pg_dbg_phase_read(buf, 1)
    lib.pg_dbg_vm_read(vm, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        run()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    lib.pg_dbg_phase_read(buf, 1)
    lib.pg_dbg_vm_read(vm, 1)
    w = list(vm)
    print(f"{name}: wait for the next K tile's DMAs (s_waitcnt vmcnt(0)), per wave of block 0: " + "  ".join(
        f"w{i} {w[4 * i] / max(1, w[4 * i + 1]) * 10:.0f} ns ({100.0 * w[4 * i + 2] / max(1, w[4 * i + 1]):.1f} % > 200 ns)" for i in range(8)), flush=True)
    v = list(buf)
    for grp, nm in ((0, "wave 0 (leader group)"), (1, "wave 4 (follower group)")):
        ld, b1, mm, b2, n = [v[grp * 8 + i] for i in range(5)]
        n = max(1, n)
        tot = (ld + b1 + mm + b2) / n * 10
        print(f"{name} N={N} K={K}: launch {ms:.3f} ms; {nm}: per phase {tot:.0f} ns = LOAD {ld / n * 10:.0f} + barrier {b1 / n * 10:.0f} + "
              f"24 MFMAs {mm / n * 10:.0f} + barrier {b2 / n * 10:.0f}  ({n} phases; 4 phases = one 384x256x64 K tile: {4 * tot:.0f} ns; "
              f"24 MFMAs at 16 cycles are {24 * 16 / 1.8:.0f} ns at 1.8 GHz, {24 * 16 / 2.4:.0f} at 2.4)", flush=True)
    del A, W
