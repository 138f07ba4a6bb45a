"""Where does a persistent GEMM block spend a tile period?  Tools build only (python -m pigeon_amd.build --dev):
PG_TS stamps the 100 MHz wall clock inside gemm_pp.hip / gemm_pp6.hip (blocks 0 and 100, all 8 waves, first 16 tiles):
slot 0 tile start (B_0 passed), 1 mainloop left, 2 re-aligned, 3.. slab i done, 9 epilogue done.

   PIGEON_HIP_LIB=pigeon_amd/libpigeon_hip_dev.so python tools/epi_timeline.py [out|fc2|qkv|fc1 ...]
"""
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
FORMS = {"out": (1024, 1024, "resid_stat"), "fc2": (1024, 4096, "resid_stat"), "qkv": (3072, 1024, "qkv_ln"), "fc1": (4096, 1024, "gelu_ln"),
         "fc1q": (4096, 1024, "qkv_ln")}       # fc1's shape with the QKV epilogue (no QuickGELU): what the transcendentals + 3 packed ops cost
which = sys.argv[1:] or list(FORMS)
NSLAB = {"resid_stat": 4, "qkv_ln": 6, "gelu_ln": 6}
for name in which:
    N, K, kind = FORMS[name]
    A = torch.randn((M, K), generator=g, device=dev).to(dt)
    W = (torch.randn((N, K), generator=g, device=dev) * 0.03).to(dt)
    bias = torch.zeros(N, device=dev)
    cs = torch.zeros(N, device=dev)
    rs = torch.ones((M, 2), device=dev)
    X = torch.zeros((M + 384, N), device=dev)[:M] if kind == "resid_stat" else None   # + the slack the residual epilogue may read

    def run():
        if kind == "resid_stat":
            hip_ops.gemm16_resid_stat(A, W, bias, X)
        else:
            hip_ops.gemm16_ln(A, W, bias, cs, rs, _lib.EPI_QKV_LN if kind == "qkv_ln" else _lib.EPI_GELU_LN, qscale=0.125, qcols=1024)

    for _ in range(6):
        run()
    torch.cuda.synchronize()
    buf = torch.zeros(2 * 16 * 8 * 12, dtype=torch.int64, device=dev)
    lib.pg_dbg_timestamps(C.c_void_p(buf.data_ptr()))
    run()
    torch.cuda.synchronize()
    lib.pg_dbg_timestamps(C.c_void_p(0))
    t = buf.cpu().view(2, 16, 8, 12).double() / 100.0          # us
    ns = NSLAB[kind]
    print(f"== {name}: N={N} K={K} {kind}; us relative to the tile start of wave 0; tiles 4..7 of blocks 0 and 100")
    print("  start of block 100's tile minus block 0's (us), tiles 0..9: " + " ".join(f"{(t[1, i, 0, 0] - t[0, i, 0, 0]):6.1f}" for i in range(10)))
    print("  tile period of block 0 (us), tiles 0..9: " + " ".join(f"{(t[0, i + 1, 0, 0] - t[0, i, 0, 0]):6.1f}" for i in range(10)))
    # one line for sweeps (PIGEON_GEMM_BLOCKS=n caps the persistent grid: fewer CUs share HBM / the fabric -- does a tile's epilogue
    # get shorter?): block 0, mean over tiles 4..11
    its = range(4, 12)
    ml = sum(float(t[0, i, 0, 1] - t[0, i, 0, 0]) for i in its) / 8
    ea = sum(float(t[0, i, 0, 9] - t[0, i, 0, 1]) for i in its) / 8
    eb = sum(float(t[0, i, 4, 9] - t[0, i, 4, 1]) for i in its) / 8
    per = sum(float(t[0, i + 1, 0, 0] - t[0, i, 0, 0]) for i in its) / 8
    print(f"SUMMARY {name} blocks={os.environ.get('PIGEON_GEMM_BLOCKS', 'all')}: mainloop {ml:6.1f} us  epilogue waves 0-3 {ea:5.1f}  waves 4-7 {eb:5.1f}  tile period {per:6.1f}")
    if os.environ.get("EPI_TIMELINE_SUMMARY_ONLY"):
        del A, W
        continue
    for blk in (0, 1):
        for it in (4, 5, 6, 7):
            t0 = t[blk, it, 0, 0]
            for w in (0, 3, 4, 7):
                r = t[blk, it, w]
                slabs = " ".join(f"{(r[3 + i] - t0):6.1f}" for i in range(ns))
                nxt = t[blk, it + 1, w, 0] - t0
                print(f"  blk {blk * 100:3d} tile {it} wave {w}: start {r[0] - t0:5.1f}  mainloop end {r[1] - t0:6.1f}  realigned {r[2] - t0:6.1f}  slabs {slabs}  epi end {r[9] - t0:6.1f}  next start {nxt:6.1f}")
    del A, W
