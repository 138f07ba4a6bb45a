"""A/B of library builds on the model's three 384 x 256 GEMMs (QKV / fc1 with the LayerNorm-fold epilogues, fc2 with the residual +
statistics epilogue), 512 images per launch: time per launch and a hash of every output (the arms must be bit-identical).
   python tools/gemm_ab.py                      # this process, library from PIGEON_HIP_LIB (default: the product build)
   python tools/gemm_ab.py --libs e4 e8 e12     # product build vs pigeon_amd/libpigeon_hip_<name>.so, alternating, 3 rounds"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import torch
    from pigeon_amd import _lib, hip_ops
    dev, dt = "cuda", torch.float16
    M = 512 * 577
    g = torch.Generator(device=dev).manual_seed(1)
    out = []
    V = int(os.environ.get("GEMM_AB_VARIANT", "56"))          # 56: 384 x 256 tiles where they exist; 36: 256 x 256 everywhere
    for name, (N, K, kind) in {"qkv": (3072, 1024, "qkv_ln"), "fc1": (4096, 1024, "gelu_ln"), "fc2": (1024, 4096, "resid_stat"),
                               "out": (1024, 1024, "resid_stat")}.items():
        A = torch.randn((M, K), generator=g, device=dev).to(dt)
        W = (torch.randn((N, K), generator=g, device=dev) * 0.03).to(dt)
        bias = torch.randn(N, generator=g, device=dev) * 0.1
        cs = torch.randn(N, generator=g, device=dev) * 0.1
        rs = torch.rand((M, 2), generator=g, device=dev) + 0.5
        X = torch.zeros((M + 384, N), device=dev)[:M] if kind == "resid_stat" else None   # + the slack the residual epilogue may read

        def run():
            if kind == "resid_stat":
                return hip_ops.gemm16_resid_stat(A, W, bias, X, variant=V)
            return hip_ops.gemm16_ln(A, W, bias, cs, rs, _lib.EPI_QKV_LN if kind == "qkv_ln" else _lib.EPI_GELU_LN, qscale=0.125, qcols=1024, variant=V)
        r = run()
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for t in (r if isinstance(r, tuple) else (r,)) + ((X,) if X is not None else ()):
            h.update(t.cpu().numpy().tobytes())
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(8):
                run()
            b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 8)
        ts.sort()
        out.append(f"{name} {ts[2]:.3f} ms ({h.hexdigest()[:10]})")
        del A, W, X
    print(os.path.basename(os.environ.get("PIGEON_HIP_LIB", "libpigeon_hip.so")) + f" variant {V}: " + "  ".join(out), flush=True)


if __name__ == "__main__":
    if "--libs" in sys.argv:
        names = sys.argv[sys.argv.index("--libs") + 1:]
        libs = [os.path.join(ROOT, "pigeon_amd", "libpigeon_hip.so")] + [os.path.join(ROOT, "pigeon_amd", f"libpigeon_hip_{n}.so") for n in names]
        for rnd in range(3):
            for lib in libs:
                subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, PIGEON_HIP_LIB=lib))
    else:
        one()
