"""A/B of the 384 x 256 kernel's two 16-bit epilogues (PIGEON_GEMM_PARK16=0: fp32 slabs, 1: finished in the accumulator layout,
16-bit slabs), optionally with the XCD start stagger (PIGEON_GEMM_STAGGER).  One setting per process: prints a checksum of every
output (all settings must print the same checksums) and the time per launch on the QKV / fc1 shapes."""
import hashlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pigeon_amd import _lib, hip_ops

_lib.require_gpu()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(7)
tag = f"park16={os.environ.get('PIGEON_GEMM_PARK16', '0')} stagger={os.environ.get('PIGEON_GEMM_STAGGER', '0')}"


def digest(t):
    return hashlib.sha256(t.cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:16]


for dt in (torch.float16, torch.bfloat16):
    for (M, N, K) in ((384 * 5 + 37, 512, 256), (295424, 3072, 1024), (295424, 4096, 1024)):
        if dt == torch.bfloat16 and M > 100000:
            continue
        A = torch.randn((M, K), generator=g, device=dev).to(dt)
        W = (torch.randn((N, K), generator=g, device=dev) * 0.05).to(dt)
        bias = torch.randn(N, generator=g, device=dev)
        cs = torch.randn(N, generator=g, device=dev)
        rs = torch.rand((M, 2), generator=g, device=dev) + 0.5
        out = torch.empty((M, N), dtype=dt, device=dev)
        forms = {
            "qkv": lambda: hip_ops.gemm16(A, W, bias, out, _lib.EPI_QKV, qscale=0.125, qcols=256, variant=56),
            "gelu": lambda: hip_ops.gemm16(A, W, bias, out, _lib.EPI_GELU, variant=56),
            "qkv_ln": lambda: hip_ops.gemm16_ln(A, W, bias, cs, rs, _lib.EPI_QKV_LN, qscale=0.125, qcols=256, variant=56),
            "gelu_ln": lambda: hip_ops.gemm16_ln(A, W, bias, cs, rs, _lib.EPI_GELU_LN, variant=56),
        }
        for name, fn in forms.items():
            if M > 100000 and name in ("qkv", "gelu"):
                continue
            out.fill_(7.0)
            r = fn()
            r = out if r is None or r is out else r
            torch.cuda.synchronize()
            line = f"{str(dt)[6:]:9s} {M}x{N}x{K} {name:8s} {digest(r)}"
            if M > 100000:
                for _ in range(20):
                    fn()
                torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(20):
                        fn()
                    b.record(); torch.cuda.synchronize()
                    ts.append(a.elapsed_time(b) / 20)
                line += f"   {sorted(ts)[2]:.4f} ms   [{tag}]"
            print(line, flush=True)
        del A, W, out
