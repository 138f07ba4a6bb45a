"""One-shot GPU diagnostic: checks every kernel of libpigeon_hip.so against torch references and times the
GEMM variants / attention / full ViT.  Meant for `gpurun -- python tools/gpu_diag.py` (writes
gpurun_out/diag.json + prints a report).  Each section is independent and failures are recorded, not fatal,
so one call yields as much information as possible.  Sections: --only a,b,c to restrict."""
import argparse
import json
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pigeon_amd import _lib, hip_ops, synthetic  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
R = {}
dev = "cuda"


def log(*a):
    print(*a, flush=True)


def section(name):
    def deco(fn):
        fn._section = name
        return fn
    return deco


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


DT = torch.float16


def bf16_rand(shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DT).to(dev)


@section("gemm_correct")
def gemm_correct():
    res = {}
    for variant in (1, 2, 3, 4, 5, 7, 11):
        for (M, N, K) in ((300, 256, 64), (300, 256, 128), (1000, 512, 320), (577 * 3, 1024, 1024), (2000, 3072, 192)):
            A = bf16_rand((M, K), 1 + M)
            W = bf16_rand((N, K), 2 + N, 0.05)
            bias = torch.randn(N, device=dev)
            ref = A.float() @ W.float().T + bias
            out = torch.full((M, N), float("nan"), device=dev)
            key = f"v{variant}_M{M}_N{N}_K{K}"
            try:
                hip_ops.gemm16(A, W, bias, out, _lib.EPI_F32, variant=variant)
                torch.cuda.synchronize()
                err = (out - ref).abs().max().item()
                errT = float("nan")
                if M == N:
                    errT = (out.T - ref).abs().max().item()
                res[key] = dict(max_abs_err=err, ref_absmax=ref.abs().max().item(), nan=int(torch.isnan(out).sum().item()))
                log("gemm", key, res[key])
            except Exception as e:  # noqa
                res[key] = dict(error=str(e))
                log("gemm", key, "ERROR", e)
    # transposition / layout detector: A = identity-like, asymmetric W
    M = N = K = 256
    A = torch.eye(256, device=dev).to(DT)
    W = (torch.arange(N, device=dev)[:, None] * 1.0 + torch.arange(K, device=dev)[None, :] * 0.001).to(DT)
    out = torch.zeros((M, N), device=dev)
    hip_ops.gemm16(A, W, None, out, _lib.EPI_F32, variant=2)
    torch.cuda.synchronize()
    ref = A.float() @ W.float().T
    res["identity_err"] = (out - ref).abs().max().item()
    res["identity_err_if_transposed"] = (out.T - ref).abs().max().item()
    log("gemm identity", res["identity_err"], res["identity_err_if_transposed"])
    return res


@section("gemm_epilogues")
def gemm_epilogues():
    res = {}
    for variant in (1, 2, 3, 4, 5, 7, 11):
        M, N, K = 1154, 1024, 256
        A = bf16_rand((M, K), 5)
        W = bf16_rand((N, K), 6, 0.05)
        bias = torch.randn(N, device=dev)
        acc = A.float() @ W.float().T
        # QKV
        out = torch.zeros((M, N), dtype=DT, device=dev)
        hip_ops.gemm16(A, W, bias, out, _lib.EPI_QKV, qscale=0.25, qcols=512, variant=variant)
        ref = acc + bias
        ref[:, :512] *= 0.25
        res[f"v{variant}_qkv"] = (out.float() - ref).abs().max().item() / ref.abs().max().item()
        # GELU
        out = torch.zeros((M, N), dtype=DT, device=dev)
        hip_ops.gemm16(A, W, bias, out, _lib.EPI_GELU, variant=variant)
        y = acc + bias
        ref = y * torch.sigmoid(1.702 * y)
        res[f"v{variant}_gelu"] = (out.float() - ref).abs().max().item() / ref.abs().max().item()
        # RESID
        X0 = torch.randn((M, N), device=dev)
        X = X0.clone()
        hip_ops.gemm16(A, W, bias, X, _lib.EPI_RESID, variant=variant)
        ref = X0 + acc + bias
        res[f"v{variant}_resid"] = (X - ref).abs().max().item() / ref.abs().max().item()
        # PATCH: M = n*576 rows -> rows img*577+1+p
        n = 2
        Ap = bf16_rand((n * 576, 640), 7)
        Wp = bf16_rand((N, 640), 8, 0.05)
        pos = torch.randn((577, N), device=dev)
        Xp = torch.full((n * 577, N), 7.0, device=dev)
        hip_ops.gemm16(Ap, Wp, None, Xp, _lib.EPI_PATCH, aux=pos, variant=variant)
        refp = (Ap.float() @ Wp.float().T).view(n, 576, N) + pos[1:][None]
        got = Xp.view(n, 577, N)
        res[f"v{variant}_patch"] = (got[:, 1:] - refp).abs().max().item() / refp.abs().max().item()
        res[f"v{variant}_patch_cls_untouched"] = float((got[:, 0] == 7.0).all().item())
        torch.cuda.synchronize()
        log("epilogues", variant, {k: v for k, v in res.items() if k.startswith(f"v{variant}_")})
    return res


@section("gemm_perf")
def gemm_perf():
    res = {}
    M = 256 * 577
    for (N, K, name) in ((3072, 1024, "qkv"), (1024, 1024, "out"), (4096, 1024, "fc1"), (1024, 4096, "fc2")):
        A = bf16_rand((M, K), 11)
        W = bf16_rand((N, K), 12, 0.03)
        bias = torch.zeros(N, device=dev)
        for variant in (4, 5, 6, 7):
            if name in ("qkv", "fc1"):
                out = torch.empty((M, N), dtype=DT, device=dev)
                epi = _lib.EPI_QKV if name == "qkv" else _lib.EPI_GELU
            else:
                out = torch.zeros((M, N), dtype=torch.float32, device=dev)
                epi = _lib.EPI_RESID
            try:
                ms = timeit(lambda: hip_ops.gemm16(A, W, bias, out, epi, qscale=0.18, qcols=1024, variant=variant), iters=5)
                tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
                res[f"{name}_v{variant}"] = dict(ms=ms, tflops=tf)
                log("gemm_perf", name, "variant", variant, f"{ms:.3f} ms  {tf:.1f} TF/s")
            except Exception as e:  # noqa
                res[f"{name}_v{variant}"] = dict(error=str(e))
                log("gemm_perf", name, variant, "ERROR", e)
            del out
        del A, W
    return res


@section("rowops")
def rowops():
    res = {}
    x = torch.randn((1000, 1024), device=dev) * 3 + 0.5
    g = torch.randn(1024, device=dev) * 0.1 + 1
    b = torch.randn(1024, device=dev) * 0.1
    ref = torch.nn.functional.layer_norm(x, (1024,), g, b, 1e-5)
    y = hip_ops.layernorm(x, g, b, out_dtype=torch.float32)
    res["ln_f32_max_abs"] = (y - ref).abs().max().item()
    for dt in (torch.float16, torch.bfloat16):
        y = hip_ops.layernorm(x, g, b, out_dtype=dt)
        res[f"ln_{dt}_max_abs_vs_rounded"] = (y.float() - ref.to(dt).float()).abs().max().item()
    px = torch.randn((3, 3, 336, 336), device=dev)
    col = hip_ops.im2col(px, DT)
    refc = torch.nn.functional.unfold(px, kernel_size=14, stride=14).transpose(1, 2).reshape(3 * 576, 588)
    res["im2col_max_abs"] = (col[:, :588].float() - refc.to(DT).float()).abs().max().item()
    res["im2col_pad_zero"] = float((col[:, 588:] == 0).all().item())
    h = torch.randn((5, 577, 1024), device=dev)
    res["token_mean_max_abs"] = (hip_ops.token_mean(h) - h.mean(dim=1)).abs().max().item()
    z = torch.randn(1000003, device=dev)
    res["cast_exact_f16"] = float((hip_ops.cast_f32(z, torch.float16) == z.to(torch.float16)).all().item())
    res["cast_exact_bf16"] = float((hip_ops.cast_f32(z, torch.bfloat16) == z.to(torch.bfloat16)).all().item())
    log("rowops", res)
    return res


def attention_ref(qkv, n):
    q, k, v = qkv.float().view(n, 577, 3, 16, 64).permute(2, 0, 3, 1, 4)      # (3, n, 16, 577, 64)
    s = q @ k.transpose(-1, -2) * float(np.log(2.0))                           # q carries log2(e)/8
    p = torch.softmax(s, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(n * 577, 1024)


@section("attention")
def attention():
    res = {}
    n = 3
    qkv = bf16_rand((n * 577, 3072), 21)
    qkv[:, :1024] *= 0.18 * 1.5
    out = hip_ops.attention(qkv, n)
    torch.cuda.synchronize()
    ref = attention_ref(qkv, n)
    res["max_abs_err"] = (out.float() - ref).abs().max().item()
    res["ref_absmax"] = ref.abs().max().item()
    res["rel_fro"] = ((out.float() - ref).norm() / ref.norm()).item()
    # spiky keys: forces the online-softmax rescale path
    qkv2 = qkv.clone()
    qkv2[300, 1024:2048] *= 40
    out2 = hip_ops.attention(qkv2, n)
    ref2 = attention_ref(qkv2, n)
    res["spiky_rel_fro"] = ((out2.float() - ref2).norm() / ref2.norm()).item()
    log("attention", res)
    n = 256
    qkv = bf16_rand((n * 577, 3072), 22)
    qkv[:, :1024] *= 0.18
    ms = timeit(lambda: hip_ops.attention(qkv, n), iters=5)
    flops = n * 16 * 4.0 * 577 * 577 * 64
    res["perf_ms_n256"] = ms
    res["perf_tflops"] = flops / (ms * 1e-3) / 1e12
    log("attention perf", ms, "ms", res["perf_tflops"], "TF/s")
    return res


@section("head")
def head():
    res = {}
    B, Cn, k = 37, 10000, 50
    emb = torch.randn((B, 4, 1024), device=dev) * 0.7 + 0.1
    W, b = synthetic.make_head_weights(Cn, 0)
    W, b = W.to(dev), b.to(dev)
    cen = torch.from_numpy(synthetic.make_geocells(Cn, 0)).to(dev)
    o = hip_ops.head_forward(emb, W, b, cen, k)
    torch.cuda.synchronize()
    logits = torch.nn.functional.linear(emb.mean(dim=1), W, b)
    probs = torch.softmax(logits, dim=-1)
    tk = torch.topk(probs, k, dim=-1)
    res["logits_max_abs"] = (o["logits"] - logits).abs().max().item()
    res["argmax_mismatch"] = int((o["preds_geocell"] != probs.argmax(dim=-1)).sum().item())
    res["topk_idx_mismatch"] = int((o["topk_indices"] != tk.indices).sum().item())
    res["topk_val_max_rel"] = ((o["topk_values"] - tk.values).abs() / tk.values).max().item()
    res["llh_exact"] = float((o["preds_LLH"] == cen[o["preds_geocell"]]).all().item())
    log("head", res)
    return res


@section("refine")
def refine():
    sys.path.insert(0, ROOT)
    from oracle import pigeon_oracle as orc
    res = {}
    bank = synthetic.make_bank(200, 12, seed=2, empty_frac=0.05)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "refine.npz"))
    dbank = hip_ops.DeviceBank(bank)
    emb = torch.from_numpy(gold["embedding"]).to(dev)
    cands = torch.from_numpy(gold["candidate_cells"]).to(dev)
    probs = torch.from_numpy(gold["candidate_probs"]).to(dev)
    init = torch.from_numpy(gold["initial_preds"]).to(dev)
    for tag in ("default", "evaluate", "tight"):
        topk, T, mr = gold[f"{tag}_params"]
        llh, cell, choice = hip_ops.refine_forward(dbank, emb, init, cands, probs, int(topk), float(T), float(mr))
        torch.cuda.synchronize()
        res[f"{tag}_cell_mismatch"] = int((cell.cpu().numpy() != gold[f"{tag}_cell"]).sum())
        res[f"{tag}_llh_max_abs"] = float(np.abs(llh.cpu().numpy() - gold[f"{tag}_LLH"]).max())
    emb3 = emb[:, None, :] + torch.tensor([0.1, -0.1, 0.2, -0.2], device=dev)[None, :, None]
    llh, cell, _ = hip_ops.refine_forward(dbank, emb3.contiguous(), init, cands, None, 5, 1.6, 1000.0)
    res["noprobs3d_cell_mismatch"] = int((cell.cpu().numpy() != gold["noprobs3d_cell"]).sum())
    res["noprobs3d_llh_max_abs"] = float(np.abs(llh.cpu().numpy() - gold["noprobs3d_LLH"]).max())
    log("refine", res)
    # perf-size bank: 10000 cells x 100 protos (4.1 GB), B = 128, top-5
    try:
        big = synthetic.make_bank(10000, 100, seed=2, exact_means=False)
        dbig = hip_ops.DeviceBank(big)
        B = 128
        g = torch.Generator().manual_seed(3)
        q = torch.randn((B, 4, 1024), generator=g).to(dev)
        cand = torch.randint(0, 10000, (B, 5), generator=g).to(dev)
        cp = torch.softmax(torch.randn((B, 5), generator=g), dim=-1).to(dev)
        ini = torch.zeros((B, 2), dtype=torch.float64, device=dev)
        ms = timeit(lambda: hip_ops.refine_forward(dbig, q, ini, cand, cp, 5, 1.6, 1000.0), iters=10)
        rows = 0
        co = big.cell_off
        cnp = cand.cpu().numpy()
        for bq in range(B):
            for j in range(5):
                rows += int(co[cnp[bq, j] + 1] - co[cnp[bq, j]])
        res["perf_ms"] = ms
        res["perf_GBps_proto_rows_only"] = rows * 4096 / (ms * 1e-3) / 1e9
        log("refine perf", ms, "ms", res["perf_GBps_proto_rows_only"], "GB/s (prototype rows only)")
        # oracle cross-check on a 16-query sample of the big bank
        o = orc.proto_refiner_forward(big, q[:16].cpu(), ini[:16].cpu(), cand[:16].cpu(), cp[:16].cpu(), 5, 1.6, 1000.0)
        llh, cell, _ = hip_ops.refine_forward(dbig, q[:16].contiguous(), ini[:16].contiguous(), cand[:16].contiguous(),
                                              cp[:16].contiguous(), 5, 1.6, 1000.0)
        res["big_cell_mismatch"] = int((cell.cpu() != o[2]).sum().item())
        res["big_llh_max_abs"] = float((llh.cpu() - o[1]).abs().max().item())
        log("refine big-bank parity", res["big_cell_mismatch"], res["big_llh_max_abs"])
    except Exception as e:  # noqa
        res["perf_error"] = str(e)
        log("refine perf ERROR", e)
    return res


@section("vit2")
def vit2():
    from oracle import pigeon_oracle as orc
    res = {}
    gold = np.load(os.path.join(ROOT, "tests", "golden", "vit2.npz"))
    sd = synthetic.make_vit_weights(seed=11, layers=2, affine_jitter=True)
    px = synthetic.make_pixels(4, seed=77)
    encb = hip_ops.VitEncoder(sd, mma_dtype="bf16")
    embb = encb.forward(px.to(dev))
    res["bf16_emb_rel_err"] = orc.rel_err(embb.cpu(), torch.from_numpy(gold["embedding"]))
    encb.close()
    enc = hip_ops.VitEncoder(sd)
    res["mma_dtype"] = enc.mma_dtype
    emb, hid = enc.forward(px.to(dev), return_hidden=True)
    torch.cuda.synchronize()
    ref = torch.from_numpy(gold["embedding"])
    res["emb_rel_err"] = orc.rel_err(emb.cpu(), ref)
    res["emb_max_row_rel"] = orc.max_rel_err_rows(emb.cpu(), ref)
    rows = torch.from_numpy(gold["lhs_rows"])
    got = hid.cpu()[:, [0, 1, 2, 288, 575, 576]]
    res["hidden_rows_rel_err"] = orc.rel_err(got, rows)
    res["hidden_rows_max_abs"] = float((got - rows).abs().max())
    # per-stage localisation against the oracle
    coll = {}
    orc.vit_last_hidden_state(sd, px[:1], collect=coll)
    res["layer1_rel_err_img0"] = orc.rel_err(hid.cpu()[:1], coll["layer1"])
    log("vit2", res)
    enc.close()
    return res


@section("vit24")
def vit24():
    from oracle import pigeon_oracle as orc
    res = {}
    sd = synthetic.make_vit_weights(seed=0, layers=24)
    enc = hip_ops.VitEncoder(sd)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "vit24.npz"))
    px = synthetic.make_pixels(4, seed=1234)
    emb = enc.forward(px.to(dev))
    torch.cuda.synchronize()
    ref = torch.from_numpy(gold["embedding"])
    res["emb_rel_err"] = orc.rel_err(emb.cpu(), ref)
    res["emb_max_row_rel"] = orc.max_rel_err_rows(emb.cpu(), ref)
    res["emb_max_abs"] = float((emb.cpu() - ref).abs().max())
    log("vit24 parity", res)
    for n in (64, 256):
        pxb = torch.randn((n, 3, 336, 336), device=dev)
        ms = timeit(lambda: enc.forward(pxb), iters=3, warm=1)
        res[f"n{n}_ms"] = ms
        res[f"n{n}_img_per_s"] = n / (ms * 1e-3)
        res[f"n{n}_mfma_frac"] = n * 381.918e9 / (ms * 1e-3) / 2.5e15
        log(f"vit24 perf n={n}: {ms:.1f} ms  {res[f'n{n}_img_per_s']:.1f} img/s  frac {res[f'n{n}_mfma_frac']:.3f}")
    enc.profile_enable(True)
    enc.profile_reset()
    enc.forward(pxb)
    torch.cuda.synchronize()
    prof = enc.profile_read()
    enc.profile_enable(False)
    res["profile_n256"] = prof
    for kname, (cnt, ms) in prof.items():
        log(f"   {kname:12s} launches {cnt:4d}  total {ms:9.3f} ms")
    del sd
    # stress weights parity
    sd = synthetic.make_vit_weights(seed=5, layers=24, affine_jitter=True, scale=2.0)
    enc2 = hip_ops.VitEncoder(sd)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "vit24_stress.npz"))
    px = synthetic.make_pixels(2, seed=99)
    emb = enc2.forward(px.to(dev))
    ref = torch.from_numpy(gold["embedding"])
    res["stress_emb_rel_err"] = orc.rel_err(emb.cpu(), ref)
    res["stress_emb_max_abs"] = float((emb.cpu() - ref).abs().max())
    log("vit24 stress parity", res["stress_emb_rel_err"], res["stress_emb_max_abs"])
    return res


ALL = [gemm_correct, gemm_epilogues, rowops, attention, head, refine, vit2, gemm_perf, vit24]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--out", default=os.path.join(OUT, "diag.json"))
    args = ap.parse_args()
    only = set(args.only.split(",")) if args.only else None
    log("devices:", _lib.require_gpu(), torch.cuda.get_device_name(0))
    for fn in ALL:
        if only and fn._section not in only:
            continue
        t0 = time.time()
        try:
            R[fn._section] = fn()
        except Exception as e:  # noqa
            R[fn._section] = dict(error=str(e), trace=traceback.format_exc())
            log("SECTION", fn._section, "FAILED:", e)
            log(traceback.format_exc())
        R[fn._section + "_seconds"] = time.time() - t0
        with open(args.out, "w") as f:
            json.dump(R, f, indent=1, default=str)
    log("wrote", args.out)


if __name__ == "__main__":
    main()
