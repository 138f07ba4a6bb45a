"""The encoder's EXACT mode and the top-1 certainty signal (round 4; run with -m gpu on an MI355X).

north_star: "geocell argmax bit-exact".  The reference's `torch.argmax(geocell_probs)` (models/super_guessr.py:454) is fp32 end
to end; the fast path's 16-bit MFMA operands leave a 2.7e-4 .. 6e-4 relative embedding error, so panoramas whose top-1 / top-2
margin is inside that band may flip.  What is tested here:

  * the building blocks of pg_vit_forward_precise against fp64 torch on the same inputs: the split-fp16 GEMM (triple operands on
    the persistent MFMA kernel), fp32 attention, LayerNorm -> triple, QuickGELU -> triple;
  * pg_vit_forward_precise against the reference-generated goldens (vit2, vit24, pipeline24_spread) at a tolerance 100x tighter
    than the fast path's (EXACT_TOL);
  * pg_head_margin against torch (the certainty kernels that superseded it: tests/test_gpu_certainty.py);
  * the exact mode at the edges of its inputs.  The end-to-end contract on the REAL reference's fixtures: tests/test_gpu_top1.py.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"
EXACT_TOL = 1e-5        # embeddings of the exact mode vs the fp32 reference (the fast path's tolerance is 1e-3); measured ~1e-6
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def env():
    from pigeon_amd import _lib, hip_ops, synthetic
    from oracle import pigeon_oracle as orc
    _lib.require_gpu()
    return dict(lib=_lib, ops=hip_ops, syn=synthetic, orc=orc)


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def _report(lines, name):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", name), "w") as f:
        f.write("\n".join(lines) + "\n")


def test_split_gemm_vs_fp64(env, capsys):
    """[hi | lo | hi 2^-8] x [Wh | Wh | Wl 2^8] through the persistent fp16 MFMA kernel == the fp64 product to fp32 accuracy,
    including operands whose low halves are fp16 SUBNORMALS (|x| < 0.12: lo < 6.1e-5) and rows of mixed magnitude."""
    ops, lib = env["ops"], env["lib"]
    g = torch.Generator().manual_seed(5)
    M, K, N = 1000, 1024, 512
    A = torch.randn((M, K), generator=g)
    A[:, ::3] *= 0.01                      # a third of the columns two orders of magnitude down: their lo halves are subnormal
    A[::7] *= 30.0
    W = torch.randn((N, K), generator=g) * 0.02
    bias = torch.randn((N,), generator=g)
    ref = (A.double() @ W.double().t() + bias.double())
    A3 = ops.x3_split(A.to(DEV))
    W3 = ops.x3_pack_weight(W).to(DEV)
    out = torch.empty((M, N), dtype=torch.float32, device=DEV)
    ops.gemm16(A3, W3, bias.to(DEV), out, lib.EPI_F32, variant=36)
    e3 = _rel(out.cpu(), ref)
    # the plain fp16 product of the same operands, for scale
    out16 = torch.empty((M, N), dtype=torch.float32, device=DEV)
    ops.gemm16(A.to(DEV).half().contiguous(), W.to(DEV).half().contiguous(), bias.to(DEV), out16, lib.EPI_F32, variant=36)
    e1 = _rel(out16.cpu(), ref)
    e32 = _rel((A @ W.t() + bias), ref)
    # the triple reconstructs the fp32 value to 2^-22
    t = A3.cpu().float()
    rec = t[:, :K] + t[:, K:2 * K]
    # ... or, below fp16's normal range, to half the subnormal quantum 2^-24
    erec = float(((rec.double() - A.double()).abs() / A.double().abs().clamp_min(2.0 ** -4)).max())
    with capsys.disabled():
        print(f"\nsplit GEMM vs fp64: triple {e3:.2e}, plain fp16 {e1:.2e}, torch fp32 on the CPU {e32:.2e}; hi+lo reconstructs x to {erec:.2e}")
    assert torch.equal(t[:, 2 * K:], (t[:, :K] * (1.0 / 256)).half().float())
    assert erec < 2.0 ** -20
    assert e3 < 2e-6, "the split-fp16 GEMM must be fp32-grade (are fp16 subnormal operands flushed by the MFMA?)"
    assert e3 < e1 / 100


def test_gemm_parts_one_launch(env):
    """PgGemmExtra::parts (round 5): S products in ONE persistent launch.  Every part must equal -- bit for bit -- the same kernel
    launched on its column slice alone (same tiles, same K order), the bias must ride in part 0 only, and the fixed-order sum of the
    parts of a triple GEMM must be the fp32-grade product; ragged M, more tiles than CUs, S = 3 and 6."""
    ops, lib = env["ops"], env["lib"]
    g = torch.Generator().manual_seed(15)
    for (M, N, Kp, S) in ((1000, 512, 256, 3), (2308, 1024, 512, 6), (70 * 256 + 19, 1024, 128, 3)):
        A = torch.randn((M, S * Kp), generator=g).half().to(DEV)
        W = (torch.randn((N, S * Kp), generator=g) * 0.05).half().to(DEV)
        bias = torch.randn((N,), generator=g).to(DEV)
        parts = ops.gemm16_parts(A, W, bias, S)
        for p in range(S):
            one = torch.empty((M, N), dtype=torch.float32, device=DEV)
            ops.gemm16(A[:, p * Kp:(p + 1) * Kp], W[:, p * Kp:(p + 1) * Kp], bias if p == 0 else None, one, lib.EPI_F32, variant=36)
            assert torch.equal(parts[p], one), (M, N, Kp, S, p)
        ref = A.double() @ W.double().t() + bias.double()
        assert _rel(parts.double().sum(dim=0).cpu(), ref.cpu()) < 1e-6


def test_x3_layernorm_and_gelu(env):
    ops = env["ops"]
    g = torch.Generator().manual_seed(6)
    x = torch.randn((77, 1024), generator=g) * 3 + 0.5
    gam = 1 + 0.1 * torch.randn((1024,), generator=g)
    bet = 0.05 * torch.randn((1024,), generator=g)
    t = ops.x3_layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV)).cpu().float()
    ref = F.layer_norm(x.double(), (1024,), gam.double(), bet.double(), 1e-5)
    assert _rel(t[:, :1024] + t[:, 1024:2048], ref) < 1e-6
    h = torch.randn((33, 4096), generator=g) * 2
    t = ops.x3_split(h.to(DEV), gelu=True).cpu().float()
    ref = h.double() * torch.sigmoid(1.702 * h.double())
    assert _rel(t[:, :4096] + t[:, 4096:8192], ref) < 1e-6


def test_gelu_x3_epilogue_equals_f32_epilogue_plus_split_kernel(env):
    """EPI_GELU_X3 (round 6: the exact mode's fc1 writes the triple fc2 reads from its own epilogue) against the form it replaces --
    EPI_F32 into an fp32 buffer, then split_x3_kernel<true> -- BIT for bit: ragged M (a last row panel of 19 rows, one of 1 row),
    more tiles than CUs, fc1's own shape; the guard rows behind row M and nothing else are left alone."""
    ops, lib = env["ops"], env["lib"]
    g = torch.Generator().manual_seed(41)
    for (M, N, K) in ((1000, 512, 256), (70 * 256 + 19, 1024, 128), (2 * 577, 4096, 3072), (257, 256, 384)):
        A = torch.randn((M, K), generator=g).half().to(DEV)
        W = (torch.randn((N, K), generator=g) * 0.05).half().to(DEV)
        bias = torch.randn((N,), generator=g).to(DEV)
        f32 = torch.empty((M, N), dtype=torch.float32, device=DEV)
        ops.gemm16(A, W, bias, f32, lib.EPI_F32, variant=36)
        want = ops.x3_split(f32, gelu=True)
        got = torch.full((M + 3, 3 * N), 7.0, dtype=torch.float16, device=DEV)
        ops.gemm16(A, W, bias, got, lib.EPI_GELU_X3, variant=36, M=M)
        torch.cuda.synchronize()
        assert torch.equal(got[:M].view(torch.int16), want.view(torch.int16)), (M, N, K)
        assert bool((got[M:].float() == 7.0).all())
        # and it is the triple of QuickGELU(A W^T + b) to fp32 grade
        ref = A.double() @ W.double().t() + bias.double()
        ref = ref * torch.sigmoid(1.702 * ref)
        t = got[:M].float().double()
        assert _rel((t[:, :N] + t[:, N:2 * N]).cpu(), ref.cpu()) < 1e-6
    # shapes / operands the epilogue does not exist for are refused, not mis-run
    bad = torch.empty((8, 3 * 512), dtype=torch.float16, device=DEV)
    with pytest.raises(lib.PigeonHipError):                                  # row stride 2 N instead of 3 N
        ops.gemm16(A[:8, :256].contiguous(), W[:512, :256].contiguous(), bias[:512].contiguous(),
                   torch.empty((8, 2 * 512), dtype=torch.float16, device=DEV), lib.EPI_GELU_X3, variant=36)
    with pytest.raises(lib.PigeonHipError):
        ops.gemm16(A[:8, :256].bfloat16().contiguous(), W[:512, :256].bfloat16().contiguous(), bias[:512].contiguous(), bad.bfloat16(), lib.EPI_GELU_X3, variant=36)


def test_exact_pass_fusion_changes_no_bit(env):
    """pg_tune_exact_fusion: the exact pass with its activation splits inside their producers (attention_x3_kernel<true>, EPI_GELU_X3)
    against the unfused form (fp32 buffers + split_x3 launches) -- embeddings AND the last hidden state bit for bit, at a batch whose fc1
    runs as one persistent launch (the fused epilogue), and at one whose GEMMs go through the small-batch kernel (only the attention is
    fused there)."""
    ops, syn = env["ops"], env["syn"]
    enc = ops.VitEncoder(syn.make_vit_weights(seed=5, layers=3), layers=3, precise=True)
    g = torch.Generator().manual_seed(9)
    try:
        for n in (30, 2):
            px = torch.randn((n, 3, 336, 336), generator=g).to(DEV)
            ops.tune_exact_fusion(False)
            e0, h0 = enc.forward_precise(px, return_hidden=True)
            ops.tune_exact_fusion(True)
            e1, h1 = enc.forward_precise(px, return_hidden=True)
            torch.cuda.synchronize()
            assert torch.equal(e0, e1) and torch.equal(h0, h1), n
    finally:
        ops.tune_exact_fusion(True)


@pytest.mark.parametrize("kernel", ["split_fp16", "fp32_mfma"])
def test_attention_f32_vs_fp64(env, capsys, kernel):
    """The exact mode's attention against fp64: round 5's kernel on split-fp16 operands (v_mfma_f32_32x32x16_f16, three partial
    products per product) and round 4's on the fp32 MFMA (the A/B arm, pg_tune_exact_attention(1))."""
    ops = env["ops"]
    g = torch.Generator().manual_seed(7)
    n = 2
    qkv = torch.randn((n * 577, 3072), generator=g)
    qkv[:, :1024] *= 3.0                   # score spread ~ +-25: peaky rows next to flat ones
    qkv[:577, 1024:2048] *= 0.2
    env["lib"].check(env["lib"].load().pg_tune_exact_attention(1 if kernel == "fp32_mfma" else 0), "pg_tune_exact_attention")
    try:
        out = ops.attention_f32(qkv.to(DEV).contiguous(), n).cpu()
    finally:
        env["lib"].load().pg_tune_exact_attention(0)
    q, k, v = [qkv[:, i * 1024:(i + 1) * 1024].double().reshape(n, 577, 16, 64).transpose(1, 2) for i in range(3)]
    p = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1)
    ref = (p @ v).transpose(1, 2).reshape(n * 577, 1024)
    e = _rel(out, ref)
    worst = float((out.double() - ref).abs().max() / ref.abs().max())
    with capsys.disabled():
        print(f"\nattention_f32 [{kernel}] vs fp64: rel {e:.2e}, worst element / max {worst:.2e}")
    assert e < 3e-6 and worst < 1e-5

def test_head_margin_vs_torch(env):
    ops, syn = env["ops"], env["syn"]
    g = torch.Generator().manual_seed(8)
    B, C = 37, 3001
    emb = torch.randn((B, 4, 1024), generator=g)
    W, b = syn.make_head_weights(C, seed=4)
    W = W * 8
    cent = torch.from_numpy(syn.make_geocells(C, seed=1))
    o = ops.head_forward(emb.to(DEV), W.to(DEV), b.to(DEV), cent.to(DEV), 5)
    margin, sens, top2 = ops.head_margin(o["logits"], emb.to(DEV), W.to(DEV))
    lg = o["logits"].cpu()
    t2 = torch.topk(lg, 2, dim=-1)
    assert torch.equal(top2.cpu(), t2.indices[:, 1]) and torch.equal(o["preds_geocell"].cpu(), t2.indices[:, 0])
    assert torch.equal(margin.cpu(), t2.values[:, 0] - t2.values[:, 1])
    pe = emb.mean(dim=1)
    want = pe.norm(dim=1) * (W[t2.indices[:, 0]] - W[t2.indices[:, 1]]).norm(dim=1) / 32.0
    assert torch.allclose(sens.cpu(), want, rtol=1e-5)
    # one geocell: nothing to be uncertain about
    o1 = ops.head_forward(emb.to(DEV), W[:1].contiguous().to(DEV), b[:1].to(DEV), cent[:1].to(DEV), 1)
    m1, s1, i1 = ops.head_margin(o1["logits"], emb.to(DEV), W[:1].contiguous().to(DEV))
    assert torch.isinf(m1).all() and (s1 == 0).all() and (i1 == 0).all()


@pytest.mark.parametrize("name", ["vit2", "vit24"])
def test_precise_encoder_vs_reference_golden(env, golden_dir, name, capsys):
    """pg_vit_forward_precise against the REAL reference's embeddings: 100x tighter than the fast path's tolerance."""
    ops, syn = env["ops"], env["syn"]
    gz = np.load(os.path.join(golden_dir, f"{name}.npz"))
    wseed, layers, jitter, n, pseed = [int(x) for x in gz["meta"]]
    sd = syn.make_vit_weights(seed=wseed, layers=layers, affine_jitter=bool(jitter))
    enc = ops.VitEncoder(sd, layers=layers, precise=True)
    px = syn.make_pixels(n, seed=pseed).to(DEV)
    ref = torch.from_numpy(gz["embedding"])
    fast = enc(px).cpu()
    emb, hid = enc.forward_precise(px, return_hidden=True)
    e_fast, e_exact = _rel(fast, ref), _rel(emb.cpu(), ref)
    with capsys.disabled():
        print(f"\n{name}: embedding rel err fast {e_fast:.2e}, exact mode {e_exact:.2e}")
    if "lhs_rows" in gz.files:
        rows = hid.cpu()[:, [0, 1, 2, 288, 575, 576]]
        assert _rel(rows, torch.from_numpy(gz["lhs_rows"])) < 10 * EXACT_TOL
    assert e_exact < EXACT_TOL
    assert e_exact < e_fast / 20
    # the same batch again: the same bits.  Another batch SIZE may cut its GEMMs into another number of K-parts (vit.hip
    # precise_parts: a pure function of the shape), i.e. another fp32 summation order: equal to ~1e-7, two orders below the
    # exact tier's own floor (rel_tol_exact 5e-6)
    assert torch.equal(enc.forward_precise(px).cpu(), emb.cpu())
    again = enc.forward_precise(px[1:3].contiguous()).cpu()
    assert _rel(again, emb.cpu()[1:3]) < 1e-6
    with pytest.raises(env["lib"].PigeonHipError):
        ops.VitEncoder(sd, layers=layers).forward_precise(px)
    enc.close()


def test_encoder_graph_replay_bit_identical(env, capsys):
    """pg_vit_forward replays the encoder body from a captured hipGraph from the second forward of a (workspace, n) key on:
    same kernels, same order -> the same bits as the eager launches; profiling switches the replay off for the call."""
    ops, syn = env["ops"], env["syn"]
    sd = syn.make_vit_weights(seed=11, layers=2, affine_jitter=True)
    enc = ops.VitEncoder(sd, layers=2)
    px = syn.make_pixels(24, seed=5).to(DEV)
    enc.graph(False)
    eager = enc(px).clone()
    enc.graph(True)
    r0, c0 = enc.graph()
    outs = [enc(px).clone() for _ in range(4)]          # 1st: eager (first sight of the key), 2nd: capture + launch, then replays
    torch.cuda.synchronize()
    r1, c1 = enc.graph()
    with capsys.disabled():
        print(f"\nencoder graph: {r1 - r0} replays, {c1 - c0} capture(s) in 4 forwards")
    assert c1 - c0 == 1 and r1 - r0 == 3, "the graph path did not engage (capture unsupported on this runtime?)"
    for o in outs:
        assert torch.equal(o, eager)
    other = enc(px[:8].contiguous())                     # another key: eager again, same rows
    assert torch.equal(other, eager[:8])
    enc.profile_enable(True)
    prof = enc(px).clone()
    enc.profile_enable(False)
    assert torch.equal(prof, eager) and enc.graph()[0] == r1     # bracketed with events: no replay
    assert enc.profile_read()["gemm_fc1"][0] == 2
    enc.close()


def test_exact_mode_edges(env, tmp_path, capsys):
    """The exact mode at the edges of its inputs: single-image (non-panorama) models, fp16 pixels as the GPU preprocessing writes
    them, bf16 fast-path operands (the exact pass always uses fp16 halves), batches in which nothing / everything is re-encoded,
    the serving tuple, and the auto-calibration of the certainty bound."""
    from pigeon_amd.clip_embedder import HipCLIPVisionModel
    from pigeon_amd.super_guessr import SuperGuessr
    syn, ops = env["syn"], env["ops"]
    C = 300
    gp = os.path.join(str(tmp_path), "g.csv")
    syn.write_geocell_csv(gp, syn.make_geocells(C, seed=0))
    sd = syn.make_vit_weights(seed=11, layers=2, affine_jitter=True)
    vit = HipCLIPVisionModel(sd, layers=2).to(DEV)
    W, b = syn.make_head_weights(C, seed=1)

    def model(**kw):
        m = SuperGuessr(vit, freeze_base=True, num_candidates=5, geocell_path=gp, **kw)
        with torch.no_grad():
            m.cell_layer.weight.copy_(W * 64); m.cell_layer.bias.copy_(b)
        return m.to(DEV).eval()
    px = syn.make_pixels(12, seed=3).to(DEV)
    # single images, everything uncertain (kappa huge) -> every sample re-encoded: embeddings == the exact encoder's, bit for bit
    m = model(panorama=False, exact_top1=True, margin_kappa=1e9, margin_autocalibrate=False)
    out = m(pixel_values=px, labels_clf=None)
    assert m.last_reencoded.numel() == 12
    want = vit._encoder(torch.device(DEV)).forward_precise(px)
    assert torch.equal(out.embedding, want)
    # ... nothing uncertain (kappa 0): the fast path's outputs, no exact pass
    m0 = model(panorama=False, exact_top1=True, margin_kappa=0.0, margin_autocalibrate=False)
    out0 = m0(pixel_values=px, labels_clf=None)
    assert m0.last_reencoded.numel() == 0 and torch.equal(out0.embedding, vit.embed(px)) and bool(m0.last_certain.all())
    # the exact mode is the product default; PIGEON_EXACT_TOP1=0 / exact_top1=False is the opt-out, which still reports certainty
    assert model(panorama=False).exact_top1 is True
    mf = model(panorama=False, exact_top1=False, margin_kappa=1e9)
    outf = mf(pixel_values=px, labels_clf=None)
    assert mf.last_reencoded.numel() == 0 and torch.equal(outf.embedding, vit.embed(px)) and not bool(mf.last_certain.any())
    # panoramas with fp16 pixels (what pg_prep_forward hands over) + the serving tuple
    mp = model(panorama=True, serving=True, exact_top1=True, margin_kappa=1e9, margin_autocalibrate=False)
    px16 = px.half().reshape(3, 12, 336, 336)
    llh, topk, emb = mp(pixel_values=px16)
    assert emb.shape == (3, 4, 1024) and mp.last_reencoded.tolist() == [0, 1, 2]
    assert torch.equal(emb.reshape(12, 1024), vit._encoder(torch.device(DEV)).forward_precise(px.half()))
    # exactness claim on the panorama path: agrees with the oracle's embedding of the SAME (fp16-rounded) pixels to 1e-5
    ref = env["orc"].clip_embedding(sd, px.half().float().cpu())
    assert _rel(emb.reshape(12, 1024).cpu(), ref) < EXACT_TOL
    # calibration: explicit, or by the first forward that sees >= 8 samples; frozen afterwards (later batches do not move it)
    assert not m.certainty.calibrated and m.margin_rel_tol == 1e-3
    mc = model(panorama=False, exact_top1=True)
    mc(pixel_values=px, labels_clf=None)
    st = dict(mc.certainty.stats)
    assert mc.certainty.calibrated and st["samples"] == 12 and 1e-5 < mc.margin_rel_tol < 1e-3 and 1e-4 < st["fast_vs_exact_rms"] < 5e-4
    mc(pixel_values=px.flip(0).contiguous(), labels_clf=None)
    assert mc.certainty.stats == st
    rms = mc.calibrate_certainty(px[:8])
    assert mc.certainty.stats["samples"] == 8 and abs(rms - mc.certainty.stats["fast_vs_exact_rms"]) < 1e-12
    # a server's traffic: one sample per call -> collected, calibrated once 16 have been seen, nothing kept afterwards
    ms = model(panorama=False, exact_top1=True)
    for i in range(15):
        ms(pixel_values=px[i % 12:i % 12 + 1], labels_clf=None)
    assert not ms.certainty.calibrated and len(ms._cal_buffer) == 15
    ms(pixel_values=px[3:4], labels_clf=None)
    assert ms.certainty.calibrated and ms.certainty.stats["samples"] == 16 and ms._cal_buffer == []
    # bf16 operands on the fast path: the exact pass is unaffected
    vb = HipCLIPVisionModel(sd, layers=2).to(DEV)
    vb.enable_precise(True)
    os.environ["PIGEON_MMA_DTYPE"] = "bf16"
    try:
        eb = vb.embed_precise(px[:4])
        assert vb._encoder(torch.device(DEV)).mma_dtype == "bf16"
    finally:
        del os.environ["PIGEON_MMA_DTYPE"]
    assert _rel(eb, want[:4]) < 1e-6                              # (another batch size: another K-part count, see precise_parts)
    # B = 0 and B = 1
    e0 = m(pixel_values=px[:0], labels_clf=None)
    assert e0.embedding.shape[0] == 0 and m.last_certain.numel() == 0
    e1 = m(pixel_values=px[:1], labels_clf=None)
    assert _rel(e1.embedding, want[:1]) < 1e-6


@pytest.mark.parametrize("tower", ["trained_like+spread", "all_heads_high_gain"])
def test_stress_towers_vs_reference_module(env, capsys, tower):
    """Stress regimes at 24 layers against the reference's own module (transformers.CLIPVisionModel, fp32, eager attention, stock
    PyTorch-ROCm on this GPU; never part of the product) -- no committed fixture, the checker is computed here:
      trained_like+spread   massive activations / rows with |mean| >> std (make_vit_weights_trained_like) AND input-selected global
                            attention in a quarter of the heads (make_vit_weights_spread);
      all_heads_high_gain   (round 5) large q.k gain and jittered affine parameters on ALL 16 heads of every layer: the regime in
                            which 16-bit Q / K operands cost the most.
    Fast path: within the 1e-3 contract PER IMAGE, no fp16 range alarm; exact mode within EXACT_TOL."""
    from transformers import CLIPVisionConfig, CLIPVisionModel
    ops, syn = env["ops"], env["syn"]
    if tower == "trained_like+spread":
        sd = syn.make_vit_weights_spread(seed=31, layers=24, base=syn.make_vit_weights_trained_like(seed=21, layers=24))
    else:
        sd = syn.make_vit_weights_spread(seed=47, layers=24, heads_frac=1.0)
    cfg = CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=336,
                           patch_size=14, projection_dim=768)
    hf = CLIPVisionModel._from_config(cfg, attn_implementation="eager")
    hf.load_state_dict(sd, strict=True)
    hf = hf.to(DEV).eval()
    torch.backends.cuda.matmul.allow_tf32 = False
    px = syn.make_pixels(16, seed=2718).to(DEV)
    with torch.no_grad():
        hid = hf(pixel_values=px).last_hidden_state
    ref = hid.mean(dim=1).cpu()
    absmax = float(hid.abs().max())
    ie = ref / ref.norm(dim=1, keepdim=True)
    cs = (ie @ ie.t())[~torch.eye(16, dtype=torch.bool)]
    del hf, hid
    torch.cuda.empty_cache()
    enc = ops.VitEncoder(sd, layers=24, precise=True)
    fast = enc(px).cpu()
    alarm = enc.range_alarm_read()
    exact = enc.forward_precise(px).cpu()
    e_fast, e_exact = _rel(fast, ref), _rel(exact, ref)
    worst = float(((fast.double() - ref.double()).norm(dim=1) / ref.double().norm(dim=1)).max())
    line = (f"{tower}: |hidden| max {absmax:.0f}, image cos-sim mean {float(cs.mean()):.2f} max {float(cs.max()):.2f}; "
            f"embedding rel err fast {e_fast:.2e} (worst image {worst:.2e}), exact {e_exact:.2e}; fp16 range alarm rows {alarm}")
    with capsys.disabled():
        print("\n" + line)
    _report([line], f"stress_tower_{tower.replace('+', '_')}.txt")
    enc.close()
    assert alarm == 0
    assert e_exact < EXACT_TOL
    if tower == "trained_like+spread":
        assert e_fast < 1e-3 and worst < 1e-3, "the fast path leaves the 1e-3 contract on this tower (per image)"
        return
    # all heads at high gain: the 16-bit path itself is at / beyond the contract (measured 9.5e-4 overall, worst image 1.1e-3).  The
    # PRODUCT must notice: the calibration (on other images) measures it and sends every sample through the exact encoder.
    from pigeon_amd.clip_embedder import HipCLIPVisionModel
    from pigeon_amd.super_guessr import SuperGuessr
    import tempfile
    gp = os.path.join(tempfile.mkdtemp(prefix="pigeon_stress_"), "g.csv")
    syn.write_geocell_csv(gp, syn.make_geocells(300, seed=0))
    model = SuperGuessr(HipCLIPVisionModel(sd, layers=24).to(DEV), freeze_base=True, num_candidates=5, geocell_path=gp).to(DEV).eval()
    model.calibrate_certainty(syn.make_pixels(32, seed=99).to(DEV))
    st = model.certainty.stats
    out = model(pixel_values=px, labels_clf=None)
    worst_p = float(((out.embedding.cpu().double() - ref.double()).norm(dim=1) / ref.double().norm(dim=1)).max())
    with capsys.disabled():
        print(f"   product on this tower: calibration per image {st['image_rel_err']:.2e} (worst {st['worst_image_rel_err']:.2e}) -> force_exact "
              f"{st['force_exact']}; embeddings returned: worst image {worst_p:.2e}, re-encoded {model.last_reencoded.numel()}/16")
    assert st["force_exact"] and model.last_reencoded.numel() == 16 and worst_p < EXACT_TOL
    # ... and so must the EMBED path (`run.py embed` -> CLIPEmbedding, reference models/clip_embedder.py:63-65,
    # preprocessing/embed.py:16-43): its first batch measures the 16-bit encoder against the exact one and, outside the contract,
    # everything it writes comes from the exact encoder (round 6)
    from pigeon_amd.clip_embedder import CLIPEmbedding
    emb = CLIPEmbedding("random", device=DEV, clip_model=model.base_model)
    got = emb(px).cpu()
    worst_e = float(((got.double() - ref.double()).norm(dim=1) / ref.double().norm(dim=1)).max())
    with capsys.disabled():
        print(f"   CLIPEmbedding on this tower: guard {emb.guard_stats}; embeddings returned: worst image {worst_e:.2e}")
    assert emb.guard_stats["outside"] and emb.force_exact and worst_e < EXACT_TOL
    with pytest.raises(RuntimeError, match="outside the 1e-3 embedding contract"):
        CLIPEmbedding("random", device=DEV, clip_model=model.base_model, contract_guard="raise")(px)
    off = CLIPEmbedding("random", device=DEV, clip_model=model.base_model, contract_guard="off")
    assert torch.equal(off(px).cpu(), fast) and off.guard_stats is None
