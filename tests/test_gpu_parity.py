"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI behind the reference's
class surface, against (a) the golden vectors produced by the REAL reference (tests/golden, oracle/make_golden.py)
and (b) the CPU oracle on fresh seeded inputs.

Tolerances (BASELINE.json north_star): geocell argmax / top-k indices / refined geocell bit-exact, refined (lng,lat)
exact (they are picked from a discrete set), embeddings within 1e-3 relative (||a-b||/||b||, whole tensor and worst
row).  The MFMA operands are fp16 (default); the bf16 variant is checked against its own measured floor.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

EMB_TOL = 1e-3          # north_star: embeddings within 1e-3 relative
DEV = "cuda"


@pytest.fixture(scope="module")
def env():
    from pigeon_amd import _lib, hip_ops, synthetic
    from oracle import pigeon_oracle as orc
    _lib.require_gpu()                       # fails loudly if the HIP library / GPU is missing -- no fallback
    return dict(lib=_lib, ops=hip_ops, syn=synthetic, orc=orc)


@pytest.fixture(scope="module")
def vit2(env):
    from pigeon_amd.clip_embedder import HipCLIPVisionModel
    sd = env["syn"].make_vit_weights(seed=11, layers=2, affine_jitter=True)
    return sd, HipCLIPVisionModel(sd, layers=2).to(DEV)


@pytest.fixture(scope="module")
def vit24(env):
    from pigeon_amd.clip_embedder import HipCLIPVisionModel
    sd = env["syn"].make_vit_weights(seed=0, layers=24)
    return sd, HipCLIPVisionModel(sd, layers=24).to(DEV)


def _gold(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _geocells_csv(tmp_path, C, seed=0):
    from pigeon_amd import synthetic
    p = os.path.join(str(tmp_path), f"geocells_{C}.csv")
    synthetic.write_geocell_csv(p, synthetic.make_geocells(C, seed=seed))
    return p


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("variant,K", [(0, 384), (8, 320), (33, 384), (36, 640), (56, 640)])
def test_gemm_epilogues(env, dt, variant, K):
    ops, L = env["ops"], env["lib"]
    g = torch.Generator().manual_seed(3)
    M, N = 1154 + 37, 512                               # ragged M tail; K = 5 tiles (odd) or 6 / 10 (persistent kernel)
    A = torch.randn((M, K), generator=g).to(dt).to(DEV)
    W = (torch.randn((N, K), generator=g) * 0.05).to(dt).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    acc = (A.float().cpu() @ W.float().cpu().T)         # plain fp32 reference on the same rounded operands
    out = torch.zeros((M, N), device=DEV)
    ops.gemm16(A, W, bias, out, L.EPI_F32, variant=variant)
    assert torch.allclose(out.cpu(), acc + bias.cpu(), rtol=1e-5, atol=1e-4)
    o16 = torch.zeros((M, N), dtype=dt, device=DEV)
    ops.gemm16(A, W, bias, o16, L.EPI_QKV, qscale=0.25, qcols=256, variant=variant)
    ref = acc + bias.cpu()
    ref[:, :256] *= 0.25
    ulp = 2.0 ** (-8 if dt == torch.bfloat16 else -11)
    assert (o16.float().cpu() - ref).abs().max() <= 2 * ref.abs().max() * ulp
    ops.gemm16(A, W, bias, o16, L.EPI_GELU, variant=variant)
    y = acc + bias.cpu()
    ref = y * torch.sigmoid(1.702 * y)
    assert (o16.float().cpu() - ref).abs().max() <= 2 * ref.abs().max() * ulp
    X0 = torch.randn((M, N), generator=g)
    X = X0.to(DEV).clone()
    ops.gemm16(A, W, bias, X, L.EPI_RESID, variant=variant)
    assert torch.allclose(X.cpu(), X0 + acc + bias.cpu(), rtol=1e-5, atol=1e-4)


MFMA16_VARIANTS = (33, 36, 56)          # kernels built on v_mfma_f32_16x16x32 (k = 32 per instruction); the others use 32x32x16


def test_gemm_persistent_many_tiles_bit_identical(env):
    """More output tiles than CUs (persistent blocks walk several tiles, the next tile's first K tile is prefetched
    under the epilogue), ragged M, padded leading dimensions: the persistent ping-pong kernel must reproduce the
    one-tile-per-block kernel (variant 8, v_mfma_f32_32x32x16) to fp32 rounding -- the persistent kernels use
    v_mfma_f32_16x16x32, which adds the k products of an instruction in another association -- must agree BIT for bit with
    each other where they share that instruction (33 == 36: same K order, different tile raster), and must not touch rows
    past M."""
    ops, L = env["ops"], env["lib"]
    g = torch.Generator().manual_seed(9)
    M, N, K = 70 * 256 + 19, 1024, 256
    A = torch.empty((M, K + 64), dtype=torch.float16, device=DEV)[:, :K]
    A.copy_(torch.randn((M, K), generator=g).to(torch.float16))
    W = (torch.randn((N, K), generator=g) * 0.05).to(torch.float16).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    X0 = torch.randn((M, N), generator=g).to(DEV)
    for epi in (L.EPI_QKV, L.EPI_GELU, L.EPI_RESID, L.EPI_F32):
        outs = []
        for var in (8, 33, 36, 56):                     # 56: 384 x 256 tiles for the 16-bit epilogues (47 row panels here)
            if epi in (L.EPI_QKV, L.EPI_GELU):
                o = torch.full((M + 3, N), 7.0, dtype=torch.float16, device=DEV)
            elif epi == L.EPI_RESID:
                # >= 384 guard rows: the wrapper then runs the kernel IN PLACE (no padded copy), so a write past row M would be seen
                o = torch.cat([X0, torch.full((387, N), 7.0, device=DEV)]).contiguous()
            else:
                o = torch.full((M + 3, N), 7.0, device=DEV)
            ops.gemm16(A, W, bias, o, epi, qscale=0.25, qcols=256, variant=var, M=M)
            outs.append(o)
        torch.cuda.synchronize()
        for var, o in zip((33, 36, 56), outs[1:]):
            if var in MFMA16_VARIANTS:                  # v_mfma_f32_16x16x32 sums an instruction's k products in another association
                a, b = outs[0][:M].float(), o[:M].float()
                assert bool(((a - b).abs() <= 1e-5 * a.abs().max() + 2e-3 * a.abs()).all()), f"epilogue {epi} variant {var}"
            else:
                assert torch.equal(outs[0], o), f"epilogue {epi} variant {var}"
            assert bool((o[M:].float() == 7.0).all())
        assert torch.equal(outs[1], outs[2]), f"epilogue {epi}: 33 vs 36"
        assert torch.equal(outs[2], outs[3]), f"epilogue {epi}: 36 vs 56"     # 384-row tiles: same instruction, same K order


def _gemm_all_epilogues(ops, L, A, W, bias, X0, cs, rs, M, variant):
    """Every epilogue of one (A, W) problem through `variant`; outputs carry 3 guard rows where the caller owns the buffer."""
    N = W.shape[0]
    outs = []
    for epi in (L.EPI_QKV, L.EPI_GELU, L.EPI_RESID, L.EPI_F32):
        if epi in (L.EPI_QKV, L.EPI_GELU):
            o = torch.full((M + 3, N), 7.0, dtype=A.dtype, device=DEV)
        elif epi == L.EPI_RESID:
            # >= 384 guard rows: the wrapper then runs the kernel IN PLACE (no padded copy), so a write past row M would be seen
            o = torch.cat([X0, torch.full((387, N), 7.0, device=DEV)]).contiguous()
        else:
            o = torch.full((M + 3, N), 7.0, device=DEV)
        ops.gemm16(A, W, bias, o, epi, qscale=0.25, qcols=256, variant=variant, M=M)
        outs.append(o)
    for epi in (L.EPI_QKV_LN, L.EPI_GELU_LN):
        outs.append(ops.gemm16_ln(A, W, bias, cs, rs, epi, qscale=0.25, qcols=256, variant=variant))
    X = X0.clone()
    x16, part = ops.gemm16_resid_stat(A, W, bias, X, variant=variant)
    outs += [X, x16, part]
    torch.cuda.synchronize()
    return outs


def _tail_problem(M, N, K, seed, dt=torch.float16):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn((M, K), generator=g).to(dt).to(DEV)
    W = (torch.randn((N, K), generator=g) * 0.05).to(dt).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    X0 = torch.randn((M, N), generator=g).to(DEV)
    rs = torch.stack([torch.rand(M, generator=g) + 0.5, torch.randn(M, generator=g)], dim=1).contiguous().to(DEV)
    cs = torch.randn(N, generator=g).to(DEV)
    return A, W, bias, X0, cs, rs


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_gemm_tail_kernel_bit_identical_to_persistent(env, dt):
    """gemm_tail.hip (variant 70 = a whole problem through it): 32 x 64 one-wave tiles, operands straight from L2.  Same MFMA
    chain over k and the same epilogue expressions as the persistent kernel, so every epilogue -- 16-bit, fp32, residual,
    LayerNorm-fold, residual + 16-bit copy + row statistics -- must come out BIT-identical, ragged M, guard rows untouched."""
    ops, L = env["ops"], env["lib"]
    M, N, K = 1000 + 13, 1024, 512
    A, W, bias, X0, cs, rs = _tail_problem(M, N, K, 21, dt)
    ref = _gemm_all_epilogues(ops, L, A, W, bias, X0, cs, rs, M, 36)
    got = _gemm_all_epilogues(ops, L, A, W, bias, X0, cs, rs, M, 70)
    names = ("qkv", "gelu", "resid", "f32", "qkv_ln", "gelu_ln", "resid_stat.X", "resid_stat.x16", "resid_stat.part")
    for n, a, b in zip(names, ref, got):
        assert torch.equal(a, b), n
    for o in got[:4]:
        assert bool((o[M:].float() == 7.0).all())


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_gemm_mid_kernel_bit_identical_to_persistent(env, dt):
    """gemm_mid.hip (round 6; variant 71 = a whole problem through it): 128 x 128 one-tile-per-block tiles through a 3-stage LDS
    ring, the kernel of batches too small to fill the persistent kernels.  Same MFMA chain over k, same slab geometry, same
    gemm_epi.h expressions: every epilogue must come out BIT-identical to the persistent kernel -- ragged M (a last row tile of 13
    rows), one K tile, two, three (the ring's wrap) and 64 of them (fc2's K), guard rows untouched."""
    ops, L = env["ops"], env["lib"]
    names = ("qkv", "gelu", "resid", "f32", "qkv_ln", "gelu_ln", "resid_stat.X", "resid_stat.x16", "resid_stat.part")
    for M, N, K, seed in ((1000 + 13, 1024, 512, 21), (2308, 3072, 1024, 22), (577, 1024, 4096, 23), (130, 1024, 128, 24), (64, 1024, 384, 25)):
        A, W, bias, X0, cs, rs = _tail_problem(M, N, K, seed, dt)
        ref = _gemm_all_epilogues(ops, L, A, W, bias, X0, cs, rs, M, 36)
        got = _gemm_all_epilogues(ops, L, A, W, bias, X0, cs, rs, M, 71)
        for n, a, b in zip(names, ref, got):
            assert torch.equal(a, b), (n, M, N, K)
        for o in got[:4]:
            assert bool((o[M:].float() == 7.0).all())


def test_gemm_tail_split_changes_nothing(env):
    """pg_gemm_launch cuts a problem whose tiles do not fill the persistent kernel's last round: whole rounds to the persistent
    kernel, the few rows beyond them to a small-tile kernel -- gemm_mid.hip where its model says it is the cheaper one (round 6: every
    K = 1024 GEMM of the model), gemm_tail.hip otherwise (fc2's K = 4096) or when gemm_mid is switched off.  With the split switched
    off (tail rows 0) the same call must give the same bits -- 256-row tiles (N = 1024: 776 tiles on 256 CUs -> 3 rounds = 49152 rows +
    300) and 384-row tiles (N = 3072: 1548 tiles -> 6 rounds = 49152 rows + 300), every epilogue (the residual + statistics one writes
    three row-indexed buffers through the moved bases), guard rows untouched.  (More rows than the small-batch routing of
    pg_gemm_launch looks at: a variant means its own kernel here.)"""
    ops, L = env["ops"], env["lib"]
    M = 192 * 256 + 300
    try:
        ops.tune_gemm_tail_shape(0, 0)                       # gemm_tail.hip for every shape gemm_mid.hip does not take
        for K in (256, 1024):
            for N, variant in ((1024, 36), (3072, 56)):
                A, W, bias, X0, cs, rs = _tail_problem(M, N, K, 22 + N + K)
                ops.tune_gemm_tail_rows(0)
                ref = _gemm_all_epilogues(ops, L, A, W, bias, X0, cs, rs, M, variant)
                ops.tune_gemm_tail_rows(768)
                for mid in (1, 0):
                    ops.tune_gemm_mid(mid)
                    got = _gemm_all_epilogues(ops, L, A, W, bias, X0, cs, rs, M, variant)
                    for i, (a, b) in enumerate(zip(ref, got)):
                        assert torch.equal(a, b), (N, K, variant, mid, i)
                    for o in got[:4]:
                        assert bool((o[M:].float() == 7.0).all())
    finally:
        ops.tune_gemm_mid(1)
        ops.tune_gemm_tail_rows(768)
        ops.tune_gemm_tail_shape(2048, 4096)


def test_small_batch_routing_changes_nothing(env):
    """pg_gemm_launch routes batches of up to ~64 images between the 384 x 256 kernel, the 256 x 256 kernel and gemm_mid.hip by a cost
    model of how their row panels fill rounds of the CUs (round 6).  A routing decision must never be a numerical one: with the
    routing off (pg_tune_gemm_mid(0): the variant's own kernel) every epilogue gives the same bits -- QKV's shape at 16 images (the
    model sends it to the 256 x 256 kernel), fc1's at one panorama (256 x 256), fc2's at 16 images (256 x 256 instead of 384 x 256),
    QKV's at one image (gemm_mid)."""
    ops, L = env["ops"], env["lib"]
    try:
        for M, N, K in ((16 * 577, 3072, 1024), (4 * 577, 4096, 1024), (16 * 577, 1024, 4096), (577, 3072, 1024)):
            A, W, bias, X0, cs, rs = _tail_problem(M, N, K, 77 + N + M)
            ops.tune_gemm_mid(0)
            ref = _gemm_all_epilogues(ops, L, A, W, bias, X0, cs, rs, M, 56)
            ops.tune_gemm_mid(1)
            got = _gemm_all_epilogues(ops, L, A, W, bias, X0, cs, rs, M, 56)
            for i, (a, b) in enumerate(zip(ref, got)):
                assert torch.equal(a, b), (M, N, K, i)
            for o in got[:4]:
                assert bool((o[M:].float() == 7.0).all())
    finally:
        ops.tune_gemm_mid(1)


def test_gemm_raster_knob_changes_nothing(env):
    """pg_tune_gemm_raster (round 5): which N tiles an XCD round of the 384 x 256 kernel walks is a RASTER choice -- groups of 4 (default),
    all of them (-1), 2, 6, and 8 (which does not divide fc1's 12 column tiles and is therefore ignored there) -- and must not change a bit
    of any epilogue's output, many rounds and a ragged M tail included."""
    ops, L = env["ops"], env["lib"]
    A, W, bias, X0, cs, rs = _tail_problem(2 * 384 * 70 + 211, 3072, 1024, seed=5)      # 141 row panels x 12 column tiles: 6.6 rounds
    # the residual epilogue runs on the 384 x 256 kernel for K >= 2048 (fc2's shape class): K = 2048, 4 column tiles (gn = 2 regroups them)
    A2, W2 = torch.cat([A, A], dim=1).contiguous(), torch.cat([W[:1024], W[1024:2048]], dim=1).contiguous()
    ref = None
    try:
        for gn in (0, -1, 2, 6, 8):
            ops.tune_gemm_raster(gn)
            outs = [ops.gemm16_ln(A, W, bias, cs, rs, epi, qscale=0.25, qcols=1024, variant=56) for epi in (L.EPI_QKV_LN, L.EPI_GELU_LN)]
            X = torch.cat([X0[:, :1024], torch.full((389, 1024), 7.0, device=DEV)]).contiguous()
            x16, part = ops.gemm16_resid_stat(A2, W2, bias[:1024].contiguous(), X[:A.shape[0]], variant=56)
            torch.cuda.synchronize()
            outs += [X.clone(), x16, part]
            if ref is None:
                ref = outs
            else:
                for a, b in zip(ref, outs):
                    assert torch.equal(a, b), gn
    finally:
        ops.tune_gemm_raster(0)


def test_gemm_resid_stat_on_384_row_tiles_bit_identical(env):
    """Round 3: fc2 (K >= 2048) runs its fp32-residual + 16-bit-copy + row-statistics epilogue on the 384 x 256 kernel
    (gemm_pp6.hip epilogue6_resid).  Same MFMA chain, same epilogue arithmetic (gemm_epi.h), same split-halves geometry: the new
    residual rows, their 16-bit copy and the (sum, sum of squares) partials must equal the 256 x 256 kernel's bit for bit --
    several persistent rounds, a ragged M tail, guard rows untouched, with the tail split on and off."""
    ops, L = env["ops"], env["lib"]
    g = torch.Generator().manual_seed(77)
    M, N, K = 3 * 256 * 96 + 211, 1024, 2048                  # 193 row panels of 384 x 4 column tiles = 772 tiles: 3 rounds + a tail
    A = torch.randn((M, K), generator=g).to(torch.float16).to(DEV)
    W = (torch.randn((N, K), generator=g) * 0.03).to(torch.float16).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    # 389 guard rows past M (>= 384: gemm16_resid_stat then runs in place on this very buffer, so the guard check below is real)
    X0 = torch.cat([torch.randn((M, N), generator=g), torch.full((389, N), 7.0)]).to(DEV)
    res = {}
    try:
        for tail in (0, 768):
            ops.tune_gemm_tail_rows(tail)
            for v in (36, 56):
                X = X0.clone()
                x16, part = ops.gemm16_resid_stat(A, W, bias, X[:M], variant=v)
                torch.cuda.synchronize()
                assert bool((X[M:] == 7.0).all()), (tail, v)
                res[(tail, v)] = (X[:M].clone(), x16.clone(), part.clone())
    finally:
        ops.tune_gemm_tail_rows(768)
    ref = res[(0, 36)]
    for key, got in res.items():
        for name, a, b in zip(("X", "x16", "statpart"), ref, got):
            assert torch.equal(a, b), (key, name)
    # and the values themselves: X = X0 + A W^T + b on the rounded operands
    want = X0[:2048] + A[:2048].float() @ W.float().T + bias
    assert torch.allclose(ref[0][:2048], want, rtol=2e-3, atol=2e-3)


def test_gemm_identity_is_not_transposed(env):
    ops, L = env["ops"], env["lib"]
    A = torch.eye(256).to(torch.float16).to(DEV)
    W = (torch.arange(256)[:, None] * 1.0 + torch.arange(256)[None, :] * 0.001).to(torch.float16).to(DEV)   # asymmetric
    out = torch.zeros((256, 256), device=DEV)
    ops.gemm16(A, W, None, out, L.EPI_F32)
    assert torch.equal(out.cpu(), W.float().cpu().T)


def test_rowops(env):
    ops = env["ops"]
    g = torch.Generator().manual_seed(5)
    x = torch.randn((1003, 1024), generator=g) * 3 + 0.5
    gam, bet = torch.randn(1024, generator=g) * 0.1 + 1, torch.randn(1024, generator=g) * 0.1
    ref = torch.nn.functional.layer_norm(x, (1024,), gam, bet, 1e-5)
    y = ops.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV), out_dtype=torch.float32)
    assert torch.allclose(y.cpu(), ref, rtol=1e-5, atol=2e-6)
    y16 = ops.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV), out_dtype=torch.float16)
    assert (y16.float().cpu() - ref).abs().max() <= 2.0 ** -9 * ref.abs().max()
    px = torch.randn((3, 3, 336, 336), generator=g)
    col = ops.im2col(px.to(DEV), torch.float16).cpu()
    refc = torch.nn.functional.unfold(px, kernel_size=14, stride=14).transpose(1, 2).reshape(3 * 576, 588)
    assert torch.equal(col[:, :588], refc.to(torch.float16)) and bool((col[:, 588:] == 0).all())
    h = torch.randn((5, 577, 1024), generator=g)
    assert torch.allclose(ops.token_mean(h.to(DEV)).cpu(), h.mean(dim=1), rtol=1e-5, atol=1e-6)
    z = torch.randn(100003, generator=g) * 100
    assert torch.equal(ops.cast_f32(z.to(DEV), torch.float16).cpu(), z.to(torch.float16))
    assert torch.equal(ops.cast_f32(z.to(DEV), torch.bfloat16).cpu(), z.to(torch.bfloat16))


@pytest.mark.parametrize("dt,tol", [(torch.float16, 6e-4), (torch.bfloat16, 4e-3)])
def test_attention_vs_fp32(env, dt, tol):
    ops = env["ops"]
    n = 2
    g = torch.Generator().manual_seed(21)
    qkv = torch.randn((n * 577, 3072), generator=g)
    qkv[:, :1024] *= 0.125 * 1.4426950408889634 * 2.0     # Q carries log2(e)/8 (x2: sharper softmax)
    qkv[300, 1024:2048] *= 30                              # one spiky key: forces an online-softmax rescale
    qkv = qkv.to(dt)
    out = ops.attention(qkv.to(DEV), n).float().cpu()
    q, k, v = qkv.float().view(n, 577, 3, 16, 64).permute(2, 0, 3, 1, 4)
    p = torch.softmax(q @ k.transpose(-1, -2) * float(np.log(2.0)), dim=-1)
    ref = (p @ v).permute(0, 2, 1, 3).reshape(n * 577, 1024)
    assert float((out - ref).norm() / ref.norm()) < tol


# ------------------------------------------------------------------------------------------------ ViT vs the reference
def test_vit2_embedding_matches_reference(env, vit2, golden_dir):
    from pigeon_amd.clip_embedder import CLIPEmbedding
    sd, model = vit2
    g = _gold(golden_dir, "vit2.npz")
    px = env["syn"].make_pixels(4, seed=77)
    emb = CLIPEmbedding("unused", device=DEV, clip_model=model)(px)           # reference call surface
    ref = torch.from_numpy(g["embedding"])
    assert env["orc"].rel_err(emb.cpu(), ref) < EMB_TOL
    assert env["orc"].max_rel_err_rows(emb.cpu(), ref) < EMB_TOL
    hid = model(pixel_values=px).last_hidden_state.cpu()
    rows = torch.from_numpy(g["lhs_rows"])
    assert env["orc"].rel_err(hid[:, [0, 1, 2, 288, 575, 576]], rows) < EMB_TOL   # incl. CLS row 0 and last patch 576


def test_vit24_embedding_matches_reference(env, vit24, golden_dir):
    sd, model = vit24
    g = _gold(golden_dir, "vit24.npz")
    emb = model.embed(env["syn"].make_pixels(4, seed=1234)).cpu()
    ref = torch.from_numpy(g["embedding"])
    assert env["orc"].rel_err(emb, ref) < EMB_TOL
    assert env["orc"].max_rel_err_rows(emb, ref) < EMB_TOL


def test_vit24_stress_weights_match_reference(env, golden_dir):
    """jittered biases / LayerNorm affine + 2x projection scale: a far less benign numeric regime"""
    from pigeon_amd.clip_embedder import HipCLIPVisionModel
    sd = env["syn"].make_vit_weights(seed=5, layers=24, affine_jitter=True, scale=2.0)
    model = HipCLIPVisionModel(sd, layers=24).to(DEV)
    g = _gold(golden_dir, "vit24_stress.npz")
    emb = model.embed(env["syn"].make_pixels(2, seed=99)).cpu()
    assert env["orc"].rel_err(emb, torch.from_numpy(g["embedding"])) < EMB_TOL


def test_vit_bf16_operands_floor(env, vit2, golden_dir):
    """bf16 MFMA operands are supported; their measured error floor (2e-3, weight rounding survives the token mean,
    DESIGN.md) is why fp16 is the default."""
    sd, _ = vit2
    enc = env["ops"].VitEncoder(sd, mma_dtype="bf16")
    emb = enc.forward(env["syn"].make_pixels(4, seed=77).to(DEV)).cpu()
    e = env["orc"].rel_err(emb, torch.from_numpy(_gold(golden_dir, "vit2.npz")["embedding"]))
    assert 2e-4 < e < 4e-3
    enc.close()


def test_vit_batch_invariance_and_chunking(env, vit2):
    """Size-independent properties: an image's embedding is BIT-identical whatever batch it rides in (row results do
    not depend on the tile they land in), across internal chunking, and across repeated runs."""
    sd, model = vit2
    px = env["syn"].make_pixels(11, seed=5).to(DEV)
    full = model.embed(px)
    assert torch.equal(full, model.embed(px))                                     # deterministic
    assert torch.equal(full[3:4], model.embed(px[3:4].contiguous()))              # batch of 1
    assert torch.equal(full[5:9], model.embed(px[5:9].contiguous()))
    enc = env["ops"].VitEncoder(sd, max_chunk=4)                                   # 11 images = chunks 4+4+3
    assert torch.equal(full, enc.forward(px))
    enc.close()
    e16 = model.embed(px.to(torch.bfloat16))
    assert env["orc"].rel_err(e16.cpu(), full.cpu()) < 5e-3


def test_vit_fresh_inputs_vs_oracle(env, vit2):
    sd, model = vit2
    px = env["syn"].make_pixels(3, seed=4242) * 1.7 + 0.3
    ref = env["orc"].clip_embedding(sd, px)
    assert env["orc"].rel_err(model.embed(px).cpu(), ref) < EMB_TOL


# ------------------------------------------------------------------------------------------------ head
def test_head_matches_reference(env, golden_dir, tmp_path):
    from pigeon_amd.super_guessr import SuperGuessr
    g = _gold(golden_dir, "head.npz")
    C, seed, B, eseed, k = [int(x) for x in g["meta"]]
    model = SuperGuessr(None, panorama=True, num_candidates=k, geocell_path=_geocells_csv(tmp_path, C))
    W, b = env["syn"].make_head_weights(C, seed=seed)
    with torch.no_grad():
        model.cell_layer.weight.copy_(W)
        model.cell_layer.bias.copy_(b)
    model.to(DEV).eval()
    gen = torch.Generator().manual_seed(eseed)
    emb = torch.randn((B, 4, 1024), generator=gen) * 0.7 + 0.1
    out = model(embedding=emb, labels=torch.zeros(B, 2, dtype=torch.float64), labels_clf=torch.zeros(B, dtype=torch.long))
    assert np.array_equal(out.preds_geocell.cpu().numpy(), g["preds_geocell"])              # bit-exact argmax
    assert np.array_equal(out.top5_geocells.indices.cpu().numpy(), g["topk_indices"])       # bit-exact top-50
    assert np.array_equal(out.preds_LLH.cpu().numpy(), g["preds_LLH"])                      # float64 gather
    np.testing.assert_allclose(out.top5_geocells.values.cpu().numpy(), g["topk_values"], rtol=1e-5)
    assert abs(float(out.loss_clf) - float(g["loss_clf"])) < 1e-4
    # soft labels by distance (models/super_guessr.py:469-471): pg_haversine_matrix + pg_smooth_labels feed the cross entropy
    soft = SuperGuessr(None, panorama=True, num_candidates=k, should_smooth_labels=True, geocell_path=_geocells_csv(tmp_path, C))
    with torch.no_grad():
        soft.cell_layer.weight.copy_(W)
        soft.cell_layer.bias.copy_(b)
    soft.to(DEV).eval()
    lab = torch.from_numpy(g["smooth_labels_in"])
    o64 = soft(embedding=emb, labels=lab, labels_clf=torch.zeros(B, dtype=torch.long))
    o32 = soft(embedding=emb, labels=lab.float(), labels_clf=torch.zeros(B, dtype=torch.long))
    assert abs(float(o64.loss_clf) / float(g["loss_clf_smooth"]) - 1) < 1e-6
    assert abs(float(o32.loss_clf) / float(g["loss_clf_smooth_f32labels"]) - 1) < 1e-5      # fp32 labels: fp32 deg2rad / cos(lat)
    v = out.top5_geocells.values
    assert bool((v[:, :-1] >= v[:, 1:]).all())                                              # sorted descending
    model.serving = True
    llh, topk, e = model(embedding=emb, labels_clf=torch.zeros(B, dtype=torch.long))         # serving tuple (:462-466)
    assert np.array_equal(llh.cpu().numpy(), g["preds_LLH"]) and e.shape == (B, 4, 1024)


def test_head_ties_pick_lowest_index(env):
    ops = env["ops"]
    C, k = 300, 7
    W = torch.zeros((C, 1024)); b = torch.zeros(C)
    b[[17, 5, 200]] = 1.0                                        # exact three-way tie for the maximum
    cen = torch.arange(2 * C, dtype=torch.float64).view(C, 2)
    o = ops.head_forward(torch.randn((2, 1, 1024)).to(DEV), W.to(DEV), b.to(DEV), cen.to(DEV), k)
    assert o["topk_indices"][0, :3].tolist() == [5, 17, 200] and int(o["preds_geocell"][0]) == 5
    assert o["topk_indices"][0, 3:].tolist() == [0, 1, 2, 3]


@pytest.mark.parametrize("B,P,C,k", [(7, 4, 38399, 50), (1, 1, 17, 17), (130, 4, 1031, 5), (9, 4, 100003, 50)])
def test_head_sizes_outside_the_fixtures(env, B, P, C, k):
    """Geocell counts that are a multiple of nothing, k = C, one row, more rows than a block: the head's outputs are consistent
    with each other (top-k = the k largest of its own softmax in descending order, ties aside; argmax = top-1; prediction =
    that cell's centroid; C = 100 003: the large-head path over a device scratch instead of LDS) and its logits equal an fp64 restatement of mean-over-panels + Linear (super_guessr.py:437-447)."""
    ops = env["ops"]
    g = torch.Generator().manual_seed(C)
    emb = torch.randn((B, P, 1024), generator=g)
    W = torch.randn((C, 1024), generator=g) * 0.05
    b = torch.randn(C, generator=g)
    cen = torch.stack([torch.rand(C, generator=g, dtype=torch.float64) * 360 - 180, torch.rand(C, generator=g, dtype=torch.float64) * 180 - 90], 1)
    o = ops.head_forward(emb.to(DEV), W.to(DEV), b.to(DEV), cen.to(DEV), k)
    logits = o["logits"].cpu()
    want = (emb.double().mean(1) @ W.double().t() + b.double())
    assert float((logits.double() - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))
    probs = torch.softmax(logits, dim=-1)
    tv, ti = o["topk_values"].cpu(), o["topk_indices"].cpu()
    assert torch.equal(o["preds_geocell"].cpu(), ti[:, 0]) and torch.equal(o["preds_LLH"].cpu(), cen[ti[:, 0]])
    assert bool((tv[:, :-1] >= tv[:, 1:]).all()) and all(len(set(r.tolist())) == k for r in ti)
    np.testing.assert_allclose(tv.numpy(), torch.gather(probs, 1, ti).numpy(), rtol=2e-6, atol=1e-12)
    kth = torch.topk(probs, k, dim=-1).values[:, -1]
    assert bool((tv[:, -1] >= kth * (1 - 1e-6)).all())                       # nothing larger was left out
    if C > 38400:        # round 4: beyond the LDS-resident row (38 400 cells) the same kernel runs over a device scratch -- same results
        sub = 38000      # ... as the LDS form gives on a head cut to fit it, for rows whose top-k lie inside the cut
        o2 = ops.head_forward(emb.to(DEV), W[:sub].contiguous().to(DEV), b[:sub].contiguous().to(DEV), cen[:sub].contiguous().to(DEV), k)
        assert torch.equal(o2["logits"].cpu(), logits[:, :sub])
        m, s_, t2 = ops.head_margin(o["logits"], emb.to(DEV), W.to(DEV))
        top2 = torch.topk(logits, 2, dim=-1)
        assert torch.equal(t2.cpu(), top2.indices[:, 1]) and torch.equal(m.cpu(), top2.values[:, 0] - top2.values[:, 1])


# ------------------------------------------------------------------------------------------------ refiner
def _bank(env, g):
    C, ppc, bseed = [int(x) for x in g["meta"][:3]]
    return env["syn"].make_bank(C, ppc, seed=bseed, empty_frac=0.05)


@pytest.mark.parametrize("tag", ["default", "evaluate", "tight"])
def test_refiner_matches_reference(env, golden_dir, tag, capsys):
    from pigeon_amd.proto_refiner import ProtoRefiner
    g = _gold(golden_dir, "refine.npz")
    topk, T, mr = g[f"{tag}_params"]
    ref = ProtoRefiner(topk=int(topk), max_refinement=float(mr), temperature=float(T), bank=_bank(env, g)).eval()
    loss, llh, cell = ref(torch.from_numpy(g["embedding"]).to(DEV), initial_preds=torch.from_numpy(g["initial_preds"]).to(DEV),
                          candidate_cells=torch.from_numpy(g["candidate_cells"]).to(DEV),
                          candidate_probs=torch.from_numpy(g["candidate_probs"]).to(DEV))
    assert loss is None and llh.dtype == torch.float32 and cell.dtype == torch.int64
    assert np.array_equal(cell.cpu().numpy(), g[f"{tag}_cell"])
    assert np.array_equal(llh.cpu().numpy(), g[f"{tag}_LLH"])
    assert "Changed geocell predictions of" in capsys.readouterr().out                      # reference's status print


def test_refiner_3d_embedding_no_probs_and_assert(env, golden_dir):
    from pigeon_amd.proto_refiner import ProtoRefiner
    g = _gold(golden_dir, "refine.npz")
    ref = ProtoRefiner(topk=5, bank=_bank(env, g)).eval()
    emb = torch.from_numpy(g["embedding"])
    emb3 = (emb[:, None, :] + torch.tensor([0.1, -0.1, 0.2, -0.2])[None, :, None]).contiguous()
    _, llh, cell = ref(emb3.to(DEV), initial_preds=torch.from_numpy(g["initial_preds"]).to(DEV),
                       candidate_cells=torch.from_numpy(g["candidate_cells"]).to(DEV), candidate_probs=None, quiet=True)
    assert np.array_equal(cell.cpu().numpy(), g["noprobs3d_cell"]) and np.array_equal(llh.cpu().numpy(), g["noprobs3d_LLH"])
    with pytest.raises(AssertionError):                                                       # proto_refiner.py:135
        ref(emb.to(DEV), initial_preds=torch.zeros(48, 2, dtype=torch.float64), candidate_cells=torch.zeros((48, 3), dtype=torch.long))


def test_empty_and_single_sample_batches(env, vit2, golden_dir, tmp_path):
    """Edge of the batch dimension through the class surface: B = 0 (a rank whose shard is empty must still be able to step) returns
    correctly-shaped empty tensors from the encoder, the head and the refiner without touching the GPU kernels' grids; B = 1 (the
    serving case, one panorama) equals row 0 of the same inputs run as a batch.  (The reference returns empties for B = 0 as well,
    except that its panorama reshape `(B, 4, -1)` is ambiguous for an empty tensor, super_guessr.py:404-405; here the width is explicit.)"""
    from pigeon_amd.proto_refiner import ProtoRefiner
    from pigeon_amd.super_guessr import SuperGuessr
    sd, base = vit2
    C = 120
    model = SuperGuessr(base, panorama=True, freeze_base=True, num_candidates=5, geocell_path=_geocells_csv(tmp_path, C)).to(DEV).eval()
    g = _gold(golden_dir, "refine.npz")
    ref = ProtoRefiner(topk=5, bank=_bank(env, g)).eval()
    with torch.no_grad():
        # B = 0
        e0 = base.embed(torch.zeros((0, 3, 336, 336), device=DEV)) if hasattr(base, "embed") else None
        if e0 is not None:
            assert tuple(e0.shape) == (0, 1024)
        o0 = model(pixel_values=torch.zeros((0, 12, 336, 336), device=DEV), labels=torch.zeros((0, 2), dtype=torch.float64, device=DEV),
                   labels_clf=torch.zeros((0,), dtype=torch.long, device=DEV))
        assert tuple(o0.embedding.shape) == (0, 4, 1024) and tuple(o0.preds_LLH.shape) == (0, 2) and o0.preds_LLH.dtype == torch.float64
        assert tuple(o0.preds_geocell.shape) == (0,) and tuple(o0.top5_geocells.indices.shape) == (0, 5)
        _, l0, c0 = ref(torch.zeros((0, 1024), device=DEV), initial_preds=torch.zeros((0, 2), dtype=torch.float64, device=DEV),
                        candidate_cells=torch.zeros((0, 5), dtype=torch.long, device=DEV),
                        candidate_probs=torch.zeros((0, 5), device=DEV), quiet=True)
        assert tuple(l0.shape) == (0, 2) and l0.dtype == torch.float32 and tuple(c0.shape) == (0,) and c0.dtype == torch.long
        # B = 1 against the same sample inside a batch of 3
        px = env["syn"].make_pixels(12, seed=77).view(3, 12, 336, 336).to(DEV)
        lab = dict(labels=torch.zeros((3, 2), dtype=torch.float64, device=DEV), labels_clf=torch.zeros((3,), dtype=torch.long, device=DEV))
        o3 = model(pixel_values=px, **lab)
        o1 = model(pixel_values=px[:1], labels=lab["labels"][:1], labels_clf=lab["labels_clf"][:1])
        assert torch.equal(o1.embedding, o3.embedding[:1]) and torch.equal(o1.preds_geocell, o3.preds_geocell[:1])
        assert torch.equal(o1.top5_geocells.indices, o3.top5_geocells.indices[:1]) and torch.equal(o1.preds_LLH, o3.preds_LLH[:1])
        emb = torch.from_numpy(g["embedding"]).to(DEV)
        args = lambda sl: dict(initial_preds=torch.from_numpy(g["initial_preds"]).to(DEV)[sl], candidate_cells=torch.from_numpy(g["candidate_cells"]).to(DEV)[sl],
                               candidate_probs=torch.from_numpy(g["candidate_probs"]).to(DEV)[sl], quiet=True)
        _, l1, c1 = ref(emb[:1], **args(slice(0, 1)))
        assert np.array_equal(c1.cpu().numpy(), g["default_cell"][:1]) and np.array_equal(l1.cpu().numpy(), g["default_LLH"][:1])


def test_input_forms_do_not_change_results(env, vit2, golden_dir, tmp_path):
    """What callers really hand over: strided views (top-k slices, a channel-last pixel tensor viewed as NCHW), host tensors, int32
    candidate cells, fp32 initial predictions that are exactly representable, fp16 pixels -- the host side canonicalises
    (device, dtype, contiguity) before a pointer crosses the C ABI, so the results equal those of the canonical tensors; shapes the
    kernels do not implement are refused by name, not read out of bounds."""
    from pigeon_amd.proto_refiner import ProtoRefiner
    sd, base = vit2
    L = env["lib"]
    g = _gold(golden_dir, "refine.npz")
    ref = ProtoRefiner(topk=5, bank=_bank(env, g)).eval()
    emb, init = torch.from_numpy(g["embedding"]), torch.from_numpy(g["initial_preds"])
    cand, probs = torch.from_numpy(g["candidate_cells"]), torch.from_numpy(g["candidate_probs"])
    want_cell, want_llh = g["default_cell"], g["default_LLH"]
    B, k = cand.shape
    wide_c = torch.cat([cand, torch.zeros((B, 3), dtype=cand.dtype)], dim=1)[:, :k]        # a (B,k) slice of a (B,k+3) top-k: strided
    wide_p = torch.cat([probs, torch.zeros((B, 3))], dim=1)[:, :k]
    emb_t = emb.t().contiguous().t()                                                        # column-major embedding
    assert not wide_c.is_contiguous() and not emb_t.is_contiguous()
    for kw in (dict(e=emb_t.to(DEV), i=init.to(DEV), c=wide_c.to(DEV), p=wide_p.to(DEV)),   # strided device tensors
               dict(e=emb, i=init, c=cand, p=probs),                                        # everything on the host (reference: .to('cuda') inside)
               dict(e=emb.to(DEV).double(), i=init.to(DEV), c=cand.to(DEV).int(), p=probs.to(DEV).double())):   # other dtypes
        _, llh, cell = ref(kw["e"], initial_preds=kw["i"], candidate_cells=kw["c"], candidate_probs=kw["p"], quiet=True)
        assert llh.is_cuda and np.array_equal(cell.cpu().numpy(), want_cell) and np.array_equal(llh.cpu().numpy(), want_llh)
    # encoder: NHWC storage viewed as NCHW, a host tensor, fp16 pixels (exactly representable values) -- same embedding
    px = (torch.randint(-8, 9, (3, 3, 336, 336), generator=torch.Generator().manual_seed(5)).float() / 4).to(DEV)
    want = base.embed(px)
    nhwc_view = px.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    assert not nhwc_view.is_contiguous()
    for form in (nhwc_view, px.cpu(), px.half(), px.double()):
        assert torch.equal(base.embed(form), want)
    for bad in (torch.zeros((2, 3, 224, 224), device=DEV), torch.zeros((3, 336, 336), device=DEV), torch.zeros((2, 1, 336, 336), device=DEV)):
        with pytest.raises(L.PigeonHipError):
            base.embed(bad)
    # head: k > C and a wrong embedding width are refused
    ops = env["ops"]
    W, b, cen = torch.zeros((6, 1024), device=DEV), torch.zeros(6, device=DEV), torch.zeros((6, 2), dtype=torch.float64, device=DEV)
    with pytest.raises(L.PigeonHipError):
        ops.head_forward(torch.zeros((2, 1, 1024), device=DEV), W, b, cen, 7)
    with pytest.raises(L.PigeonHipError):
        ops.head_forward(torch.zeros((2, 1, 768), device=DEV), W, b, cen, 3)
    # around the path: a member index past the training bank, CSR offsets that do not end at the list, a (N,3) point list
    tr = torch.zeros((4, 1024), device=DEV)
    with pytest.raises(L.PigeonHipError):
        ops.proto_build(tr, torch.tensor([0, 2], device=DEV), torch.tensor([1, 4], device=DEV))
    with pytest.raises(L.PigeonHipError):
        ops.proto_build(tr, torch.tensor([0, 3], device=DEV), torch.tensor([1, 2], device=DEV))
    with pytest.raises(L.PigeonHipError):
        ops.haversine_matrix(torch.zeros((3, 3), dtype=torch.float64, device=DEV), torch.zeros((2, 2), dtype=torch.float64, device=DEV))


def test_refiner_built_from_reference_files(env, golden_dir, tmp_path):
    """ProtoRefiner(proto_path=CSV, dataset_path=HF dataset dir): the reference's own on-disk inputs."""
    from pigeon_amd.proto_refiner import ProtoRefiner
    g = _gold(golden_dir, "refine.npz")
    bank = _bank(env, g)
    csv, ds = os.path.join(str(tmp_path), "p.csv"), os.path.join(str(tmp_path), "ds")
    env["syn"].write_bank_reference_files(bank, csv, ds)
    ref = ProtoRefiner(topk=5, max_refinement=1000, temperature=1.6, proto_path=csv, dataset_path=ds).eval()
    assert np.array_equal(ref.host_bank.proto_emb, bank.proto_emb)
    _, llh, cell = ref(torch.from_numpy(g["embedding"]).to(DEV), initial_preds=torch.from_numpy(g["initial_preds"]).to(DEV),
                       candidate_cells=torch.from_numpy(g["candidate_cells"]).to(DEV),
                       candidate_probs=torch.from_numpy(g["candidate_probs"]).to(DEV), quiet=True)
    assert np.array_equal(cell.cpu().numpy(), g["default_cell"]) and np.array_equal(llh.cpu().numpy(), g["default_LLH"])


def test_refiner_full_size_bank_vs_oracle(env):
    """BASELINE-size bank (10 000 cells x 100 prototypes = 1M x 1024 fp32): parity on a 24-query sample against the
    oracle, plus the size-independent property that the result does not depend on the other queries in the batch."""
    ops, syn, orc = env["ops"], env["syn"], env["orc"]
    bank = syn.make_bank_device(10000, 100, seed=2, device=DEV)
    dbank = ops.DeviceBank(bank, device=DEV)
    g = torch.Generator().manual_seed(3)
    B = 128
    q = torch.randn((B, 4, 1024), generator=g)
    cand = torch.randint(0, 10000, (B, 5), generator=g)
    cp = torch.softmax(torch.randn((B, 5), generator=g), dim=-1)
    ini = torch.stack([torch.rand(B, generator=g, dtype=torch.float64) * 360 - 180, torch.rand(B, generator=g, dtype=torch.float64) * 180 - 90], 1)
    llh, cell, choice = ops.refine_forward(dbank, q.to(DEV), ini.to(DEV), cand.to(DEV), cp.to(DEV), 5, 1.6, 1000.0)
    llh2, cell2, _ = ops.refine_forward(dbank, q[40:64].contiguous().to(DEV), ini[40:64].contiguous().to(DEV),
                                        cand[40:64].contiguous().to(DEV), cp[40:64].contiguous().to(DEV), 5, 1.6, 1000.0)
    assert torch.equal(llh[40:64], llh2) and torch.equal(cell[40:64], cell2)

    class Lazy:
        def __init__(s, t): s.t = t
        def __getitem__(s, i):
            i = torch.from_numpy(i) if isinstance(i, np.ndarray) else i
            return s.t[i.to(DEV) if torch.is_tensor(i) else i].cpu()

    class HB: pass
    hb = HB()
    hb.proto_emb, hb.train_emb, hb.train_lnglat = Lazy(bank["proto_emb"]), Lazy(bank["train_emb"]), Lazy(bank["train_lnglat"])
    for k in ("cell_off", "proto_count", "member_off", "member_idx", "proto_lnglat"):
        setattr(hb, k, bank[k].cpu().numpy())
    _, o_llh, o_cell = orc.proto_refiner_forward(hb, q[:24], ini[:24], cand[:24], cp[:24], 5, 1.6, 1000.0)
    assert torch.equal(cell[:24].cpu(), o_cell)
    assert np.allclose(llh[:24].cpu().numpy(), o_llh.numpy(), rtol=1e-6, atol=0)


def test_refiner_ragged_bank_vs_oracle(env):
    """Cell sizes the fixtures do not have: empty cells next to a 4 099-prototype cell (more than one block's worth of rows, not a
    multiple of anything), a single-prototype cell, a 300-member cluster (the within-cluster farthest-member search over many rows),
    the same cell twice in one candidate list, zero candidate probabilities, topk < k -- bit-for-bit against the oracle."""
    ops, orc = env["ops"], env["orc"]
    rng = np.random.default_rng(12)
    sizes = np.array([0, 1, 4099, 2, 0, 257, 33], dtype=np.int64)
    cell_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    P = int(cell_off[-1])
    count = np.where(rng.random(P) < 0.5, rng.integers(2, 9, P), 1).astype(np.int32)
    count[1] = 300                                   # a big cluster inside the big cell
    count[0] = 1                                     # the single-prototype cell is a singleton cluster
    member_off = np.concatenate([[0], np.cumsum(count.astype(np.int64))]).astype(np.int64)
    Ntr = int(member_off[-1])
    member_idx = rng.permutation(Ntr).astype(np.int64)
    train_emb = rng.standard_normal((Ntr, 1024)).astype(np.float32)
    train_lnglat = np.stack([rng.uniform(-180, 180, Ntr), rng.uniform(-90, 90, Ntr)], 1).astype(np.float32)
    proto_emb = np.stack([train_emb[member_idx[member_off[p]:member_off[p + 1]]].mean(0) for p in range(P)]).astype(np.float32)
    proto_lnglat = np.stack([rng.uniform(-180, 180, P), rng.uniform(-90, 90, P)], 1).astype(np.float32)

    class HB:
        pass
    hb = HB()
    hb.proto_emb, hb.cell_off, hb.proto_lnglat, hb.proto_count = proto_emb, cell_off, proto_lnglat, count
    hb.member_off, hb.member_idx, hb.train_emb, hb.train_lnglat = member_off, member_idx, train_emb, train_lnglat
    dbank = ops.DeviceBank({k: getattr(hb, k) for k in ("proto_emb", "cell_off", "proto_lnglat", "proto_count", "member_off",
                                                        "member_idx", "train_emb", "train_lnglat")}, device=DEV)
    B, k = 40, 6
    g = torch.Generator().manual_seed(9)
    # queries near prototypes of the big cell / the big cluster so that those really win, plus far-away ones
    q = torch.randn((B, 1024), generator=g)
    q[:8] = torch.from_numpy(proto_emb[1:9]) + 0.01 * torch.randn((8, 1024), generator=g)
    q[8:12] = torch.from_numpy(proto_emb[1]) + 0.2 * torch.randn((4, 1024), generator=g)
    cand = torch.randint(0, len(sizes), (B, k), generator=g)
    cand[:12, 0] = 2                                 # the big cell first
    cand[12:16] = torch.tensor([2, 2, 5, 5, 1, 0])   # duplicates and an empty cell
    cand[16:18] = torch.tensor([0, 4, 0, 4, 0, 4])   # nothing but empty cells
    cp = torch.softmax(torch.randn((B, k), generator=g), dim=-1)
    cp[18:22, 1:4] = 0.0                             # zero probabilities
    ini = torch.stack([torch.rand(B, generator=g, dtype=torch.float64) * 360 - 180, torch.rand(B, generator=g, dtype=torch.float64) * 180 - 90], 1)
    for topk, T, mr in ((6, 1.6, 1000.0), (4, 0.6, 100000.0), (1, 1.0, 50.0)):
        llh, cell, choice = ops.refine_forward(dbank, q.to(DEV), ini.to(DEV), cand.to(DEV), cp.to(DEV), topk, T, mr)
        _, o_llh, o_cell = orc.proto_refiner_forward(hb, q, ini, cand, cp, topk, T, mr)
        assert torch.equal(cell.cpu(), o_cell), (topk, T, mr)
        assert np.array_equal(llh.cpu().numpy(), o_llh.numpy()), (topk, T, mr)
    # the big cluster was really searched: a query at the big cluster's prototype ends on one of its 300 members (or was vetoed)
    members = set(map(tuple, train_lnglat[member_idx[member_off[1]:member_off[2]]].tolist()))
    llh, cell, choice = ops.refine_forward(dbank, q.to(DEV), ini.to(DEV), cand.to(DEV), cp.to(DEV), 6, 1.6, 1e9)
    assert sum(tuple(r) in members for r in llh[8:12].cpu().numpy().tolist()) >= 1


# ------------------------------------------------------------------------------------------------ end to end
def test_pipeline_matches_reference(env, vit2, golden_dir, tmp_path):
    """pixels -> SuperGuessr(ViT + head) -> ProtoRefiner, against the real reference's outputs (pipeline.npz)."""
    from pigeon_amd.super_guessr import SuperGuessr
    from pigeon_amd.proto_refiner import ProtoRefiner
    sd, vit = vit2
    g = _gold(golden_dir, "pipeline.npz")
    wseed, layers, jitter, n, pseed, C, ppc, bseed, hseed = [int(x) for x in g["meta"]]
    model = SuperGuessr(vit, panorama=True, hierarchical=False, multi_task=False, heading=False, freeze_base=True,
                        num_candidates=50, geocell_path=_geocells_csv(tmp_path, C))
    W, b = env["syn"].make_head_weights(C, seed=hseed)
    with torch.no_grad():
        model.cell_layer.weight.copy_(W * 8)
        model.cell_layer.bias.copy_(b)
    model.to(DEV).eval()
    px = env["syn"].make_pixels(n, seed=pseed, panorama=True)
    out = model(pixel_values=px, labels=torch.zeros(2, 2, dtype=torch.float64), labels_clf=torch.zeros(2, dtype=torch.long))
    assert env["orc"].rel_err(out.embedding.cpu(), torch.from_numpy(g["embedding"])) < EMB_TOL
    assert np.array_equal(out.preds_geocell.cpu().numpy(), g["preds_geocell"])
    assert np.array_equal(out.preds_LLH.cpu().numpy(), g["preds_LLH"])
    # top-k ORDER deep in the tail may swap where the reference's own margins are below the fp16 error; the first 5
    # (the ones refinement consumes) must match
    assert np.array_equal(out.top5_geocells.indices.cpu().numpy()[:, :5], g["topk_indices"][:, :5])
    refiner = ProtoRefiner(topk=5, max_refinement=1000, temperature=1.6,
                           bank=env["syn"].make_bank(C, ppc, seed=bseed, empty_frac=0.05)).eval()
    _, llh, cell = refiner(out.embedding, initial_preds=out.preds_LLH, candidate_cells=out.top5_geocells.indices,
                           candidate_probs=out.top5_geocells.values, quiet=True)
    assert np.array_equal(cell.cpu().numpy(), g["refined_cell"])
    assert np.array_equal(llh.cpu().numpy(), g["refined_LLH"])


def test_full_size_step_properties(env, vit24, tmp_path):
    """BASELINE configs[2] size (128 panoramas = 512 images, C = 10 000): size-independent checks."""
    from pigeon_amd.super_guessr import SuperGuessr
    sd, vit = vit24
    C = 10000
    model = SuperGuessr(vit, panorama=True, freeze_base=True, num_candidates=5, geocell_path=_geocells_csv(tmp_path, C))
    W, b = env["syn"].make_head_weights(C, seed=0)
    with torch.no_grad():
        model.cell_layer.weight.copy_(W); model.cell_layer.bias.copy_(b)
    model.to(DEV).eval()
    g = torch.Generator(device=DEV).manual_seed(1)
    px = torch.randn((128, 12, 336, 336), generator=g, device=DEV)
    out = model(pixel_values=px, labels_clf=None)
    assert out.embedding.shape == (128, 4, 1024) and bool(torch.isfinite(out.embedding).all())
    re_all = set(model.last_reencoded.tolist())
    sub = model(pixel_values=px[17:19].contiguous(), labels_clf=None)               # same panoramas, tiny batch
    if re_all & {17, 18}:      # re-encoded rows: the exact tier's K-part count follows the batch size (fp32 summation order, ~1e-7)
        assert torch.allclose(sub.embedding, out.embedding[17:19], rtol=1e-5, atol=1e-6)
    else:
        assert torch.equal(sub.embedding, out.embedding[17:19])
    assert torch.equal(sub.preds_geocell, out.preds_geocell[17:19])
    assert torch.equal(out.top5_geocells.indices[:, 0], out.preds_geocell)
    ref = env["orc"].super_guessr_forward(W, b, model.lla_geocells.data.cpu(), 5, embedding=out.embedding.cpu())
    assert torch.equal(out.preds_geocell.cpu(), ref["preds_geocell"])               # head argmax bit-exact on 128 rows
    assert torch.equal(out.top5_geocells.indices.cpu(), ref["topk"].indices)


# ------------------------------------------------------------------------------------------------ LayerNorm fold
def test_ln_fold_building_blocks(env):
    """The pieces of the LayerNorm-folded GEMM chain, each against plain torch on the same rounded operands:
    rowstat_cast, the residual GEMM that also emits the 16-bit row copy + partial statistics, rowstat_finalize, and the
    GEMM whose epilogue applies rstd * acc - mean*rstd * colsum + c."""
    ops, L = env["ops"], env["lib"]
    g = torch.Generator().manual_seed(31)
    M, D, K2, N2 = 700 + 13, 1024, 256, 512
    x = (torch.randn((M, D), generator=g) * 1.7 + 0.4)
    x[:, 5] += 30.0                                                 # an outlier channel
    x16, rs = ops.rowstat_cast(x.to(DEV))
    mu, var = x.mean(-1), x.var(-1, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    assert torch.equal(x16.cpu(), x.to(torch.float16))
    assert torch.allclose(rs[:, 0].cpu(), rstd, rtol=2e-6) and torch.allclose(rs[:, 1].cpu(), mu * rstd, rtol=2e-5, atol=1e-6)
    # residual GEMM with statistics: X (M,1024) += A (M,K2) W^T
    A = torch.randn((M, K2), generator=g).to(torch.float16)
    W = (torch.randn((D, K2), generator=g) * 0.05).to(torch.float16)
    b = torch.randn(D, generator=g)
    X = x.clone().to(DEV)
    xn16, part = ops.gemm16_resid_stat(A.to(DEV), W.to(DEV), b.to(DEV), X)
    Xref = x.clone().to(DEV)
    ops.gemm16(A.to(DEV), W.to(DEV), b.to(DEV), Xref, L.EPI_RESID, variant=36)
    assert torch.equal(X, Xref)                                     # same bits as the plain residual epilogue of the same kernel
    Xref8 = x.clone().to(DEV)
    ops.gemm16(A.to(DEV), W.to(DEV), b.to(DEV), Xref8, L.EPI_RESID, variant=8)   # 32x32x16 MFMAs: another association inside an instruction
    assert torch.allclose(X, Xref8, rtol=2e-3, atol=1e-5 * float(Xref8.abs().max()))
    assert torch.equal(xn16.cpu(), X.cpu().to(torch.float16))
    Xc = X.cpu()
    ps = Xc.view(M, D // 64, 64)
    assert torch.allclose(part[:, :, 0].cpu().T, ps.sum(-1), rtol=1e-5, atol=1e-4)         # slot-major (N/64, M, 2)
    assert torch.allclose(part[:, :, 1].cpu().T, (ps * ps).sum(-1), rtol=1e-5, atol=1e-4)
    rs2 = ops.rowstat_finalize(part)
    mu2, var2 = Xc.mean(-1), Xc.var(-1, unbiased=False)
    rstd2 = 1.0 / torch.sqrt(var2 + 1e-5)
    assert torch.allclose(rs2[:, 0].cpu(), rstd2, rtol=2e-5) and torch.allclose(rs2[:, 1].cpu(), mu2 * rstd2, rtol=1e-4, atol=1e-5)
    # LN-applied-in-the-epilogue GEMM against LayerNorm followed by a plain matmul on the same folded operands
    gamma, beta = 1.0 + 0.1 * torch.randn(D, generator=g), 0.05 * torch.randn(D, generator=g)
    Wl = torch.randn((N2, D), generator=g) * 0.03
    bl = torch.randn(N2, generator=g) * 0.1
    Wf = (Wl * gamma[None, :]).to(torch.float16)
    colsum = Wf.float().sum(-1)
    cb = (Wl.double() @ beta.double()).float() + bl
    for epi in (L.EPI_QKV_LN, L.EPI_GELU_LN):
        out = ops.gemm16_ln(xn16, Wf.to(DEV), cb.to(DEV), colsum.to(DEV), rs2, epi, qscale=0.25, qcols=256).float().cpu()
        acc = xn16.cpu().float() @ Wf.float().T
        y = rs2[:, :1].cpu() * acc - rs2[:, 1:].cpu() * colsum[None, :] + cb[None, :]
        if epi == L.EPI_QKV_LN:
            y[:, :256] *= 0.25
        else:
            y = y * torch.sigmoid(1.702 * y)
        assert (out - y).abs().max() <= 2.0 ** -10 * y.abs().max() + 1e-3, epi
        # and against the unfused definition LN(x) W^T + b (fp32): only the 16-bit rounding points differ
        yref = torch.nn.functional.layer_norm(Xc, (D,), gamma, beta, 1e-5) @ Wl.T + bl
        if epi == L.EPI_QKV_LN:
            yref[:, :256] *= 0.25
        else:
            yref = yref * torch.sigmoid(1.702 * yref)
        assert env["orc"].rel_err(out, yref) < 2e-3, epi


def test_ln_fold_encoder_matches_reference(env, golden_dir, monkeypatch):
    """The LayerNorm-folded encoder (the default) and the separate-LayerNorm chain (PIGEON_LN_FOLD=0) against the same golden
    vectors and tolerance, and against each other."""
    from pigeon_amd.clip_embedder import HipCLIPVisionModel
    sd = env["syn"].make_vit_weights(seed=11, layers=2, affine_jitter=True)
    px = env["syn"].make_pixels(4, seed=77).to(DEV)
    ref = torch.from_numpy(_gold(golden_dir, "vit2.npz")["embedding"])
    monkeypatch.setenv("PIGEON_LN_FOLD", "1")
    fold = HipCLIPVisionModel(sd, layers=2).to(DEV).embed(px).cpu()
    monkeypatch.setenv("PIGEON_LN_FOLD", "0")
    base = HipCLIPVisionModel(sd, layers=2).to(DEV).embed(px).cpu()
    assert env["orc"].rel_err(fold, ref) < EMB_TOL
    assert env["orc"].rel_err(base, ref) < EMB_TOL
    assert env["orc"].rel_err(fold, base) < 5e-4
    assert not torch.equal(fold, base)                    # two different chains really ran


# ------------------------------------------------------------------------------------------------ serving path
def test_serve_predict_panorama(env, vit2, tmp_path):
    """pigeon_amd.serve.predict_panorama (the handler behind POST /api/v1/predict): four PIL views -> GPU preprocessing ->
    SuperGuessr(serving=True) tuple -> ProtoRefiner; equals the explicit chain and, up to the refinement, the oracle head."""
    from PIL import Image
    from pigeon_amd import serve
    from pigeon_amd.clip_embedder import gpu_preprocess
    from pigeon_amd.proto_refiner import ProtoRefiner
    from pigeon_amd.super_guessr import SuperGuessr
    sd, vit = vit2
    C = 200
    model = SuperGuessr(vit, panorama=True, serving=True, freeze_base=True, num_candidates=5, geocell_path=_geocells_csv(tmp_path, C))
    W, b = env["syn"].make_head_weights(C, seed=3)
    with torch.no_grad():
        model.cell_layer.weight.copy_(W); model.cell_layer.bias.copy_(b)
    model.to(DEV).eval()
    rng = np.random.default_rng(5)
    views = [Image.fromarray(rng.integers(0, 255, (400 + 16 * i, 640, 3), dtype=np.uint8)) for i in range(4)]
    got = serve.predict_panorama(views, model)
    px = gpu_preprocess(views).reshape(1, 12, 336, 336)
    llh, topk, emb = model(pixel_values=px)
    assert {k: got[k] for k in ("lat", "lng")} == {"lat": float(llh[0, 1]), "lng": float(llh[0, 0])}
    # round 4: the certainty of the geocell top-1 rides along (extra keys; the extension reads lat / lng only)
    assert isinstance(got["geocell_certain"], bool) and got["geocell_margin"] >= 0 and isinstance(got["reencoded_exact"], bool)
    ref = env["orc"].super_guessr_forward(W, b, model.lla_geocells.data.cpu(), 5, embedding=emb.cpu())
    assert torch.equal(topk.indices.cpu(), ref["topk"].indices)
    refiner = ProtoRefiner(topk=5, bank=env["syn"].make_bank(C, 20, seed=4, empty_frac=0.05)).eval()
    got_r = serve.predict_panorama(views, model, refiner)
    # the explicit chain: the model call with the refiner's decisions included in the certainty pass, then the refinement
    from pigeon_amd.evaluate import certain_forward
    (llh, topk, emb), _ = certain_forward(model, refiner, pixel_values=px)
    _, want, _ = refiner(embedding=emb, initial_preds=llh, candidate_cells=topk.indices, candidate_probs=topk.values)
    assert {k: got_r[k] for k in ("lat", "lng")} == {"lat": float(want[0, 1]), "lng": float(want[0, 0])}


def test_token_mean_summation_order_is_pinned(env):
    """pg_op_token_mean (reference models/clip_embedder.py:64-65 `last_hidden_state.mean(dim=1)`): four partial sums, rows dealt
    round-robin, ((s0 + s1) + (s2 + s3)) / 577 in fp32 -- restated in numpy float32 and compared BIT FOR BIT, so that a faster kernel
    (round 6: 16 loads in flight instead of 4) cannot change an embedding's bits; and within fp32 rounding of torch's own mean."""
    ops = env["ops"]
    g = torch.Generator().manual_seed(4)
    x = (torch.randn((3, 577, 1024), generator=g) * 3 + 0.5).contiguous()
    got = ops.token_mean(x.to(DEV)).cpu().numpy()
    xn = x.numpy()
    s = [np.zeros((3, 1024), dtype=np.float32) for _ in range(4)]
    for t in range(577):
        j = t % 4 if t < 576 else 0
        s[j] = (s[j] + xn[:, t, :]).astype(np.float32)
    want = (((s[0] + s[1]).astype(np.float32) + (s[2] + s[3]).astype(np.float32)).astype(np.float32) / np.float32(577)).astype(np.float32)
    assert np.array_equal(got, want)
    assert np.allclose(got, xn.mean(axis=1, dtype=np.float64), rtol=0, atol=2e-6)
