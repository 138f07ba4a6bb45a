"""GPU parity of the CLIP preprocessing kernels (pigeon_amd/csrc/preprocess.hip, through the C ABI) against the
numpy oracle, which is itself pinned to Pillow (tests/test_preprocess_cpu.py).  Integer work: bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def env():
    from pigeon_amd import _lib, hip_ops
    from oracle import clip_preprocess_oracle as orc
    _lib.require_gpu()
    return dict(ops=hip_ops, orc=orc)


@pytest.mark.parametrize("h,w", [(640, 640), (480, 640), (640, 480), (200, 300), (336, 336), (337, 336), (336, 500),
                                 (1000, 350)])
def test_prep_matches_oracle_bit_exact(env, h, w):
    ops, orc = env["ops"], env["orc"]
    rng = np.random.default_rng(h * 7 + w)
    imgs = rng.integers(0, 256, (3, h, w, 3), dtype=np.uint8)          # white noise: every rounding / clipping path
    imgs[1] = np.kron(rng.integers(0, 256, (h // 16 + 1, w // 16 + 1, 3), dtype=np.uint8),
                      np.ones((16, 16, 1), dtype=np.uint8))[:h, :w]      # blocky image: overshoot at edges
    prep = ops.Preprocessor(h, w)
    nh, nw = orc.resize_output_size(h, w)
    assert (prep.resized_h, prep.resized_w) == (nh, nw)
    got = prep(torch.from_numpy(imgs).to(DEV), torch.float32).cpu().numpy()
    got16 = prep(torch.from_numpy(imgs).to(DEV), torch.float16).cpu()
    for i in range(3):
        ref = orc.clip_preprocess(imgs[i])
        assert np.array_equal(got[i], ref), f"image {i}: {np.abs(got[i] - ref).max()}"
        assert torch.equal(got16[i], torch.from_numpy(ref).to(torch.float16))


def test_prep_matches_pillow_golden(env, golden_dir):
    ops = env["ops"]
    gold = np.load(os.path.join(golden_dir, "preprocess.npz"))
    lut = env["orc"].normalise_lut()
    for tag in ["square", "landscape", "portrait", "upscale", "crop_only"]:
        img, u8 = gold[f"{tag}_img"], gold[f"{tag}_u8"]
        got = ops.Preprocessor(img.shape[0], img.shape[1])(torch.from_numpy(img[None]).to(DEV)).cpu().numpy()[0]
        ref = np.stack([lut[c][u8[:, :, c]] for c in range(3)])
        assert np.array_equal(got, ref), tag


def test_clip_embedding_accepts_raw_images(env):
    """CLIPEmbedding.forward(PIL images / uint8 arrays) == forward(host-preprocessed tensor): the GPU preprocessing
    path feeds the encoder the same operand bits (fp16 pixels = what im2col makes of the fp32 values)."""
    Image = pytest.importorskip("PIL.Image")
    from pigeon_amd import synthetic
    from pigeon_amd.clip_embedder import CLIPEmbedding, HipCLIPVisionModel, clip_preprocess
    sd = synthetic.make_vit_weights(seed=11, layers=2, affine_jitter=True)
    emb = CLIPEmbedding("synthetic", device="cuda", clip_model=HipCLIPVisionModel(sd, layers=2))
    rng = np.random.default_rng(5)
    arrs = [rng.integers(0, 256, s, dtype=np.uint8) for s in [(480, 640, 3), (640, 480, 3), (480, 640, 3)]]
    pil = [Image.fromarray(a) for a in arrs]
    e_raw = emb(pil).cpu()
    e_host = emb(clip_preprocess(pil)).cpu()
    assert e_raw.shape == (3, 1024)
    assert torch.equal(e_raw, e_host)
    e_u8 = emb(torch.from_numpy(np.stack([arrs[0], arrs[2]]))).cpu()
    assert torch.equal(e_u8, e_raw[[0, 2]])


# ------------------------------------------------------------------------------------------------ SURVEY 8f rows 3-4
def test_proto_build_matches_torch_mean(env):
    """pg_proto_build == torch's `embeddings.mean(dim=1).mean(dim=0)` (reference proto_refiner.py:370-378), bit for bit,
    with and without the 4-panel axis; empty prototypes give zeros."""
    ops = env["ops"]
    g = torch.Generator().manual_seed(4)
    ntr = 500
    for panels in (1, 4):
        train = torch.randn((ntr, panels, 1024) if panels == 4 else (ntr, 1024), generator=g)
        lens = [1, 3, 0, 8, 2, 17, 1, 64, 300, 16, 33]   # >= 16 members: torch's cascade order; 300 reaches level 2
        idx = [torch.randint(0, ntr, (n,), generator=g) for n in lens]
        off = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int64)
        got = ops.proto_build(train.to(DEV), off.to(DEV), torch.cat(idx).to(DEV)).cpu()
        for p, ix in enumerate(idx):
            if len(ix) == 0:
                assert bool((got[p] == 0).all())
                continue
            e = train[ix]
            if e.dim() == 3:
                e = e.mean(dim=1)
            assert torch.equal(got[p], e.mean(dim=0)), (panels, p)


def test_build_bank_gpu_equals_host(env, tmp_path):
    from pigeon_amd import proto_refiner as pr
    rng = np.random.default_rng(3)
    train = rng.standard_normal((300, 1024)).astype(np.float32)
    lens = rng.integers(0, 9, 120)
    lens[::17] = rng.integers(16, 70, lens[::17].shape[0])       # some prototypes in torch's cascade regime
    off = np.zeros(121, dtype=np.int64); np.cumsum(lens, out=off[1:])
    idx = rng.integers(0, 300, int(off[-1])).astype(np.int64)
    host = pr._segmented_mean_host(train, off, idx, lens.astype(np.int64))
    gpu = pr._segmented_mean_gpu(train, off, idx)
    assert np.array_equal(host, gpu)


@pytest.mark.parametrize("xdt", [torch.float64, torch.float32])
def test_haversine_matrix_and_smooth_labels(env, xdt):
    """Transcendental path: compared with the reference formula evaluated by torch on the CPU.  Tolerance: 1e-12
    relative when everything is fp64; with fp32 labels torch evaluates deg2rad / cos(lat) of x in fp32, and the
    device's cosf and the host's differ by an fp32 ulp; one ulp of cos(lat) moves near-antipodal distances by up to
    0.2 km (1e-5 relative, measured on the CPU by perturbing the host value), so 0.5 km / 3e-5 there (device vs host
    libm -- nothing on this path is bit-exact by contract)."""
    from pigeon_amd import geo_utils
    g = torch.Generator().manual_seed(8)
    N, M = 37, 10000
    x = torch.stack([torch.rand(N, generator=g, dtype=torch.float64) * 360 - 180,
                     torch.rand(N, generator=g, dtype=torch.float64) * 180 - 90], dim=1).to(xdt)
    y = torch.stack([torch.rand(M, generator=g, dtype=torch.float64) * 360 - 180,
                     torch.rand(M, generator=g, dtype=torch.float64) * 180 - 90], dim=1)
    y[5] = x[3].double()                                                         # a zero distance
    ref = geo_utils.haversine_matrix(x, y.t())                                   # host torch expression
    got = geo_utils.haversine_matrix(x.to(DEV), y.to(DEV).t())                   # pg_haversine_matrix
    assert got.dtype == torch.float64 and got.shape == (N, M)
    rtol = 1e-12 if xdt == torch.float64 else 3e-5
    assert torch.allclose(got.cpu(), ref, rtol=rtol, atol=1e-9 if xdt == torch.float64 else 0.5)
    ref_s = geo_utils.smooth_labels(got.cpu(), 65)                               # same distances in: isolates the kernel
    got_s = geo_utils.smooth_labels(got, 65)
    assert torch.allclose(got_s.cpu(), ref_s, rtol=1e-11, atol=1e-300)
    assert bool((got_s.cpu().max(dim=1).values == 1.0).all())                    # the nearest cell always gets 1
    d = got.clone(); d[2, 7] = float("nan"); d[4, 9] = float("inf")
    s = geo_utils.smooth_labels(d, 65).cpu()
    assert bool((s[2] == 0).all()) and s[4, 9] == 0 and torch.isfinite(s).all()  # torch.min propagates NaN -> row of 0
