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


@pytest.mark.parametrize("h,w", [(1, 1), (2, 900), (900, 2), (3000, 4000), (50, 51), (17, 4000), (1, 336)])
def test_prep_extreme_geometries(env, h, w):
    """One pixel, aspect ratios of 450:1 (the resized image would be 151 200 pixels long -- only the centre crop is ever computed),
    an 8.9x downscale (36-tap bicubic support), a 6.7x upscale: bit-exact against the oracle, which equals Pillow + transformers
    4.23.1 on exactly these geometries (checked on the CPU when this test was written)."""
    ops, orc = env["ops"], env["orc"]
    img = np.random.default_rng(h * 7 + w).integers(0, 256, (1, h, w, 3), dtype=np.uint8)
    prep = ops.Preprocessor(h, w)
    assert (prep.resized_h, prep.resized_w) == orc.resize_output_size(h, w)
    got = prep(torch.from_numpy(img).to(DEV), torch.float32).cpu().numpy()[0]
    ref = orc.clip_preprocess(img[0])
    assert np.array_equal(got, ref), f"{np.abs(got - ref).max()}"


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


def test_gpu_preprocess_single_image_modes_and_handle_cache(env, monkeypatch):
    """One PIL image (not a list), grayscale / RGBA / palette images (converted to RGB as the reference's processor does), and
    more input geometries than the handle cache keeps (least recently used handles are destroyed): bit-identical to the host path."""
    Image = pytest.importorskip("PIL.Image")
    from pigeon_amd import clip_embedder as ce
    rng = np.random.default_rng(6)
    rgb = Image.fromarray(rng.integers(0, 256, (300, 400, 3), dtype=np.uint8))
    one = ce.gpu_preprocess(rgb).cpu()
    assert one.shape == (1, 3, 336, 336) and torch.equal(one, ce.clip_preprocess(rgb))
    modes = [rgb.convert("L"), rgb.convert("RGBA"), rgb.convert("P"), Image.fromarray(rng.integers(0, 256, (350, 350), dtype=np.uint8))]
    assert torch.equal(ce.gpu_preprocess(modes).cpu(), ce.clip_preprocess(modes))
    monkeypatch.setattr(ce, "_PREPROCESSORS_MAX", 2)
    sizes = [(340, 500), (500, 340), (336, 336), (400, 400), (340, 500), (700, 350)]
    ims = [Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)) for h, w in sizes]
    for _ in range(2):                                             # second pass: evicted geometries are rebuilt
        for im in ims:
            assert torch.equal(ce.gpu_preprocess([im]).cpu(), ce.clip_preprocess([im]))
    assert len(ce._PREPROCESSORS) <= 2
    assert torch.equal(ce.gpu_preprocess(ims).cpu(), ce.clip_preprocess(ims))       # one call, five geometries, cap 2
    assert ce.gpu_preprocess([]).shape == (0, 3, 336, 336)


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
