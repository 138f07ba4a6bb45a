"""CPU tests of the CLIP preprocessing restatement (oracle/clip_preprocess_oracle.py): pinned to golden vectors made
by the real Pillow (tests/golden/preprocess.npz, oracle/make_preprocess_golden.py), to the live Pillow when it is
importable, and the host-side product function pigeon_amd.clip_embedder.clip_preprocess against both."""
import os

import numpy as np
import pytest

from oracle import clip_preprocess_oracle as orc

TAGS = ["square", "landscape", "portrait", "upscale", "crop_only"]


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "preprocess.npz"))


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_matches_pillow_golden(gold, tag):
    got = orc.clip_preprocess_u8(gold[f"{tag}_img"])
    assert got.shape == (336, 336, 3) and got.dtype == np.uint8
    assert np.array_equal(got, gold[f"{tag}_u8"])                 # integer resampling: bit-exact


def test_oracle_matches_live_pillow():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(7)
    for (h, w) in [(480, 640), (641, 479), (100, 100), (350, 1000)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)      # white noise: worst case for rounding / clipping
        nh, nw = orc.resize_output_size(h, w)
        ref = np.asarray(Image.fromarray(img).resize((nw, nh), resample=Image.BICUBIC))
        assert np.array_equal(orc.pil_resize_bicubic(img, nh, nw), ref), (h, w)


def test_resize_geometry_follows_transformers_4_23():
    assert orc.resize_output_size(480, 640) == (336, 448)
    assert orc.resize_output_size(640, 480) == (448, 336)
    assert orc.resize_output_size(500, 333) == (504, 336)          # int(336 * 500 / 333) = 504 (floor, not round)
    assert orc.resize_output_size(336, 999) == (336, 999)


def test_float_part_and_lut(gold):
    u8 = gold["landscape_u8"]
    pv = orc.clip_preprocess(gold["landscape_img"])
    mean = np.array(orc.OPENAI_CLIP_MEAN).astype(np.float32)
    std = np.array(orc.OPENAI_CLIP_STD).astype(np.float32)
    ref = ((u8.astype(np.float32) / 255.0 - mean) / std).transpose(2, 0, 1)
    assert pv.dtype == np.float32 and np.array_equal(pv, ref)
    lut = orc.normalise_lut()
    assert np.array_equal(np.stack([lut[c][u8[:, :, c]] for c in range(3)]), ref)


def test_host_product_preprocess_matches_oracle(gold):
    Image = pytest.importorskip("PIL.Image")
    from pigeon_amd.clip_embedder import clip_preprocess
    for tag in TAGS:
        img = gold[f"{tag}_img"]
        got = clip_preprocess(Image.fromarray(img)).numpy()
        assert np.array_equal(got[0], orc.clip_preprocess(img)), tag
