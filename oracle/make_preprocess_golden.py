"""Generate tests/golden/preprocess.npz with the REAL Pillow (run in the authoring container; the fixture travels).

For a few seeded uint8 images of different geometries: the image, and what
`PIL.Image.fromarray(img).resize(shorter edge -> 336, BICUBIC)` + centre crop gives (uint8, 336x336x3) --
the integer part of CLIPProcessor the HIP kernel and the numpy oracle must reproduce bit for bit.
    python oracle/make_preprocess_golden.py
"""
import os
import sys

import numpy as np
import PIL
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import clip_preprocess_oracle as orc  # noqa: E402


def photo_like(rng, h, w):
    # blocky random tiles + a smooth ramp + sparse salt pixels: edges, gradients and overshoot (clipping) are all
    # exercised while the file stays compressible (white noise is covered by the live-Pillow test)
    base = rng.random((h // 8 + 2, w // 8 + 2, 3))
    im = np.kron(base, np.ones((8, 8, 1)))[:h, :w] * 200
    yy, xx = np.mgrid[0:h, 0:w]
    im += ((yy // 4 + xx // 6) % 23)[:, :, None] * 2.0
    salt = rng.random((h, w)) < 0.002
    im[salt] = 255
    return np.clip(im, 0, 255).astype(np.uint8)


def main():
    rng = np.random.default_rng(2024)
    out = {"pillow_version": np.array(PIL.__version__)}
    for tag, (h, w) in {"square": (448, 448), "landscape": (300, 420), "portrait": (400, 340), "upscale": (90, 120),
                        "crop_only": (336, 380)}.items():
        img = photo_like(rng, h, w)
        nh, nw = orc.resize_output_size(h, w)
        pim = Image.fromarray(img)
        if (nh, nw) != (h, w):
            pim = pim.resize((nw, nh), resample=Image.BICUBIC)
        top, left = (nh - 336) // 2, (nw - 336) // 2
        ref = np.asarray(pim.crop((left, top, left + 336, top + 336)))
        out[f"{tag}_img"] = img
        out[f"{tag}_u8"] = ref
    path = os.path.join(ROOT, "tests", "golden", "preprocess.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
