"""HTTP contract of the serving shim (pigeon_amd/serve.py) against what the reference's browser extension sends and
reads (bot/chrome_extension/scripts/duel.js:51-70, :86-97).  CPU only: the model is a stub, the preprocessing is the
host restatement of the CLIP processor."""
import base64
import io
from collections import namedtuple

import numpy as np
import pytest
import torch
from PIL import Image
from starlette.testclient import TestClient

from pigeon_amd import serve
from pigeon_amd.clip_embedder import clip_preprocess

TopK = namedtuple("TopK", ["values", "indices"])


def _data_uri(seed, size=(80, 60), fmt="PNG"):
    rng = np.random.default_rng(seed)
    img = Image.fromarray(rng.integers(0, 255, (size[1], size[0], 3), dtype=np.uint8))
    buf = io.BytesIO()
    img.save(buf, format=fmt)
    mime = "png" if fmt == "PNG" else "jpeg"
    return f"data:image/{mime};base64," + base64.b64encode(buf.getvalue()).decode()


class StubModel:
    """Serving-tuple contract of SuperGuessr(serving=True): (pred_LLH [lng, lat] f64, TopK, embedding)."""

    def __init__(self):
        self.seen = None

    def __call__(self, pixel_values=None):
        self.seen = pixel_values
        mean = float(pixel_values.mean())
        llh = torch.tensor([[12.5 + mean, -33.25]], dtype=torch.float64)            # [lng, lat]
        return llh, TopK(torch.ones(1, 5), torch.arange(5)[None]), torch.zeros(1, 4, 1024)


class StubRefiner:
    def __call__(self, embedding=None, initial_preds=None, candidate_cells=None, candidate_probs=None):
        assert embedding.shape == (1, 4, 1024) and candidate_cells.shape == (1, 5)
        return None, initial_preds.to(torch.float32) + torch.tensor([[1.0, 2.0]]), torch.zeros(1, dtype=torch.int64)


@pytest.fixture()
def client():
    model, log = StubModel(), []
    app = serve.make_app(model, refiner=None, preprocess=clip_preprocess, game_log=log)
    return TestClient(app), model, log


def test_predict_four_views(client):
    c, model, _ = client
    body = {"gameID": "abc", "roundID": 3, "image": _data_uri(1), "image_2": _data_uri(2, fmt="JPEG"),
            "image_3": _data_uri(3), "image_4": _data_uri(4)}
    r = c.post("/api/v1/predict", json=body)
    assert r.status_code == 200
    out = r.json()
    assert out["gameID"] == "abc" and out["roundID"] == 3
    assert set(out["results"]) == {"lat", "lng"}                 # duel.js:68 reads guess.results.lat / .lng
    assert out["results"]["lat"] == -33.25                       # the model's tuple is [lng, lat]
    assert model.seen.shape == (1, 12, 336, 336)                 # one panorama, four panels along the channel axis
    # panel order = heading order: panel 1 is the preprocessing of image_2
    want = clip_preprocess([serve.decode_data_uri(body["image_2"])])[0]
    assert torch.equal(model.seen[0, 3:6], want)
    assert abs(out["results"]["lng"] - (12.5 + float(model.seen.mean()))) < 1e-9


def test_predict_single_view_and_refiner():
    model = StubModel()
    c = TestClient(serve.make_app(model, refiner=StubRefiner(), preprocess=clip_preprocess))
    r = c.post("/api/v1/predict", json={"image": _data_uri(7)})   # classic.js: one screenshot
    assert r.status_code == 200
    assert torch.equal(model.seen[0, 0:3], model.seen[0, 9:12])  # the view repeated four times
    assert r.json()["results"]["lat"] == pytest.approx(-31.25)   # refiner output is what is returned


def test_bad_requests(client):
    c, _, _ = client
    assert c.post("/api/v1/predict", json={"gameID": 1}).status_code == 400
    assert c.post("/api/v1/predict", json={"image": "data:image/png;base64,AAAA"}).status_code == 400
    assert c.post("/api/v1/predict", json={"image": _data_uri(1), "image_2": _data_uri(2)}).status_code == 400
    assert c.post("/api/v1/predict", content=b"not json", headers={"Content-Type": "application/json"}).status_code == 400
    assert c.get("/api/v1/predict").status_code == 405


def test_game_log(client):
    c, _, log = client
    r = c.post("/api/v1/game", json={"gameID": "abc", "roundID": 2, "game": {"score": 4999}})
    assert r.status_code == 200 and log == [{"gameID": "abc", "roundID": 2, "game": {"score": 4999}}]


def test_decode_variants():
    uri = _data_uri(11)
    a = serve.decode_data_uri(uri)
    b = serve.decode_data_uri(uri.split(",", 1)[1])               # bare base64 without the data: prefix
    assert a.mode == "RGB" and a.size == (80, 60) and list(a.getdata()) == list(b.getdata())
    rgba = Image.new("RGBA", (10, 8), (10, 20, 30, 128))
    buf = io.BytesIO(); rgba.save(buf, format="PNG")
    c = serve.decode_data_uri("data:image/png;base64," + base64.b64encode(buf.getvalue()).decode())
    assert c.mode == "RGB" and c.getpixel((0, 0)) == (10, 20, 30)  # alpha dropped, as PIL's convert('RGB') does
    with pytest.raises(serve.BadRequest):
        serve.decode_data_uri("")
    with pytest.raises(serve.BadRequest):
        serve.decode_data_uri(None)
