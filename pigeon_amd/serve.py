"""The serving endpoint the reference's GeoGuessr bot talks to.

The reference ships only the CLIENT: `bot/chrome_extension/scripts/duel.js:51-64` posts four street-view screenshots
(headings 0 / 90 / 180 / 270 degrees, `chrome.tabs.captureVisibleTab` data URIs) as JSON to
`POST http://127.0.0.1:5000/api/v1/predict` and reads `guess.results.lat` / `guess.results.lng` from the answer
(:66-70); after the round it posts the game state to `POST /api/v1/game` (:86-97) and ignores the body.  The backend
behind that port is not in the repository ("an external server running PIGEON on an NVIDIA A100", bot/README.md).  This
module is that backend over the HIP hot path: decode -> CLIP preprocessing (`pg_prep_forward` on the GPU) ->
`SuperGuessr(serving=True)` (the tuple path, reference models/super_guessr.py:462-466) -> optional `ProtoRefiner`.

    app = make_app(model, refiner)              # model: SuperGuessr(..., panorama=True, serving=True).eval()
    uvicorn.run(app, host="127.0.0.1", port=5000)

`predict_panorama` is the whole request handler without HTTP, so the GPU tests can call it directly.  The model and the
preprocessor are injected: the module itself imports neither torch.cuda nor the HIP library, and the HTTP contract is
tested on CPU with a stub model (tests/test_serve_cpu.py).
"""
from __future__ import annotations

import base64
import io
import json
from typing import Callable, Dict, List, Optional, Sequence

IMAGE_KEYS = ("image", "image_2", "image_3", "image_4")          # duel.js:59-62, in heading order


class BadRequest(ValueError):
    pass


def decode_data_uri(uri: str):
    """'data:image/png;base64,....' (or bare base64) -> RGB PIL image."""
    from PIL import Image
    if not isinstance(uri, str) or not uri:
        raise BadRequest("image field must be a non-empty data URI string")
    payload = uri.split(",", 1)[1] if uri.startswith("data:") else uri
    try:
        raw = base64.b64decode(payload, validate=False)
        img = Image.open(io.BytesIO(raw))
        img.load()
    except Exception as e:                                        # PIL raises many types for corrupt input
        raise BadRequest(f"cannot decode image: {e}") from e
    return img.convert("RGB")


def panorama_from_request(body: Dict) -> List:
    """The four views of one location.  A single-image request (`classic.js` sends only `image`) is answered from that
    view repeated four times: the panorama head averages the four panel embeddings (super_guessr.py:437)."""
    if not isinstance(body, dict) or IMAGE_KEYS[0] not in body:
        raise BadRequest("JSON body with an 'image' field expected")
    views = [decode_data_uri(body[k]) for k in IMAGE_KEYS if body.get(k)]
    if len(views) == 1:
        views = views * 4
    if len(views) != 4:
        raise BadRequest(f"1 or 4 images expected, got {len(views)}")
    return views


def predict_panorama(views: Sequence, model, refiner=None, preprocess: Optional[Callable] = None) -> Dict:
    """views: four PIL images -> {'lat', 'lng'} (+ 'geocell_margin', 'geocell_certain', 'reencoded_exact' when the model exposes the
    certainty of its top-1).  `preprocess(list of PIL) -> (4,3,336,336)` defaults to the GPU path."""
    import torch
    if preprocess is None:
        from .clip_embedder import gpu_preprocess
        preprocess = gpu_preprocess
    px = preprocess(list(views))                                   # (4,3,336,336), CLIP-normalised
    px = px.reshape(1, 12, px.shape[-2], px.shape[-1])             # one panorama, panels along the channel axis (:386-393)
    with torch.no_grad():
        if hasattr(model, "encode_head"):
            # pigeon_amd.SuperGuessr: the same call with the certainty of every discrete output (top-1 cell AND what the refiner below
            # will pick) checked, and the panorama re-encoded in the exact mode if it is not certain (exact_top1, the default)
            from .evaluate import certain_forward
            (pred_llh, topk, embedding), info = certain_forward(model, refiner, pixel_values=px)
            if refiner is not None:
                pred_llh = info["refined_LLH"]                     # the refinement certain_forward ran (once; re-run for re-encoded rows)
        else:
            pred_llh, topk, embedding = model(pixel_values=px)     # serving tuple, [lng, lat] order (:455, :462-466)
            if refiner is not None:
                _, pred_llh, _ = refiner(embedding=embedding, initial_preds=pred_llh, candidate_cells=topk.indices,
                                         candidate_probs=topk.values)
    lng, lat = (float(v) for v in pred_llh.reshape(-1, 2)[0].tolist())
    res = {"lat": lat, "lng": lng}
    # certainty of the geocell top-1 (pigeon_amd.SuperGuessr, round 4): extra keys, the extension reads lat / lng only
    if getattr(model, "last_certain", None) is not None:
        res["geocell_margin"] = float(model.last_margin.reshape(-1)[0])
        res["geocell_certain"] = bool(model.last_certain.reshape(-1)[0])
        res["reencoded_exact"] = bool(getattr(model, "last_reencoded", None) is not None and model.last_reencoded.numel() > 0)
    return res


def make_app(model, refiner=None, preprocess: Optional[Callable] = None, game_log: Optional[List] = None):
    """Starlette application with the two routes the extension calls."""
    from starlette.applications import Starlette
    from starlette.responses import JSONResponse
    from starlette.routing import Route

    async def predict(request):
        try:
            body = await request.json()
            views = panorama_from_request(body)
        except (BadRequest, json.JSONDecodeError) as e:
            return JSONResponse({"error": str(e)}, status_code=400)
        # called inline on the event loop, on purpose: the handles behind `model` are one per (device, stream) and not
        # thread-safe, so requests are answered one at a time -- a `def` handler would run in Starlette's pool
        res = predict_panorama(views, model, refiner, preprocess)
        return JSONResponse({"gameID": body.get("gameID"), "roundID": body.get("roundID"), "results": res})

    async def game(request):                                       # duel.js:86-97: fire-and-forget round log
        try:
            body = await request.json()
        except json.JSONDecodeError as e:
            return JSONResponse({"error": str(e)}, status_code=400)
        if game_log is not None:
            game_log.append(body)
        return JSONResponse({"status": "ok"})

    return Starlette(routes=[Route("/api/v1/predict", predict, methods=["POST"]),
                             Route("/api/v1/game", game, methods=["POST"])])


def main(argv=None):
    """python -m pigeon_amd.serve --head <head.model> [--geocells <csv>] [--protos <csv> --dataset <dir>] [--port 5000]"""
    import argparse
    import torch
    import uvicorn
    from .clip_embedder import HipCLIPVisionModel
    from .proto_refiner import ProtoRefiner
    from .super_guessr import SuperGuessr
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--head", default=None, help="SuperGuessr checkpoint (torch.save state dict); random init if omitted")
    ap.add_argument("--geocells", default=None)
    ap.add_argument("--protos", default=None)
    ap.add_argument("--dataset", default=None)
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--no-exact-top1", dest="exact_top1", action="store_false",
                    help="switch the exact mode off (default on: panoramas whose discrete outputs are inside the 16-bit path's error band "
                         "are re-encoded in the encoder's exact mode, so that the answer is the reference's fp32 one)")
    ap.add_argument("--host", default="127.0.0.1")
    ap.add_argument("--port", type=int, default=5000)
    args = ap.parse_args(argv)
    kw = {"geocell_path": args.geocells} if args.geocells else {}
    from .clip_embedder import load_pretrained_clip
    try:
        base = load_pretrained_clip()                                  # CLIP_MODEL from local files (env PIGEON_CLIP_MODEL / HF cache)
    except RuntimeError as why:
        print(f"[serve] no pretrained tower ({why}); using a seeded random-init ViT-L/14-336 with {args.layers} layers")
        base = HipCLIPVisionModel(seed=0, layers=args.layers)
    model = SuperGuessr(base, panorama=True, serving=True, freeze_base=True, exact_top1=args.exact_top1, **kw).to("cuda").eval()
    if args.head:
        model.load_state(args.head)
    refiner = ProtoRefiner(proto_path=args.protos, dataset_path=args.dataset) if args.protos else None
    torch.cuda.synchronize()
    uvicorn.run(make_app(model, refiner), host=args.host, port=args.port)


if __name__ == "__main__":
    main()
