"""Round 6: latency of one serving request -- what the bot's `POST /api/v1/predict` handler does between decoding the four views and
answering (pigeon_amd/serve.py predict_panorama; reference models/super_guessr.py:462-466 behind bot/README.md's one-A100 server):
GPU preprocessing of four 640 x 640 views, encoder, head, certainty, refinement, exact re-encode if the panorama is not certain, the
(lng, lat) back on the host.  24-layer tower, 10 000 geocells, 1M-row bank; 60 requests after 5 warm-ups; PIGEON_GEMM_MID=0 for the
round-5 routing.   python tools/serve_latency.py"""
import contextlib, io, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from pigeon_amd import synthetic
from pigeon_amd.clip_embedder import HipCLIPVisionModel
from pigeon_amd.proto_refiner import ProtoRefiner
from pigeon_amd.serve import predict_panorama
from pigeon_amd.super_guessr import SuperGuessr

dev, C = "cuda", 10000
base = HipCLIPVisionModel(synthetic.make_vit_weights(seed=0, layers=24), layers=24)
geo = os.path.join(tempfile.mkdtemp(prefix="pigeon_serve_"), "g.csv")
synthetic.write_geocell_csv(geo, synthetic.make_geocells(C, seed=0))
with contextlib.redirect_stdout(io.StringIO()):
    model = SuperGuessr(base, panorama=True, serving=True, freeze_base=True, num_candidates=5, geocell_path=geo)
W, b = synthetic.make_head_weights(C, seed=0)
with torch.no_grad():
    model.cell_layer.weight.copy_(W * 256); model.cell_layer.bias.copy_(b)
model.to(dev).eval()
refiner = ProtoRefiner(topk=5, max_refinement=1000, temperature=1.6, bank=synthetic.make_bank_device(C, 100, seed=2, device=dev), device=dev).eval()
rng = np.random.default_rng(0)
reqs = [[Image.fromarray(rng.integers(0, 256, (640, 640, 3), dtype=np.uint8)) for _ in range(4)] for _ in range(65)]
model.calibrate_certainty(torch.randn((32, 12, 336, 336), device=dev))
lat, exact = [], 0
for i, views in enumerate(reqs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = predict_panorama(views, model, refiner)
    torch.cuda.synchronize()
    if i >= 5:
        lat.append((time.perf_counter() - t0) * 1e3)
        exact += int(res.get("reencoded_exact", False))
lat = np.sort(np.asarray(lat))
print(f"PIGEON_GEMM_MID={os.environ.get('PIGEON_GEMM_MID', '1')}: {len(lat)} requests (4 views of 640x640 each, PIL -> answer): median {np.median(lat):.2f} ms, "
      f"p10 {lat[len(lat) // 10]:.2f}, p90 {lat[9 * len(lat) // 10]:.2f}, max {lat[-1]:.2f} ms; {exact} went through the exact tier")
if "--profile" in sys.argv:                                   # where the host side of a request goes (cProfile, 40 more requests)
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for views in reqs[5:45]:
        predict_panorama(views, model, refiner)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats("cumulative").print_stats(45)
    # and the device side: events around the whole request against the host clock
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dts = []
    for views in reqs[5:25]:
        torch.cuda.synchronize()
        ev0.record()
        predict_panorama(views, model, refiner)
        ev1.record()
        torch.cuda.synchronize()
        dts.append(ev0.elapsed_time(ev1))
    print(f"stream time first launch -> last launch of a request: median {np.median(dts):.2f} ms")
