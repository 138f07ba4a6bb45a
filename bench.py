"""Benchmark of the PIGEON inference hot path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One STEP = one pass of the hot path over one batch of synthetic input already resident in HBM:
  BASELINE.json configs[3]: 128 panoramas (4 x 3x336x336 = 512 images) per GPU -> ViT-L/14-336 (24 layers, random
  init seed 0) -> token mean -> SuperGuessr geocell head (C = 10 000) -> [N>1: one RCCL all-gather of embeddings /
  candidates through the C ABI] -> ProtoRefiner top-5 over a 1M x 1024 fp32 prototype bank (10 000 cells x 100).
Weak scaling: every rank processes its own 128 panoramas; value = total images / max-over-ranks time.
A few distinct pixel batches are resident and used in turn, and the head is centred on the mean embedding (calibrated
during warm-up), so that the synthetic panoramas spread over the geocells the way real ones do and the refinement really
streams different prototype rows from HBM every step (otherwise every query asks for the same cells and the 256 MB
Infinity Cache serves them).

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline            the dominant kernel (the fc1 GEMM), algorithmic FLOPs per launch / mean launch time measured live with
                      HIP events on the launch stream during the timed region;
  roofline_refine     the refinement kernels against the HBM roofline: algorithmic bytes (4096 B per bank row streamed,
                      SURVEY 8d formula, counted by the kernel itself) / mean time between stream events;
  h2d_inclusive       the same step fed from pinned host memory (PCIe copy inside the timed region);
  other_configs       BASELINE configs[1] (encoder only, 256 single images) and configs[2] (SuperGuessr, no refinement);
  secondary_baseline  (rank 0, N=1) stock PyTorch-ROCm on the same box in the same run: the HuggingFace CLIPVisionModel the
                      reference calls (fp32 and fp16 autocast) and torch.matmul / SDPA on the five hot shapes (hipBLASLt);
  cpu_baseline        the oracle (oracle/pigeon_oracle.py = CPU restatement of the reference path) timed on this box's
                      host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_IMAGE = 381.918e9           # patch 0.694 + 24 x 15.884 GFLOP (2*MAC, un-padded T=577)
PEAK_MFMA = 2.5e15                   # dense 16-bit MFMA peak, MI355X_MICROARCH.md (fp16 == bf16 rate)
PEAK_HBM = 8.0e12                    # HBM3E peak, MI355X_MICROARCH.md
GEMM_FLOPS = {                       # algorithmic FLOPs per token row of each GEMM class
    "gemm_qkv": 2 * 1024 * 3072, "gemm_out": 2 * 1024 * 1024, "gemm_fc1": 2 * 1024 * 4096, "gemm_fc2": 2 * 4096 * 1024,
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--panoramas", type=int, default=128, help="panoramas per GPU per step")
    ap.add_argument("--cells", type=int, default=10000)
    ap.add_argument("--protos-per-cell", type=int, default=100)
    ap.add_argument("--topk", type=int, default=5)
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--pixel-batches", type=int, default=4, help="distinct resident pixel batches used in turn")
    ap.add_argument("--cpu-images", type=int, default=16, help="images in the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-refine", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip h2d / other configs / secondary baseline (profiling runs)")
    return ap.parse_args()


class _LazyRows:
    """Row accessor over a device matrix that materialises only the rows asked for (cpu_baseline leg)."""

    def __init__(self, t):
        self.t = t

    def __getitem__(self, idx):
        if isinstance(idx, np.ndarray):
            idx = torch.from_numpy(idx)
        if torch.is_tensor(idx):
            idx = idx.to(self.t.device)
        return self.t[idx].cpu()


def cpu_baseline(args, vit_sd, model, bank_t, pixels_dev):
    """Oracle = CPU restatement of the reference path, on a bounded sample of the SAME workload."""
    from oracle import pigeon_oracle as orc
    n_img = args.cpu_images
    npano = max(1, n_img // 4)
    cores = os.cpu_count() or 1
    px = pixels_dev[:npano].cpu()
    # thread count: timed, not assumed -- one panorama (4 images) through the oracle ViT at each candidate count
    sweep = {}
    for nt in sorted({min(cores, 32), min(cores, 64), cores}):
        torch.set_num_threads(nt)
        t0 = time.time()
        orc.clip_embedding(vit_sd, px[0].reshape(4, 3, 336, 336))
        sweep[nt] = 4 / (time.time() - t0)
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    W = model.cell_layer.weight.data.cpu()
    b = model.cell_layer.bias.data.cpu()
    cen = model.lla_geocells.data.cpu()

    class B:
        pass
    hb = B()
    t0 = time.time()
    o = orc.super_guessr_forward(W, b, cen, args.topk, vit_sd=vit_sd, pixel_values=px)
    t_vit = time.time() - t0
    if not args.no_refine:
        hb.proto_emb = _LazyRows(bank_t["proto_emb"])
        hb.train_emb = _LazyRows(bank_t["train_emb"])
        for k in ("cell_off", "proto_count", "member_off", "member_idx"):
            setattr(hb, k, bank_t[k].cpu().numpy())
        hb.proto_lnglat = bank_t["proto_lnglat"].cpu().numpy()
        hb.train_lnglat = _LazyRows(bank_t["train_lnglat"])
        orc.proto_refiner_forward(hb, o["embedding"], o["preds_LLH"], o["topk"].indices, o["topk"].values, args.topk, 1.6, 1000)
    dt = time.time() - t0
    return {"value": npano * 4 / dt, "unit": "images/s", "cores": best, "box_cores": cores, "kind": "port",
            "thread_sweep_images_per_s": {str(k): round(v, 3) for k, v in sweep.items()},
            "sample": f"{npano} panoramas ({npano * 4} images) through oracle ViT-L/14 fp32 + head + top-{args.topk} refine "
                      f"(torch CPU, {best} threads = the fastest of the sweep, {dt:.1f} s, ViT+head {t_vit:.1f} s); linear in images",
            "cpu_model": _cpu_model()}, o


def _committed_traffic(kernel, rows):
    """Memory-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC pass (profiles/rNN/traffic.json:
    2 x FETCH_SIZE -- the gfx950 correction of MI355X_MICROARCH.md -- + WRITE_SIZE, separate --pmc passes over this very
    command, tools/prof_bench.sh).  Counters cannot be read from inside the process; (None, None) if there is no pass
    for this launch size."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
            if kernel in d and "hbm_bytes_per_launch_corrected" in d[kernel] and d[kernel].get("rows") == rows:
                return d[kernel]["hbm_bytes_per_launch_corrected"], {
                    "algorithmic_bytes": d[kernel].get("algorithmic_bytes"), "source": os.path.relpath(f, ROOT),
                    "note": "counts L2-miss traffic on the fabric side (Infinity Cache hits included)"}
        except (OSError, ValueError):
            pass
    return None, None


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _time_gpu(fn, iters, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def secondary_baseline(dev, vit_sd, layers):
    """Stock PyTorch-ROCm on the same GPU, same run -- NOT the product path, never imported by pigeon_amd:
    (a) transformers.CLIPVisionModel (the module the reference calls at models/clip_embedder.py:63 /
        models/super_guessr.py:395) + token mean, fp32 and fp16 autocast, 128 images per step;
    (b) torch.matmul (hipBLASLt) on the four GEMM shapes of a 512-image chunk and SDPA on its attention shape, fp16."""
    out = {}
    M = 512 * 577
    shapes = {"gemm_qkv": (M, 1024, 3072), "gemm_out": (M, 1024, 1024), "gemm_fc1": (M, 1024, 4096), "gemm_fc2": (M, 4096, 1024)}
    lib = {}
    try:
        for name, (m, k, n) in shapes.items():
            a = torch.randn((m, k), device=dev, dtype=torch.float16)
            w = torch.randn((n, k), device=dev, dtype=torch.float16)
            bias = torch.randn((n,), device=dev, dtype=torch.float16)
            t = _time_gpu(lambda: torch.nn.functional.linear(a, w, bias), 10, 3)
            lib[name] = {"ms": t * 1e3, "tflops": 2.0 * m * k * n / t / 1e12}
            del a, w
        q = torch.randn((512, 16, 577, 64), device=dev, dtype=torch.float16)
        kk, v = torch.randn_like(q), torch.randn_like(q)
        t = _time_gpu(lambda: torch.nn.functional.scaled_dot_product_attention(q, kk, v), 10, 3)
        lib["attention_sdpa"] = {"ms": t * 1e3, "tflops": 4.0 * 512 * 16 * 577 * 577 * 64 / t / 1e12}
        del q, kk, v
        out["torch_fp16_ops_512_images"] = lib
    except Exception as e:  # noqa
        out["torch_fp16_ops_512_images"] = {"error": repr(e)}
    try:
        from transformers import CLIPVisionConfig, CLIPVisionModel
        cfg = CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=layers, num_attention_heads=16,
                               image_size=336, patch_size=14, projection_dim=768)
        with contextlib.redirect_stdout(io.StringIO()):
            hf = CLIPVisionModel(cfg)
        hf.load_state_dict(vit_sd, strict=True)
        hf = hf.to(dev).eval()
        px = torch.randn((128, 3, 336, 336), device=dev)

        def fwd():
            with torch.no_grad():
                return hf(pixel_values=px).last_hidden_state.mean(dim=1)
        t32 = _time_gpu(fwd, 2, 1)

        def fwd16():
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
                return hf(pixel_values=px).last_hidden_state.mean(dim=1)
        t16 = _time_gpu(fwd16, 3, 1)
        out["hf_clip_vision_model"] = {
            "what": "transformers.CLIPVisionModel(ViT-L/14-336 config, same weights).to('cuda') + mean over tokens, 128 images per step",
            "fp32_images_per_s": 128 / t32, "fp16_autocast_images_per_s": 128 / t16,
            "fp16_autocast_mfma_frac": 128 / t16 * FLOP_PER_IMAGE / PEAK_MFMA,
            "transformers": __import__("transformers").__version__, "torch": torch.__version__}
        del hf, px
    except Exception as e:  # noqa
        out["hf_clip_vision_model"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    from pigeon_amd import _lib, distributed, hip_ops, synthetic
    from pigeon_amd.clip_embedder import HipCLIPVisionModel
    from pigeon_amd.evaluate import PanoramaPipeline
    from pigeon_amd.proto_refiner import ProtoRefiner
    from pigeon_amd.super_guessr import SuperGuessr

    comm = distributed.init_from_env()               # control plane (gloo); the data-path collective is RCCL through the C ABI
    rank, world = comm.rank, comm.world_size
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    _lib.require_gpu()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")

    # ---- model, head, bank (identical replicas on every rank) ----
    vit_sd = synthetic.make_vit_weights(seed=0, layers=args.layers)
    base = HipCLIPVisionModel(vit_sd, layers=args.layers)
    tmp = tempfile.mkdtemp(prefix="pigeon_bench_")
    geo_csv = os.path.join(tmp, f"geocells_{rank}.csv")
    synthetic.write_geocell_csv(geo_csv, synthetic.make_geocells(args.cells, seed=0))
    with contextlib.redirect_stdout(io.StringIO()):
        model = SuperGuessr(base, panorama=True, freeze_base=True, num_candidates=args.topk, geocell_path=geo_csv)
    W, b = synthetic.make_head_weights(args.cells, seed=0)
    with torch.no_grad():
        model.cell_layer.weight.copy_(W)
        model.cell_layer.bias.copy_(b)
    model.to(dev).eval()
    refiner, bank_t = None, None
    if not args.no_refine:
        bank_t = synthetic.make_bank_device(args.cells, args.protos_per_cell, seed=2, device=str(dev))
        refiner = ProtoRefiner(topk=args.topk, max_refinement=1000, temperature=1.6, bank=bank_t, device=str(dev)).eval()
    pipe = PanoramaPipeline(model, refiner, comm)

    nb = max(1, args.pixel_batches)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    pixel_batches = [torch.randn((args.panoramas, 12, 336, 336), generator=g, device=dev) for _ in range(nb)]   # resident in HBM
    index = torch.arange(args.panoramas, device=dev) + rank * args.panoramas

    # ---- warm-up (also packs weights, sizes the workspace) + head calibration ----
    out = pipe.step(pixel_batches[0], index)
    with torch.no_grad():
        # centre the synthetic head on the mean embedding and spread its logits (sigma = 4): panoramas then fall into many
        # different geocells with top-1 probabilities 0.05 .. 0.9 (tests/golden/pipeline24 uses the same construction);
        # identical on every rank (rank 0's statistics are broadcast with the control-plane group)
        pe = out["embedding"][: args.panoramas].mean(dim=1)
        stats = [pe.mean(dim=0).cpu()]
        if world > 1:
            torch.distributed.broadcast_object_list(stats, src=0)
        center = stats[0].to(dev)
        sig = float(((pe - center) @ model.cell_layer.weight.data.t()).std()) if rank == 0 else 0.0
        sc = [float(2.0 ** np.round(np.log2(4.0 / max(sig, 1e-12))))]
        if world > 1:
            torch.distributed.broadcast_object_list(sc, src=0)
        model.cell_layer.weight.mul_(sc[0])
        model.cell_layer.bias.copy_(b.to(dev) - model.cell_layer.weight.data @ center)
    for i in range(max(args.warmup, 1)):
        out = pipe.step(pixel_batches[i % nb], index)
    torch.cuda.synchronize()
    enc = base._encoder(dev)
    enc.profile_reset()
    enc.profile_enable(True)
    pipe.refine_events = [] if refiner is not None else None
    refine_rows = []

    # ---- timed region: exactly K steps between barrier + synchronize on both sides ----
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = pipe.step(pixel_batches[i % nb], index)
        if refiner is not None:
            refine_rows.append(refiner.last_scratch)             # device tensor kept; summed after the timed region
    torch.cuda.synchronize()
    comm.barrier()
    dt = time.perf_counter() - t0
    enc.profile_enable(False)
    prof = enc.profile_read()
    dt = comm.max_over_ranks(dt)
    distinct_cells = int(torch.unique(out["preds_geocell"]).numel())

    if rank != 0:
        return
    images_per_step = args.panoramas * 4 * world
    value = images_per_step * args.steps / dt
    kernels = {}
    for name, (cnt, ms) in prof.items():
        if cnt:
            kernels[name] = {"launches": cnt, "avg_ms": ms / cnt}
    # per-launch rows: the encoder processes <= max_chunk (512) images per internal pass, so a 512-image step launches
    # every layer kernel once with M = 512*577 rows
    chunk_rows = min(args.panoramas * 4, enc.max_chunk) * 577
    for name in GEMM_FLOPS:
        if name in kernels:
            kernels[name]["tflops"] = GEMM_FLOPS[name] * chunk_rows / (kernels[name]["avg_ms"] * 1e-3) / 1e12
    if "attention" in kernels:
        kernels["attention"]["tflops"] = 4.0 * 577 * 577 * 64 * 16 * (chunk_rows / 577) / (kernels["attention"]["avg_ms"] * 1e-3) / 1e12
    dom = max((k for k in kernels if k in GEMM_FLOPS), key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches"])
    achieved = kernels[dom]["tflops"]
    traffic, traffic_detail = _committed_traffic(dom, chunk_rows)
    result = {
        "metric": "images/sec end-to-end (ViT+head+refine), 4x336x336",
        "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": enc.mma_dtype, "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: SuperGuessr 4-panorama ViT-L/14-336 + 10k-geocell head + ProtoRefiner "
                               "top-5 over 1Mx1024 bank" + (" (configs[4] shape: sharded over GPUs, all-gather before refinement)" if world > 1 else ""),
                   "panoramas_per_gpu": args.panoramas, "images_per_step": images_per_step, "layers": args.layers,
                   "geocells": args.cells, "prototypes": args.cells * args.protos_per_cell, "topk": args.topk,
                   "parallelism": f"dp{world}", "weights": "random init seed 0 (HF CLIP init distributions); head centred on the mean embedding",
                   "resident_pixel_batches": nb, "distinct_argmax_cells_last_step": distinct_cells},
        "mfma_frac_end_to_end": value * FLOP_PER_IMAGE / (world * PEAK_MFMA),
        "roofline": {"bound": "mfma", "kernel": f"gemm16 {dom}: M={chunk_rows} rows x {GEMM_FLOPS[dom]} FLOP/row per launch",
                     "achieved": achieved, "peak": PEAK_MFMA / 1e12, "unit": "TFLOP/s", "frac": achieved / (PEAK_MFMA / 1e12),
                     "traffic": traffic, "traffic_detail": traffic_detail},
        "kernels": kernels,
    }
    if world > 1:
        result["rccl"] = {"nranks": comm.rccl_ranks(), "version": _lib.load().pg_comm_rccl_version(),
                          "collective": "pg_allgather_many (C ABI, csrc/comm.hip): 5 buffers, one grouped launch per step"}
    if refiner is not None and pipe.refine_events:
        ms = [a.elapsed_time(b_) for a, b_ in pipe.refine_events]
        rows = [float(s[..., 3].sum()) for s in refine_rows]
        bytes_per_launch = 4096.0 * float(np.mean(rows))
        t = float(np.mean(ms)) * 1e-3
        result["roofline_refine"] = {
            "bound": "hbm", "kernel": "refine_candidates_kernel + refine_select_kernel (one refinement of the rank's 128 queries)",
            "achieved": bytes_per_launch / t / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": bytes_per_launch / t / PEAK_HBM,
            "algorithmic_bytes_per_launch": bytes_per_launch, "avg_ms": t * 1e3,
            "traffic": _committed_traffic("refine_candidates", chunk_rows)[0],
            "note": "bytes = 4096 B x bank rows streamed (prototypes of the top-k cells + members of the chosen clusters), counted by the "
                    f"kernel; {nb} pixel batches in turn -> different cells every step, {distinct_cells} distinct argmax cells in the last step"}
    pipe.refine_events = None
    enc.profile_reset()

    if not args.no_extras and world == 1:
        # ---- the same step fed from pinned host memory (PCIe inside the timed region) ----
        try:
            host = torch.empty((args.panoramas, 12, 336, 336), dtype=torch.float32).pin_memory()
            host.copy_(pixel_batches[0])
            stage = torch.empty_like(pixel_batches[0])

            def h2d_step():
                stage.copy_(host, non_blocking=True)
                pipe.step(stage, index)
            t = _time_gpu(h2d_step, max(2, min(args.steps, 3)), 1)
            result["h2d_inclusive"] = {"value": args.panoramas * 4 / t, "unit": "images/s", "ms_per_step": t * 1e3,
                                       "what": "pinned host fp32 pixels (694 MB per 128 panoramas) copied H2D on the compute stream, then the step; no overlap",
                                       "n_gpus": 1}
            del host, stage
        except Exception as e:  # noqa
            result["h2d_inclusive"] = {"error": repr(e)}
        if True:
            # ---- BASELINE configs[1] / configs[2] ----
            oc = []
            try:
                # BASELINE configs[0] is the reference's own CPU plumbing case (run.py embed on 64 images); its shape on this path:
                from pigeon_amd.clip_embedder import CLIPEmbedding
                with contextlib.redirect_stdout(io.StringIO()):
                    embedder = CLIPEmbedding("random", device=str(dev), clip_model=base)
                px64 = pixel_batches[0].reshape(-1, 3, 336, 336)[:64].contiguous()
                t = _time_gpu(lambda: embedder(px64), 3, 1)
                oc.append({"workload": "BASELINE configs[0] shape: CLIPEmbedding.forward on 64 single 336x336 images (the reference runs it on the CPU; "
                                       "cpu_baseline is that leg)", "value": 64 / t, "unit": "images/s", "ms_per_step": t * 1e3})
                single = pixel_batches[0].reshape(-1, 3, 336, 336)[:256].contiguous()
                t = _time_gpu(lambda: base.embed(single), 3, 1)
                oc.append({"workload": "BASELINE configs[1]: ViT-L/14-336 encoder only (+ token mean), batch 256 single-panel 336x336",
                           "value": 256 / t, "unit": "images/s", "ms_per_step": t * 1e3, "mfma_frac": 256 / t * FLOP_PER_IMAGE / PEAK_MFMA})
                t = _time_gpu(lambda: model(pixel_values=pixel_batches[0], labels_clf=None), 3, 1)
                oc.append({"workload": "BASELINE configs[2]: SuperGuessr, 128 panoramas (512 images) ViT-L/14-336 + 10k-geocell head, no refinement",
                           "value": 512 * (args.panoramas / 128) / t, "unit": "images/s", "ms_per_step": t * 1e3,
                           "mfma_frac": 4 * args.panoramas / t * FLOP_PER_IMAGE / PEAK_MFMA})
            except Exception as e:  # noqa
                oc.append({"error": repr(e)})
            result["other_configs"] = oc

    if world == 1 and args.cpu_images > 0:
        try:
            cb, o = cpu_baseline(args, vit_sd, model, bank_t, pixel_batches[(args.steps - 1) % nb])
            result["cpu_baseline"] = cb
            # while we have the oracle's answer for the first panoramas: report parity of this very run
            npano = o["embedding"].shape[0]
            from oracle import pigeon_oracle as orc
            result["parity_vs_oracle_sample"] = {
                "embedding_rel_err": orc.rel_err(out["embedding"][:npano].cpu(), o["embedding"]),
                "geocell_argmax_equal": bool(torch.equal(out["preds_geocell"][:npano].cpu(), o["preds_geocell"]))}
        except Exception as e:  # noqa
            result["cpu_baseline"] = {"error": str(e)}
    if world == 1 and not args.no_extras:
        del pixel_batches
        torch.cuda.empty_cache()
        result["secondary_baseline"] = secondary_baseline(dev, vit_sd, args.layers)
    print(json.dumps(result))


if __name__ == "__main__":
    main()
