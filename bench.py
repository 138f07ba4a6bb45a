"""Benchmark of the PIGEON inference hot path on MI355X.

    python bench.py                                   # 1 GPU
    python bench.py --gpus 8 --steps K --warmup W     # spawns its 8 ranks itself (one process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W     # or under torchrun (RANK / LOCAL_RANK / WORLD_SIZE from the env)
    python bench.py --gpus 2 --dry-run                # CPU: the same control flow with stub kernels over gloo (contract test)

The reference's embed / evaluate path is launched by `accelerate`, which owns process creation
(/root/reference/preprocessing/embed.py:55-56,68); here `--gpus N` without WORLD_SIZE in the environment does the same:
the launcher process starts N workers (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT), rank 0
prints the JSON line, the launcher returns the first non-zero exit code.

One STEP = one pass of the hot path over one batch of synthetic input already resident in HBM, in the PRODUCT configuration
(round 5: SuperGuessr(exact_top1=True) -- every discrete output is the fp32 reference's; --fast times the 16-bit path alone):
  BASELINE.json configs[3]: 128 panoramas (4 x 3x336x336 = 512 images) per GPU -> ViT-L/14-336 (24 layers, random
  init seed 0) -> token mean -> SuperGuessr geocell head (C = 10 000) -> certainty of every decision the head and the refinement
  will take (pg_head_certainty, pg_refine_forward_ex + pg_refine_certainty) -> ONE exact re-encode (pg_vit_forward_precise) of the
  panoramas that are not certain, their head outputs recomputed -> [N>1: one grouped RCCL all-gather of embeddings /
  candidates through the C ABI] -> ProtoRefiner top-5 over a 1M x 1024 fp32 prototype bank (10 000 cells x 100) on the
  rank's slice -> [N>1: a second, tiny grouped all-gather of the refined (lng,lat) / geocell, so every rank holds the batch].
Weak scaling: every rank processes its own 128 panoramas; value = total images / max-over-ranks time.
A few distinct pixel batches are resident and used in turn, and the head is centred on the mean embedding (calibrated
during warm-up), so that the synthetic panoramas spread over the geocells the way real ones do and the refinement really
streams different prototype rows from HBM every step (otherwise every query asks for the same cells and the 256 MB
Infinity Cache serves them).

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  certainty           the rule, the calibration's measurements, panoramas re-encoded per step and why; fast_mode / exact_cost_vs_fast:
                      the 16-bit path alone for a few steps after the timed region; per_rank_split_ms: compute vs gather(-wait);
  roofline            the dominant kernel (the fc1 GEMM), algorithmic FLOPs per launch / mean launch time measured live with
                      HIP events on the launch stream during the timed region; `frac_rocprof` = the same fraction from the
                      committed rocprofv3 kernel stats of this command (profiles/rNN/traffic.json);
  roofline_refine     the refinement kernels against the HBM roofline: algorithmic bytes (4096 B per bank row streamed,
                      SURVEY 8d formula, counted by the kernel itself) / mean time between stream events;
  gathered_results    what rank 0 holds after the last step: refined cells of ALL ranks, restored to sample order;
  h2d_inclusive       the realistic ingest leg: uint8 (N,640,640,3) images in pinned host memory -> H2D on a side stream ->
                      pg_prep_forward (resize / crop / normalise, fp16 out) -> the step, double-buffered with events; beside it
                      round 2's number (fp32 pixels copied on the compute stream, no overlap);
  other_configs       BASELINE configs[0..2] shapes on this path;
  secondary_baseline  (rank 0, N=1) stock PyTorch-ROCm on the same box in the same run: the HuggingFace CLIPVisionModel the
                      reference calls (fp32 and fp16 autocast) and torch.matmul / SDPA on the five hot shapes (hipBLASLt);
  cpu_baseline        the oracle (oracle/pigeon_oracle.py = CPU restatement of the reference path, kind "port") timed on this
                      box's host cores on a bounded sample: 16 panoramas = 64 images (BASELINE configs[0]'s size), 4 from each
                      resident pixel batch; plus `reference_module`: transformers.CLIPVisionModel -- the module the reference
                      itself calls -- on the CPU on 32 of those images (rank 0, N=1 only);
  parity_vs_oracle_sample   the oracle's answer for those 16 panoramas FROM THE PIXELS against this very run's outputs: per
                      panorama the oracle's top-1 / top-2 logit margin, the HIP - oracle logit deltas of those two cells,
                      whether the argmax differs, and `refined_mismatch_unconditional` (refined cell or (lng, lat) different from
                      the oracle chain's, whatever the reason); the same for the fast mode beside it;
  parity_vs_reference_module_gpu_fp32   the same over ALL 512 panoramas of the resident batches, against the reference's own
                      module in fp32 on this GPU (+ the oracle's head and refinement).
"""
import argparse
import contextlib
import io
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

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
RAW_HW = 640                         # synthetic raw image geometry of the ingest leg (Street View panels are 640 x 640)
try:
    ORIG_AFFINITY = os.sched_getaffinity(0)      # before pin_to_gpu_numa narrows it; the CPU-baseline workers get the whole set back
except (AttributeError, OSError):
    ORIG_AFFINITY = None


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--panoramas", type=int, default=128, help="panoramas per GPU per step")
    ap.add_argument("--cells", type=int, default=10000)
    ap.add_argument("--protos-per-cell", type=int, default=100)
    ap.add_argument("--topk", type=int, default=5)
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--pixel-batches", type=int, default=16, help="distinct resident pixel batches used in turn (16 x 694 MB: the share of "
                    "panoramas the exact tier re-encodes is then that of fresh data, ~2.2 %%, not that of 4 lucky batches)")
    ap.add_argument("--min-flush", type=int, default=0, help="queued panoramas that trigger one exact pass (pigeon_amd.deferred); 0 = one pass quantum")
    ap.add_argument("--pass-quantum", type=int, default=-1,
                    help="an exact pass takes a multiple of this many panoramas (-1 = what fills one round of the CUs: 7 on 256 CUs; 0 = all queued)")
    ap.add_argument("--max-lag", type=int, default=12, help="steps a queued panorama may wait for the exact pass")
    ap.add_argument("--cpu-images", type=int, default=64, help="images in the bounded CPU-baseline / CPU-oracle parity sample (0 = skip); 64 = 16 panoramas")
    ap.add_argument("--cpu-port-images", type=int, default=16, help="images through the oracle restatement (kind 'port') on the CPU (0 = skip)")
    ap.add_argument("--cpu-workers", type=int, default=0, help="CPU-baseline worker processes x 16 threads (0 = hardware threads / 16)")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)       # internal: run one CPU-baseline worker from a spec file
    ap.add_argument("--fast", action="store_true",
                    help="time the 16-bit fast mode (SuperGuessr(exact_top1=False): no certainty-driven exact re-encode) as the headline; "
                         "its geocell argmax is NOT guaranteed to be the reference's -- A/B and kernel-profiling runs only")
    ap.add_argument("--fast-steps", type=int, default=10,
                    help="timed steps of the fast-mode leg after the timed region (0 = skip); the leg runs every resident batch anyway (the "
                         "parity legs want them), so timing 10 of them instead of round 5's 3 costs nothing and steadies exact_cost_vs_fast")
    ap.add_argument("--raster-gn", type=int, default=None, help="pg_tune_gemm_raster for this run (A/B of the 384x256 GEMM raster)")
    ap.add_argument("--weights", choices=["default", "spread"], default="default",
                    help="default: HF-init tower (seed 0), head centred on the mean embedding and scaled to sigma(logit) = 4 (BASELINE's synthetic "
                         "workload).  spread: synthetic.make_vit_weights_spread (input-selected global attention: embeddings spread like a trained "
                         "tower's, cos-sim ~0.6) with the head at its NATURAL scale -- the parity legs then answer the top-1 question on "
                         "non-degenerate embeddings (not the BASELINE line: say so in config.workload)")
    ap.add_argument("--no-refine", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip ingest / other configs / secondary baseline (profiling runs)")
    ap.add_argument("--profile", choices=["graph", "all", "dominant", "none"], default="graph",
                    help="graph (default): the timed region runs the product path -- encoder body replayed from its hipGraph, no events "
                         "inside -- and per-kernel HIP events bracket every launch of --profile-steps extra un-graphed steps right after it; "
                         "all / dominant: events around every launch / the dominant GEMM class INSIDE the timed region (graph off); "
                         "none: un-graphed, no events (A/B arm)")
    ap.add_argument("--profile-steps", type=int, default=5, help="bracketed un-graphed steps after the timed region (--profile graph / none / dominant)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: stub model / refiner on the CPU, gloo collectives, tiny tensors -- the launch, sharding, gather, "
                         "timing and JSON control flow of the real run (CPU contract test)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------ self launch
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args) -> int:
    """`--gpus N` without a launcher: start N workers of this very command, one per GPU, as accelerate / torchrun would."""
    port = _free_port()                                         # kept in the environment for tools that read it; the ranks' control-plane
    rdzv = os.path.join(tempfile.mkdtemp(prefix="pigeon_rdzv_"), "store")   # group rendezvouses through this FILE store (no port race)
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PIGEON_RDZV_FILE=rdzv, PIGEON_BENCH_LAUNCHER="self")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = set(range(len(procs)))
        while pending:
            for i in sorted(pending):
                r = procs[i].poll()
                if r is None:
                    continue
                pending.discard(i)
                if r != 0 and rc == 0:
                    rc = r
                    print(f"[bench launcher] rank {i} exited with {r}; stopping the other ranks", file=sys.stderr)
                    for j in pending:
                        procs[j].terminate()                     # exact children of this process, nothing else
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ------------------------------------------------------------------------------------------------------ CPU legs
class _LazyRows:
    """Row accessor over a device matrix that materialises only the rows asked for (cpu_baseline leg)."""

    def __init__(self, t):
        self.t = t

    def __getitem__(self, idx):
        import torch
        if isinstance(idx, np.ndarray):
            idx = torch.from_numpy(idx)
        if torch.is_tensor(idx):
            idx = idx.to(self.t.device)
        return self.t[idx].cpu()


def cpu_quota_cores():
    """CPUs' worth of time this container may use: min(visible cpus, cgroup v2 cpu.max quota / period, cgroup v1 cfs quota)."""
    n = float(len(os.sched_getaffinity(0)))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, float(q) / float(per))
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, q / per)
        except (OSError, ValueError):
            pass
    return n


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_worker_main(spec_path):
    """One CPU-baseline worker process (`bench.py --cpu-worker spec.json`): pins itself to its cores, rebuilds the seeded weights,
    runs its shard of images through transformers.CLIPVisionModel (kind "module": the module the reference itself calls,
    models/clip_embedder.py:63) or the oracle restatement (kind "port") and saves the token-mean embeddings.  The timed part
    starts when the parent drops the `go` file, after every worker has loaded its weights."""
    spec = json.load(open(spec_path))
    if spec.get("cpus"):
        try:
            os.sched_setaffinity(0, spec["cpus"])
        except OSError:
            pass
    import torch
    torch.set_num_threads(int(spec["threads"]))
    from pigeon_amd import synthetic
    px = torch.load(spec["pixels"])
    sd = (synthetic.make_vit_weights_spread(seed=31, layers=spec["layers"]) if spec.get("weights") == "spread"
          else synthetic.make_vit_weights(seed=spec["weight_seed"], layers=spec["layers"]))
    if spec["kind"] == "module":
        from transformers import CLIPVisionConfig, CLIPVisionModel
        cfg = CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=spec["layers"], num_attention_heads=16,
                               image_size=336, patch_size=14, projection_dim=768)
        with contextlib.redirect_stdout(io.StringIO()):
            hf = CLIPVisionModel(cfg)
        hf.load_state_dict(sd, strict=True)
        hf.eval()

        def run(x):
            with torch.no_grad():
                return hf(pixel_values=x).last_hidden_state.mean(dim=1)
    else:
        from oracle import pigeon_oracle as orc

        def run(x):
            return orc.clip_embedding(sd, x)
    open(spec["ready"], "w").close()
    t_wait = time.time() + 600
    while not os.path.exists(spec["go"]):
        if time.time() > t_wait:
            raise SystemExit("cpu worker: no go signal")
        time.sleep(0.01)
    t0 = time.time()
    bs = int(spec.get("batch", 4))
    outs = [run(px[i:i + bs]) for i in range(0, px.shape[0], bs)]
    t1 = time.time()
    torch.save(torch.cat(outs), spec["out"])
    json.dump({"seconds": t1 - t0, "end": t1, "images": int(px.shape[0])}, open(spec["done"], "w"))


def cpu_pool(kind, px_images, layers, workers, threads, weight_seed=0, weights="default"):
    """`workers` independent processes x `threads` torch threads on disjoint core slices and disjoint image shards (torch's CPU
    GEMMs regress past 32 threads in ONE process -- a fact about one process, not about the box).  Returns (embeddings in image
    order, wall seconds from the common go signal to the last worker's result, per-worker seconds, cores busy)."""
    import torch
    n = px_images.shape[0]
    workers = max(1, min(workers, n))
    avail = sorted(ORIG_AFFINITY or os.sched_getaffinity(0))     # not the GPU-side pinning of the bench process itself
    threads = max(1, min(threads, len(avail) // workers if len(avail) >= workers else 1))
    tmp = tempfile.mkdtemp(prefix="pigeon_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    bounds = [round(i * n / workers) for i in range(workers + 1)]
    procs, specs = [], []
    go = os.path.join(tmp, "go")
    for w in range(workers):
        lo, hi = bounds[w], bounds[w + 1]
        spec = {"kind": kind, "threads": threads, "layers": layers, "weight_seed": weight_seed, "weights": weights, "cpus": avail[w * threads:(w + 1) * threads],
                "pixels": os.path.join(tmp, f"px_{w}.pt"), "out": os.path.join(tmp, f"emb_{w}.pt"), "ready": os.path.join(tmp, f"ready_{w}"),
                "done": os.path.join(tmp, f"done_{w}.json"), "go": go, "batch": 4}
        torch.save(px_images[lo:hi].clone(), spec["pixels"])
        sp = os.path.join(tmp, f"spec_{w}.json")
        json.dump(spec, open(sp, "w"))
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", sp], env=env,
                                      stdout=subprocess.DEVNULL, stderr=subprocess.PIPE))
        specs.append(spec)
    try:
        t_end = time.time() + 900
        while not all(os.path.exists(sp["ready"]) for sp in specs):
            bad = [i for i, pr in enumerate(procs) if pr.poll() not in (None, 0)]
            if bad or time.time() > t_end:
                err = procs[bad[0]].stderr.read().decode()[-400:] if bad else "timeout while loading"
                raise RuntimeError(f"cpu worker {bad} failed before the go signal: {err}")
            time.sleep(0.05)
        t0 = time.time()
        open(go, "w").close()
        for i, pr in enumerate(procs):
            if pr.wait(timeout=1800) != 0:
                raise RuntimeError(f"cpu worker {i} failed: {pr.stderr.read().decode()[-400:]}")
        emb = torch.cat([torch.load(sp["out"]) for sp in specs])
        done = [json.load(open(sp["done"])) for sp in specs]
        per = [d["seconds"] for d in done]
        wall = max(d["end"] for d in done) - t0              # go signal -> last worker's last image (same host clock); not the
                                                             # interpreter teardown of the worker processes
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return emb, wall, per, workers * threads


def cpu_baseline(args, model, bank_t, px):
    """The reference's CPU path on THIS box's host cores, on a bounded sample of the SAME workload; px (S,12,336,336) on the host.
    Encoder: transformers.CLIPVisionModel -- the module the reference itself calls -- in hardware_threads / 16 independent
    16-thread processes on disjoint shards (kind "reference-module"); head + refinement: the oracle restatement on those
    embeddings (serial, added to the time).  Sub-field `port`: the oracle's own ViT restatement on --cpu-port-images of them.
    Returns (json dict, oracle head outputs, oracle refined (llh, cell, bank) or None)."""
    import torch
    from oracle import pigeon_oracle as orc
    npano = px.shape[0]
    cores = os.cpu_count() or 1
    quota = cpu_quota_cores()
    # the host's USABLE cores: the GPU boxes of this pool show 256 hardware threads but run the container under a cgroup quota of 16
    # CPUs (cpu.max 1600000 100000; tools/cpu_probe.sh: 1 x 16 threads 1.40 TFLOP/s fp32, 16 x 16 threads 0.97 with 917 throttled
    # periods) -- threads beyond the quota only add throttling, so the pool is sized to the quota, in 16-thread processes
    workers = args.cpu_workers if args.cpu_workers > 0 else max(1, int(quota) // 16)
    images = px.reshape(-1, 3, 336, 336)
    emb_i, wall, per, busy = cpu_pool("module", images, args.layers, workers, 16, weights=args.weights)
    emb = emb_i.reshape(npano, 4, 1024)
    W = model.cell_layer.weight.data.cpu()
    b = model.cell_layer.bias.data.cpu()
    cen = model.lla_geocells.data.cpu()
    torch.set_num_threads(min(cores, 32))
    t0 = time.time()
    o = orc.super_guessr_forward(W, b, cen, args.topk, embedding=emb)
    refined = None
    if bank_t is not None:
        class B:
            pass
        hb = B()
        hb.proto_emb = _LazyRows(bank_t["proto_emb"])
        hb.train_emb = _LazyRows(bank_t["train_emb"])
        for k in ("cell_off", "proto_count", "member_off", "member_idx"):
            setattr(hb, k, bank_t[k].cpu().numpy())
        hb.proto_lnglat = bank_t["proto_lnglat"].cpu().numpy()
        hb.train_lnglat = _LazyRows(bank_t["train_lnglat"])
        _, r_llh, r_cell = orc.proto_refiner_forward(hb, o["embedding"], o["preds_LLH"], o["topk"].indices, o["topk"].values,
                                                     args.topk, 1.6, 1000)
        refined = (r_llh, r_cell, hb)
    t_tail = time.time() - t0
    dt = wall + t_tail
    res = {"value": npano * 4 / dt, "unit": "images/s", "cores": busy, "box_cores": cores, "cgroup_cpu_quota_cores": quota,
           "kind": "reference-module",
           "workers": len(per), "threads_per_worker": busy // max(1, len(per)),
           "encoder_images_per_s": npano * 4 / wall, "per_worker_seconds": [round(x, 2) for x in per],
           "sample": f"{npano} panoramas = {npano * 4} images of this run's resident pixel batches through transformers.CLIPVisionModel "
                     f"(ViT-L/14-336 config, this run's weights; the module the reference calls) fp32 + token mean in {len(per)} "
                     f"independent processes x {busy // max(1, len(per))} threads on disjoint cores and image shards ({wall:.1f} s from a "
                     f"common start signal, weights already loaded), then the oracle's head + top-{args.topk} refinement on those "
                     f"embeddings ({t_tail:.1f} s, serial); linear in images",
           "transformers": __import__("transformers").__version__, "cpu_model": _cpu_model()}
    if args.cpu_port_images > 0:
        try:
            n_port = min(args.cpu_port_images, images.shape[0])
            # spread over the sample so that the restatement is checked against the module on every resident batch
            sel = torch.linspace(0, images.shape[0] - 1, n_port).round().long()
            emb_p, wall_p, per_p, busy_p = cpu_pool("port", images[sel], args.layers, min(workers, n_port), 16, weights=args.weights)
            res["port"] = {"value": n_port / wall_p, "unit": "images/s", "cores": busy_p, "kind": "port",
                           "sample": f"{n_port} of those images through oracle/pigeon_oracle.py (the CPU restatement of the reference's "
                                     f"encoder path), same process layout, {wall_p:.1f} s",
                           "embedding_rel_err_vs_module": orc.rel_err(emb_p, emb_i[sel])}
        except Exception as e:  # noqa
            res["port"] = {"error": repr(e)}
    return res, o, refined


def flip_analysis(ref_logits, hip_logits, hip_cell, where):
    """Pure part of the parity leg (CPU-testable): per panorama the reference's top-1 / top-2 logit margin, the HIP - reference
    deltas of those two logits and the flip flag.  A flip is EXPLAINED only if the reference margin is below twice the largest
    logit error measured anywhere in the sample (the margin itself moves by at most |d_top1| + |d_top2|)."""
    import torch
    ref_logits, hip_logits = ref_logits.double(), hip_logits.double()
    top2 = torch.topk(ref_logits, 2, dim=-1)
    margin = top2.values[:, 0] - top2.values[:, 1]
    delta = hip_logits - ref_logits
    err_max = float(delta.abs().max())
    flips = hip_cell != top2.indices[:, 0]
    rows = []
    for i in range(ref_logits.shape[0]):
        c1, c2 = int(top2.indices[i, 0]), int(top2.indices[i, 1])
        rows.append({"panorama": where[i], "oracle_cell": c1, "hip_cell": int(hip_cell[i]), "flip": bool(flips[i]),
                     "oracle_margin": round(float(margin[i]), 5), "d_top1": round(float(delta[i, c1]), 5),
                     "d_top2": round(float(delta[i, c2]), 5)})
    unexplained = [r for r in rows if r["flip"] and r["oracle_margin"] >= 2 * err_max]
    return {"logit_abs_err_max": err_max, "logit_sigma": float(ref_logits.std()),
            "oracle_margin_min": float(margin.min()), "oracle_margin_median": float(margin.median()),
            "flips": int(flips.sum()), "flips_unexplained": len(unexplained),
            "flip_rule": "a flip is explained only if the oracle's top-1/top-2 logit margin is below 2 x logit_abs_err_max "
                         "(the margin moves by at most |d_top1| + |d_top2|)",
            "geocell_argmax_equal": bool(not flips.any()),
            "flipped": [r for r in rows if r["flip"]],
            "smallest_margins": sorted(rows, key=lambda r: r["oracle_margin"])[:4]}, flips


def parity_report(args, dev, model, o, refined, hip):
    """The oracle's answer from the PIXELS against this run's outputs for the same panoramas."""
    import torch
    from oracle import pigeon_oracle as orc
    from pigeon_amd import hip_ops
    n = o["embedding"].shape[0]
    # the product head kernel on this run's embeddings: the logits the step's argmax was taken from
    ho = hip_ops.head_forward(hip["embedding"].to(dev).contiguous(), model.cell_layer.weight.data, model.cell_layer.bias.data,
                              model.lla_geocells.data, args.topk)
    assert torch.equal(ho["preds_geocell"].cpu(), hip["preds_geocell"].cpu()), "head is not batch-position independent"
    fa, flips = flip_analysis(o["logits"], ho["logits"].cpu(), hip["preds_geocell"].cpu(), hip["where"])
    assert torch.equal(torch.topk(o["logits"], 1, dim=-1).indices[:, 0], o["preds_geocell"])
    rep = {"n_panoramas": n, "from": "pixels (oracle ViT fp32 on the host) vs this run's step outputs",
           "embedding_rel_err": orc.rel_err(hip["embedding"].cpu(), o["embedding"]),
           "oracle": "transformers.CLIPVisionModel fp32 on the host (cpu_baseline's encoder leg) + the oracle's head / refinement",
           "embedding_rel_err_worst_image": orc.max_rel_err_rows(hip["embedding"].cpu().reshape(-1, 1024), o["embedding"].reshape(-1, 1024))}
    rep.update(fa)
    if refined is not None and "refined_geocell" in hip:
        r_llh, r_cell, _ = refined
        keep = ~flips
        # where the head agrees, refinement consumed (almost) the same candidates: cells must agree unless a candidate
        # below rank 1 traded places (reported, not asserted -- tests/test_gpu_entrypoints.py asserts it on the fixtures)
        rep["refined_cell_equal_where_argmax_equal"] = f"{int((hip['refined_geocell'].cpu()[keep] == r_cell[keep]).sum())}/{int(keep.sum())}"
        same_ll = (hip["refined_LLH"].cpu()[keep] == r_llh[keep]).all(dim=1)
        rep["refined_lnglat_equal_where_argmax_equal"] = f"{int(same_ll.sum())}/{int(keep.sum())}"
        # ... and UNCONDITIONALLY: panoramas whose refined cell or (lng, lat) differs from the reference chain's, whatever the reason
        bad = (hip["refined_geocell"].cpu() != r_cell) | (hip["refined_LLH"].cpu() != r_llh).any(dim=1)
        rep["refined_mismatch_unconditional"] = int(bad.sum())
        rep["refined_mismatched"] = [hip["where"][i] for i in torch.nonzero(bad).flatten().tolist()][:16]
    return rep


def gpu_module_parity(args, dev, vit_sd, model, pixel_batches, used, outs_default, certain_by_batch, outs_other, other_name, cpu_sample_emb, per,
                      bank_t=None):
    """Top-1 parity over EVERY panorama of the resident pixel batches (4 x 128 = 512), which the 16-CPU quota of these boxes puts out
    of the CPU oracle's reach: the reference's own module -- transformers.CLIPVisionModel, fp32, eager attention, stock
    PyTorch-ROCm -- runs on this GPU (never part of the product path), the oracle's head turns its embeddings into logits, and the
    default and exact-mode outputs of this run are compared with that.  The CPU oracle's embeddings of the bounded sample pin
    the GPU-fp32 module first (`vs_cpu_oracle_embedding_rel_err`)."""
    import torch
    from transformers import CLIPVisionConfig, CLIPVisionModel
    from oracle import pigeon_oracle as orc
    cfg = CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=args.layers, num_attention_heads=16,
                           image_size=336, patch_size=14, projection_dim=768)
    with contextlib.redirect_stdout(io.StringIO()):
        hf = CLIPVisionModel._from_config(cfg, attn_implementation="eager")    # explicit matmul / softmax / matmul in fp32
    hf.load_state_dict(vit_sd, strict=True)
    hf = hf.to(dev).eval()
    torch.backends.cuda.matmul.allow_tf32 = False
    embs = []
    t0 = time.perf_counter()
    with torch.no_grad():
        for j in used:
            px = pixel_batches[j].reshape(-1, 3, 336, 336)
            embs.append(torch.cat([hf(pixel_values=px[i:i + 32]).last_hidden_state.mean(dim=1) for i in range(0, px.shape[0], 32)]))
    torch.cuda.synchronize()
    t_ref = time.perf_counter() - t0
    ref_emb = torch.cat(embs).reshape(-1, 4, 1024).cpu()
    del hf
    torch.cuda.empty_cache()
    W, b, cen = model.cell_layer.weight.data.cpu(), model.cell_layer.bias.data.cpu(), model.lla_geocells.data.cpu()
    o = orc.super_guessr_forward(W, b, cen, args.topk, embedding=ref_emb)
    n_per = pixel_batches[0].shape[0]
    where = [f"batch {j} #{i}" for j in used for i in range(n_per)]

    def take(outs):
        h = {"embedding": torch.cat([outs[j]["embedding"] for j in used]), "preds_geocell": torch.cat([outs[j]["preds_geocell"] for j in used]),
             "where": where}
        if "refined_geocell" in outs[used[0]]:
            h["refined_geocell"] = torch.cat([outs[j]["refined_geocell"] for j in used])
            h["refined_LLH"] = torch.cat([outs[j]["refined_LLH"] for j in used])
        return h
    refined = None
    if bank_t is not None:
        # the reference chain's refinement of ALL these panoramas: the oracle's restatement (bit-exact against the real reference on
        # the fixtures) on the reference module's embeddings and the oracle head's candidates
        class _B:
            pass
        hb = _B()
        hb.proto_emb, hb.train_emb, hb.train_lnglat = _LazyRows(bank_t["proto_emb"]), _LazyRows(bank_t["train_emb"]), _LazyRows(bank_t["train_lnglat"])
        for kk in ("cell_off", "proto_count", "member_off", "member_idx"):
            setattr(hb, kk, bank_t[kk].cpu().numpy())
        hb.proto_lnglat = bank_t["proto_lnglat"].cpu().numpy()
        t1 = time.perf_counter()
        _, r_llh, r_cell = orc.proto_refiner_forward(hb, o["embedding"], o["preds_LLH"], o["topk"].indices, o["topk"].values, args.topk, 1.6, 1000)
        refined = (r_llh, r_cell, hb)
    rep = parity_report(args, dev, model, o, refined, take(outs_default))
    if refined is not None:
        rep["reference_refine_seconds"] = time.perf_counter() - t1
    rep["from"] = "pixels through transformers.CLIPVisionModel fp32 (eager attention, stock PyTorch-ROCm) on this GPU vs this run's step outputs"
    rep["oracle"] = "the reference's module in fp32 on the GPU + the oracle's head; pinned to the CPU oracle on the bounded sample below"
    rep["reference_seconds"] = t_ref
    cert = torch.cat([certain_by_batch[j][0] for j in used]).cpu()
    rep["certain"] = f"{int(cert.sum())}/{cert.numel()}"
    rep["flips_among_certain"] = int(sum(1 for r in rep["flipped"] if bool(cert[where.index(r["panorama"])])))
    rep.pop("smallest_margins", None)
    if outs_other and all(j in outs_other for j in used):
        rx = parity_report(args, dev, model, o, refined, take(outs_other))
        rep[other_name] = {k: rx[k] for k in ("embedding_rel_err", "logit_abs_err_max", "flips", "flipped", "geocell_argmax_equal",
                                              "refined_mismatch_unconditional", "refined_mismatched") if k in rx}
    if cpu_sample_emb is not None:
        sel = torch.cat([ref_emb[k * n_per:k * n_per + per] for k in range(len(used))])
        rep["vs_cpu_oracle_embedding_rel_err"] = orc.rel_err(sel, cpu_sample_emb)
    return rep


# ------------------------------------------------------------------------------------------------------ GPU side legs
def _committed_traffic(kernel, rows):
    """Memory-side bytes per launch of a kernel class from the committed rocprofv3 PMC pass (profiles/rNN/traffic.json:
    2 x FETCH_SIZE -- the gfx950 correction of MI355X_MICROARCH.md -- + WRITE_SIZE, separate --pmc passes over this very
    command, tools/prof_bench.sh) and, where recorded, the rocprofv3 average launch duration.  Counters cannot be read from
    inside the process; (None, None) if there is no pass for this launch size."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
            if kernel in d and "hbm_bytes_per_launch_corrected" in d[kernel] and d[kernel].get("rows") == rows:
                return d[kernel]["hbm_bytes_per_launch_corrected"], {
                    "algorithmic_bytes": d[kernel].get("algorithmic_bytes"), "source": os.path.relpath(f, ROOT),
                    "rocprof_avg_ms": d[kernel].get("rocprof_avg_ms"), "rocprof_tail_avg_ms": d[kernel].get("rocprof_tail_avg_ms"),
                    "note": "counts L2-miss traffic on the fabric side (Infinity Cache hits included)"}
        except (OSError, ValueError):
            pass
    return None, None


def _time_gpu(fn, iters, warm=1):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def ingest_leg(args, dev, pipe, index, resident_ms):
    """The realistic feed: decoded uint8 RGB images in pinned host memory -> H2D on a side stream -> pg_prep_forward (Pillow-exact
    bicubic resize-336 / crop / normalise, fp16 out; the step the reference does on the host with CLIPProcessor,
    dataset_creation/finetune/embed_dataset.py:17-22) -> the step.  Two host / staging / pixel buffers; the copy of batch i+1
    runs under the compute of batch i, ordered with events.  Timed region = everything from host memory to refined output."""
    import torch
    from pigeon_amd import hip_ops
    n_img = args.panoramas * 4
    prep = hip_ops.Preprocessor(RAW_HW, RAW_HW, device=dev.index or 0)
    g = torch.Generator().manual_seed(99)
    host = [torch.randint(0, 256, (n_img, RAW_HW, RAW_HW, 3), dtype=torch.uint8, generator=g).pin_memory() for _ in range(2)]
    stage = [torch.empty((n_img, RAW_HW, RAW_HW, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
    copy_stream = torch.cuda.Stream(device=dev)
    copied = [torch.cuda.Event() for _ in range(2)]
    consumed = [torch.cuda.Event() for _ in range(2)]
    main = torch.cuda.current_stream(dev)

    def issue_copy(i):
        b = i % 2
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(consumed[b])                    # the prep kernel that read stage[b] two steps ago is done
            stage[b].copy_(host[b], non_blocking=True)
            copied[b].record(copy_stream)

    def run(iters):
        for b in range(2):
            consumed[b].record(main)
        issue_copy(0)
        for i in range(iters):
            b = i % 2
            if i + 1 < iters:
                issue_copy(i + 1)                                  # next batch's H2D under this batch's compute
            main.wait_event(copied[b])
            px16 = prep(stage[b], out_dtype=torch.float16)         # (512,3,336,336) fp16
            consumed[b].record(main)
            for r in pipe.submit(px16.view(args.panoramas, 12, 336, 336), index):
                n_re.append(int(r["queued"][0]))
        for r in pipe.flush():                                     # inside the timed region: every step's outputs are final
            n_re.append(int(r["queued"][0]))

    n_re = []
    run(2)
    torch.cuda.synchronize()
    iters = max(4, min(args.steps, 6))
    n_re.clear()
    t0 = time.perf_counter()
    run(iters)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / iters
    mb = n_img * RAW_HW * RAW_HW * 3 / 1e6
    res = {"value": n_img / t, "unit": "images/s", "ms_per_step": t * 1e3, "n_gpus": 1,
           "frac_of_resident": (resident_ms / 1e3) / t,
           "reencoded_panoramas_per_step": n_re,
           "what": f"uint8 ({n_img},{RAW_HW},{RAW_HW},3) images in pinned host memory ({mb:.0f} MB per step) -> H2D on a side stream, "
                   "double-buffered with events -> pg_prep_forward (bit-exact CLIPProcessor resize / crop / normalise, fp16 out) -> "
                   "ViT + head + refine; whole chain inside the timed region.  NOTE: these are other images than the resident "
                   "batches (uniform random uint8), so the exact mode re-encodes another number of panoramas per step "
                   "(`reencoded_panoramas_per_step`) than in the timed region; `frac_of_fast_mode_resident` compares with the fast "
                   "mode's resident step (no re-encodes on either side)"}
    del host, stage
    prep.close()
    return res


def spread_tower_leg(args, dev, bank_t, pixel_batches, index):
    """The same step on the tower whose embeddings spread like a trained one's (synthetic.make_vit_weights_spread(seed 31): a quarter of
    the heads at high q.k gain, image cos-sim 0.4 .. 0.75, 16-bit embedding error 7e-4 = the closest stand-in for a trained CLIP this
    box can build), with the head centred on the mean embedding and the refiner ON: the product (deferred exact tier) and the 16-bit
    path alone, 8 + 3 steps on the resident pixel batches.  Not the BASELINE line (other weights): an `other_configs` entry."""
    import torch
    from pigeon_amd import synthetic
    from pigeon_amd.clip_embedder import HipCLIPVisionModel
    from pigeon_amd.deferred import LocalComm
    from pigeon_amd.evaluate import PanoramaPipeline
    from pigeon_amd.proto_refiner import ProtoRefiner
    from pigeon_amd.super_guessr import SuperGuessr
    nb = len(pixel_batches)
    sd = synthetic.make_vit_weights_spread(seed=31, layers=args.layers)
    base = HipCLIPVisionModel(sd, layers=args.layers)
    geo_csv = os.path.join(tempfile.mkdtemp(prefix="pigeon_bench_spread_"), "geocells.csv")
    synthetic.write_geocell_csv(geo_csv, synthetic.make_geocells(args.cells, seed=0))
    with contextlib.redirect_stdout(io.StringIO()):
        model = SuperGuessr(base, panorama=True, freeze_base=True, num_candidates=args.topk, geocell_path=geo_csv, exact_top1=True,
                            margin_autocalibrate=False)
    W, b = synthetic.make_head_weights(args.cells, seed=0)
    with torch.no_grad():
        model.cell_layer.weight.copy_(W)
        model.cell_layer.bias.copy_(b)
    model.to(dev).eval()
    refiner = ProtoRefiner(topk=args.topk, max_refinement=1000, temperature=1.6, bank=bank_t, device=str(dev)).eval()
    pipe = PanoramaPipeline(model, refiner, LocalComm(), min_flush=args.min_flush or None, max_lag=args.max_lag,
                            pass_quantum=None if args.pass_quantum < 0 else args.pass_quantum)
    model.exact_top1 = False
    out = pipe.step(pixel_batches[0], index)
    with torch.no_grad():                                           # centre the head (its natural scale is kept: these embeddings spread)
        center = out["embedding"].mean(dim=1).mean(dim=0)
        model.cell_layer.bias.copy_(b.to(dev) - model.cell_layer.weight.data @ center)
    model.exact_top1 = True
    model.calibrate_certainty(pixel_batches[nb - 1], max_samples=args.panoramas)

    def run(steps):
        pipe.submit(pixel_batches[0], index)
        pipe.flush()
        torch.cuda.synchronize()
        n_re, cells = [], []
        t0 = time.perf_counter()
        for i in range(steps):
            for r in pipe.submit(pixel_batches[i % nb], index):
                n_re.append(int(r["queued"][0])); cells.append(r["preds_geocell"])      # device tensors: nothing here waits for the GPU
        for r in pipe.flush():
            n_re.append(int(r["queued"][0])); cells.append(r["preds_geocell"])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        return dt, n_re, int(torch.unique(torch.cat(cells)).numel())
    t_x, n_re, n_cells = run(8)
    model.exact_top1 = False
    t_f, _, _ = run(3)
    st = model.certainty.stats
    return {"workload": "NOT the BASELINE line: configs[3] on synthetic.make_vit_weights_spread(seed 31) -- embeddings spread like a trained tower's "
                        "-- head centred on the mean embedding (natural scale), ProtoRefiner top-5 over the same 1Mx1024 bank",
            "value": args.panoramas * 4 / t_x, "unit": "images/s", "ms_per_step": t_x * 1e3, "steps": 8,
            "mfma_frac": args.panoramas * 4 / t_x * FLOP_PER_IMAGE / PEAK_MFMA,
            "fast_mode": {"value": args.panoramas * 4 / t_f, "ms_per_step": t_f * 1e3, "steps": 3}, "exact_cost_vs_fast": t_x / t_f,
            "reencoded_panoramas_per_step": n_re, "reencoded_share": float(np.sum(n_re)) / max(1, args.panoramas * len(n_re)),
            "distinct_argmax_cells": n_cells,
            "calibration": {k: st.get(k) for k in ("fast_vs_exact_rms", "drift_norm", "residual_rms", "image_rel_err", "worst_image_rel_err",
                                                  "force_exact", "rel_tol")}}


def secondary_baseline(dev, vit_sd, layers):
    """Stock PyTorch-ROCm on the same GPU, same run -- NOT the product path, never imported by pigeon_amd:
    (a) transformers.CLIPVisionModel (the module the reference calls at models/clip_embedder.py:63 /
        models/super_guessr.py:395) + token mean, fp32 and fp16 autocast, 128 images per step;
    (b) torch.matmul (hipBLASLt) on the four GEMM shapes of a 512-image chunk and SDPA on its attention shape, fp16."""
    import torch
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


# ------------------------------------------------------------------------------------------------------ dry-run stubs
def _cause_counts(causes):
    """certain_forward's `cause` codes of the timed steps -> {cause: panoramas}, summed over the steps."""
    names = {1: "head top-1", 2: "refiner: candidate-set boundary", 3: "refiner: nearest prototype", 4: "refiner: farthest member"}
    out = {}
    for c in causes:
        for v in c[c != 0].tolist():
            key = "head top-1" if v == 1 else ("refiner: winning candidate" if 1000 <= v < 2000 else
                                               names.get(v // 1000, "refiner: fp32 underflow" if v == -9 else f"code {v}"))
            out[key] = out.get(key, 0) + 1
    return out


def _dry_stubs(args):
    """CPU stand-ins with the call surface pigeon_amd.deferred.DeferredExact uses (pigeon_amd.SuperGuessr / ProtoRefiner on this
    path) plus the torch restatement of csrc/requeue.hip (oracle/requeue_oracle.py) as its `ops`.  They exist so that the launch / shard /
    gather / restore-order / timing / JSON control flow -- INCLUDING the deferred exact tier's protocol: rank r finds r panoramas
    uncertain per step, the queue lengths ride in the step's second gather, every rank flushes in the same step on the same number of
    slots and pays 2 ms per slot -- can be exercised without a GPU (tests/test_bench_dry_run.py); nothing here is a fallback of the
    product: the real run never constructs them."""
    import torch
    from oracle import requeue_oracle
    from pigeon_amd.certainty import Certainty
    from pigeon_amd.utils import ModelOutput, TopK
    g = torch.Generator().manual_seed(5)
    P = torch.randn((3, 1024), generator=g)
    W = torch.randn((args.cells, 1024), generator=g) * 0.05
    cen = torch.rand((args.cells, 2), generator=g, dtype=torch.float64) * 100
    k = args.topk
    rank = int(os.environ.get("RANK", "0"))

    class Model:
        cell_layer = torch.nn.Linear(1024, args.cells)
        lla_geocells = torch.nn.Parameter(cen, requires_grad=False)
        exact_top1 = not args.fast
        num_candidates = k
        certainty = Certainty()
        exact_log = []                                            # (slots, seconds) of every exact pass of this rank

        def wstats(self, exact=False):
            return torch.stack([W.norm(dim=1).max(), torch.zeros(())])

        def _head(self, emb):
            probs = torch.softmax(emb.mean(dim=1) @ W.t(), dim=-1)
            top = torch.topk(probs, k + min(4, max(0, args.cells - k)), dim=-1)           # k > cells raises, as the real head does
            cells = top.indices[:, 0].contiguous()
            n = emb.shape[0]
            return dict(embedding=emb, topk_values=top.values, topk_indices=top.indices, preds_geocell=cells, preds_LLH=cen[cells],
                        tol=torch.ones(n), margin=torch.ones(n), sens=torch.ones(n))

        def encode_head(self, pixel_values=None, embedding=None):
            B = pixel_values.shape[0]
            st = self._head(pixel_values.reshape(B, 4, 3, -1).float().mean(dim=-1) @ P)         # (B,4,1024)
            st["tol"][:min(rank, B)] = 0.0                                                      # rank r: r uncertain panoramas per step
            st.update(pixel_values=pixel_values, exact_tier=False, thr=0.5, drift=None, wstats=self.wstats())
            return st

        def exact_rows(self, pixel_rows):
            px = torch.cat(list(pixel_rows))
            t0 = time.perf_counter()
            time.sleep(2e-3 * px.shape[0])                                                      # the exact tier's cost, per SLOT run
            self.exact_log.append((int(px.shape[0]), time.perf_counter() - t0))
            return self._head(px.reshape(px.shape[0], 4, 3, -1).float().mean(dim=-1) @ P)

        def package(self, st, labels=None, labels_clf=None):
            return ModelOutput(None, None, 0, 0, 0, st["preds_LLH"], st["preds_geocell"], None, None, None,
                               TopK(st["topk_values"][:, :k], st["topk_indices"][:, :k]), st["embedding"])

    class Refiner:
        last_scratch = None

        def forward_certain(self, emb, initial_preds, candidate_cells, candidate_probs, head_weight, wstats, drift=None):
            B = emb.shape[0]
            return ((initial_preds + 0.25).float(), candidate_cells[:, min(1, k - 1)].contiguous(), torch.full((B,), 1e9),
                    torch.zeros(B, dtype=torch.int32), True)

        def __call__(self, emb, initial_preds=None, candidate_cells=None, candidate_probs=None, quiet=False):
            return None, (initial_preds + 0.25).float(), candidate_cells[:, min(1, k - 1)].contiguous()

    return Model(), (None if args.no_refine else Refiner()), requeue_oracle


# ------------------------------------------------------------------------------------------------------ worker
def worker(args):
    from pigeon_amd import distributed
    # control plane (gloo); the data-path collective is RCCL through the C ABI.  --dry-run never touches a GPU, also on a GPU box
    comm = distributed.init_from_env(set_device=not args.dry_run)
    # N = 1 included: the two grouped all-gathers of a step go through a (1-rank) RCCL communicator, so that a single-GPU line
    # exercises csrc/comm.hip and the RCCL binding exactly as the N > 1 lines do (PIGEON_FORCE_RCCL=0: the identity shortcut)
    comm.force_rccl = os.environ.get("PIGEON_FORCE_RCCL", "1") not in ("", "0") and not args.dry_run
    # every rank, on every way out: the gloo group is torn down explicitly (alive at interpreter exit it aborts the process --
    # a finished rank would then fail its launcher); the RCCL communicator only on the regular way out (_worker does it after
    # its last barrier), a failing rank abandons it
    try:
        _worker(args, comm)
    except BaseException:
        comm.close(rccl=False)
        raise
    comm.close()


def _worker(args, comm):
    import torch
    from pigeon_amd import distributed
    from pigeon_amd.evaluate import PanoramaPipeline

    rank, world = comm.rank, comm.world_size
    if os.environ.get("PIGEON_BENCH_TEST_NOISE"):                # tests/test_bench_dry_run.py: stdout stays ONE line regardless
        print("noise through print")
        os.write(1, b"noise through fd 1\n")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: either leave WORLD_SIZE unset (bench.py then starts "
                         f"its {args.gpus} ranks itself) or launch with torch.distributed.run --nproc-per-node {args.gpus}")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dry = args.dry_run
    nb = max(1, args.pixel_batches)
    base = enc = refiner = bank_t = vit_sd = dry_ops = None
    if dry:
        dev = torch.device("cpu")
        model, refiner, dry_ops = _dry_stubs(args)
        g = torch.Generator().manual_seed(1234 + rank)
        pixel_batches = [torch.randn((args.panoramas, 12, 8, 8), generator=g) for _ in range(nb)]
    else:
        from pigeon_amd import _lib, synthetic
        from pigeon_amd.clip_embedder import HipCLIPVisionModel
        from pigeon_amd.proto_refiner import ProtoRefiner
        from pigeon_amd.super_guessr import SuperGuessr
        _lib.require_gpu()
        if args.raster_gn is not None:
            from pigeon_amd import hip_ops
            hip_ops.tune_gemm_raster(args.raster_gn)
        if local >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but this box has {torch.cuda.device_count()} GPU(s)")
        torch.cuda.set_device(local)
        dev = torch.device(f"cuda:{local}")
        # host side of one-process-per-GPU: this rank's launch thread stays on the cores next to its GPU (8 launch-heavy processes
        # on a 2-socket box otherwise migrate across sockets); PIGEON_BENCH_PIN=0 switches it off
        pinned = None
        if os.environ.get("PIGEON_BENCH_PIN", "1") not in ("", "0"):
            pinned = distributed.pin_to_gpu_numa(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
        # ---- model, head, bank (identical replicas on every rank) ----
        vit_sd = (synthetic.make_vit_weights_spread(seed=31, layers=args.layers) if args.weights == "spread"
                  else synthetic.make_vit_weights(seed=0, layers=args.layers))
        base = HipCLIPVisionModel(vit_sd, layers=args.layers)
        tmp = tempfile.mkdtemp(prefix="pigeon_bench_")
        geo_csv = os.path.join(tmp, f"geocells_{rank}.csv")
        synthetic.write_geocell_csv(geo_csv, synthetic.make_geocells(args.cells, seed=0))
        with contextlib.redirect_stdout(io.StringIO()):
            model = SuperGuessr(base, panorama=True, freeze_base=True, num_candidates=args.topk, geocell_path=geo_csv,
                                exact_top1=not args.fast, margin_autocalibrate=False)
        W, b = synthetic.make_head_weights(args.cells, seed=0)
        with torch.no_grad():
            model.cell_layer.weight.copy_(W)
            model.cell_layer.bias.copy_(b)
        model.to(dev).eval()
        if not args.no_refine:
            bank_t = synthetic.make_bank_device(args.cells, args.protos_per_cell, seed=2, device=str(dev))
            refiner = ProtoRefiner(topk=args.topk, max_refinement=1000, temperature=1.6, bank=bank_t, device=str(dev)).eval()
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        pixel_batches = [torch.randn((args.panoramas, 12, 336, 336), generator=g, device=dev) for _ in range(nb)]   # resident in HBM
    rccl_error = None
    if not dry and world == 1 and comm.force_rccl:
        # the forced 1-rank communicator is an early warning, not a dependency of the single-GPU line: if RCCL cannot come up on
        # this box the step falls back to the identity gather and the line says why
        try:
            comm.gather_many([torch.zeros(4, device=dev)])
        except Exception as e:  # noqa
            rccl_error = repr(e)
            comm.force_rccl = False
    pipe = PanoramaPipeline(model, refiner, comm, min_flush=args.min_flush or None, max_lag=args.max_lag, ops=dry_ops,
                            pass_quantum=None if args.pass_quantum < 0 else args.pass_quantum)
    # sample ids as a sharded DataLoader deals them (batch i -> rank i % world, preprocessing/embed.py:68): interleaved, so the
    # gathered results really need restore_order
    index = (torch.arange(args.panoramas, device=dev) * world + rank)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    # ---- warm-up (also packs weights, sizes the workspace) + head calibration ----
    out = pipe.step(pixel_batches[0], index)
    if not dry and args.weights == "default":
        with torch.no_grad():
            # centre the synthetic head on the mean embedding and spread its logits (sigma = 4): panoramas then fall into many
            # different geocells with top-1 probabilities 0.05 .. 0.9 (tests/golden/pipeline24 uses the same construction);
            # identical on every rank (rank 0's statistics are broadcast with the control-plane group)
            pe = out["embedding"][rank * args.panoramas:(rank + 1) * args.panoramas].mean(dim=1)
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
    if not dry:
        # one-off, per set of weights: what the 16-bit path's embedding error IS on this model -- one whole batch through the fast and
        # the exact encoder: the systematic part of their difference and the RMS of the rest (pigeon_amd/certainty.py); frozen
        # afterwards.  (A whole batch, not 32 panoramas: every fast-path launch of this process then has the step's shape, so that
        # rocprofv3's per-kernel averages of this command are the step's.)
        # (Its own batch, from a seed every rank shares: the systematic part it measures is SUBTRACTED from every fast embedding --
        # pigeon_amd/certainty.py `debias` -- so the replicas of a data-parallel job must measure the same vector, and none of the
        # timed batches is in the sample it was fitted on.)
        try:
            cal_px = torch.randn((args.panoramas, 12, 336, 336), generator=torch.Generator(device=dev).manual_seed(4321), device=dev)
            model.calibrate_certainty(cal_px, max_samples=args.panoramas)
            del cal_px
        except Exception as e:  # noqa  (nothing depends on it but the size of the re-encoded set: the threshold stays at the contract's 1e-3)
            print(f"[bench] certainty calibration failed: {e!r}", file=sys.stderr)
    for i in range(max(args.warmup, 1)):
        pipe.submit(pixel_batches[i % nb], index)
    pipe.flush()                                                   # the timed region starts with an empty queue
    sync()
    if not dry:
        enc = base._encoder(dev)
        enc.profile_reset()
        if args.profile == "all":
            enc.profile_enable(True)
        elif args.profile == "dominant":
            enc.profile_enable(True, classes=["gemm_fc1"])       # the dominant class of this workload (checked below)
        elif args.profile == "none":
            enc.graph(False)                                     # A/B arm: eager launches, no events
        g0 = enc.graph()
    outs_by_batch = {}
    certain_by_batch = {}
    info_by_step = []
    flushes_before = len(pipe.engine.flush_log)

    def take(done):
        # references only (device tensors); read after the timed region
        for r in done:
            outs_by_batch[r["meta"]] = r
            certain_by_batch[r["meta"]] = (r["certain"][rank * args.panoramas:(rank + 1) * args.panoramas], None, None)
            info_by_step.append(dict(certain=r["certain"][rank * args.panoramas:(rank + 1) * args.panoramas],
                                     cause=r["cause"][rank * args.panoramas:(rank + 1) * args.panoramas],
                                     queued=r["queued"], step=r["step"]))

    # five time stamps per step on the launch stream (host clock in --dry-run): compute vs gather(-wait), per rank
    pipe.split_marks = []
    # ---- timed region: exactly K steps between barrier + synchronize on both sides.  A step is pigeon_amd.deferred's `submit`: no
    # host synchronisation; the panoramas the 16-bit encoder cannot settle wait on the device for ONE exact pass per ~min_flush of
    # them; `flush()` -- inside the timed region -- settles what is still queued after the last step, so that all K steps' outputs
    # are final (and the reference's) when the clock stops. ----
    comm.barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        take(pipe.submit(pixel_batches[i % nb], index, meta=i % nb))
    take(pipe.flush())
    sync()
    comm.barrier()
    dt = time.perf_counter() - t0
    out = outs_by_batch[(args.steps - 1) % nb]
    if len(info_by_step) != args.steps:
        raise SystemExit(f"rank {rank}: {len(info_by_step)} of {args.steps} steps were handed out")
    flush_log = pipe.engine.flush_log[flushes_before:]
    prof = {}
    ungraphed = None
    if not dry:
        g1 = enc.graph()
        enc.profile_enable(False)
        prof = enc.profile_read()
        if args.profile != "all":
            # per-kernel timing: --profile-steps extra steps right after the timed region, every encoder launch bracketed with HIP
            # events on the launch stream (which also switches the graph replay off for them); not part of `value`
            live = {k: v for k, v in prof.items() if v[0]}
            enc.profile_reset()
            enc.profile_enable(True)
            ksteps = max(1, args.profile_steps)
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for i in range(ksteps):
                pipe.step(pixel_batches[(args.steps + i) % nb], index)
            torch.cuda.synchronize()
            tp = (time.perf_counter() - tp) / ksteps
            enc.profile_enable(False)
            prof = enc.profile_read()
            prof.update(live)                                     # classes measured inside the timed region win
            ungraphed = {"ms_per_step": tp * 1e3, "value": args.panoramas * 4 / tp, "steps": ksteps,
                         "what": "the same step with the encoder launched kernel by kernel and every launch bracketed with HIP events "
                                 "(this rank, right after the timed region): where `kernels` / `roofline` are measured"}
        enc.graph(True)
    rank_ms = [x / args.steps * 1e3 for x in comm.all_values(dt)]
    # per rank: mean compute ms (encoder + head + certainty + exact re-encode + refinement) and mean gather ms (both grouped
    # all-gathers INCLUDING the wait for the slowest rank) per step: a rank that re-encodes more than the others shows up as compute
    # there and as gather-wait everywhere else
    sp = [PanoramaPipeline.split_ms(m) for m in (pipe.split_marks or []) if len(m) == 5]
    pipe.split_marks = None
    my_compute = float(np.mean([c for c, _ in sp])) if sp else 0.0
    my_gather = float(np.mean([g for _, g in sp])) if sp else 0.0
    rank_compute, rank_gather = comm.all_values(my_compute), comm.all_values(my_gather)
    my_re = float(np.mean([inf["queued"][rank] for inf in info_by_step])) if info_by_step else 0.0
    rank_reenc = comm.all_values(my_re)
    # the exact passes of the timed region: every rank takes them in the same steps on the same number of slots
    pass_ms = pipe.engine.exact_pass_ms(flush_log)
    rank_exact = comm.all_values(float(np.sum(pass_ms)) / max(args.steps, 1))
    flush_sched = [dict(at_step=f["at_step"], queued_per_rank=f["queued"], slots_run=f["slots_run"],
                        ms=(round(pass_ms[j], 3) if j < len(pass_ms) else None)) for j, f in enumerate(flush_log)]
    all_sched = [None] * world
    if world > 1:
        torch.distributed.all_gather_object(all_sched, [(f["at_step"], f["slots_run"]) for f in flush_log])
    else:
        all_sched = [[(f["at_step"], f["slots_run"]) for f in flush_log]]
    dt = comm.max_over_ranks(dt)

    # ---- what every rank (rank 0 in particular) holds after the last step: the whole batch, restorable to sample order ----
    B, n_all = args.panoramas, args.panoramas * world
    gathered = {"panoramas": int(out["index"].numel()), "ranks": world}
    ordered_idx, ordered_cells = distributed.restore_order(out["index"], out["index"], out["preds_geocell"])
    complete = ordered_idx.tolist() == list(range(n_all))
    own = slice(rank * B, (rank + 1) * B)
    if refiner is not None:
        ordered_ref, = distributed.restore_order(out["index"], out["refined_geocell"])
        complete = complete and ordered_ref.numel() == n_all and out["refined_LLH"].shape == (n_all, 2)
        gathered["refined_shape"] = list(out["refined_LLH"].shape)
    gathered["complete_and_in_sample_order"] = bool(complete)
    distinct_cells = int(torch.unique(out["preds_geocell"][own]).numel())
    if not complete:
        raise SystemExit(f"rank {rank}: gathered batch is incomplete: {gathered}")

    # All collectives are behind us.  Every rank tears its RCCL communicator and the control-plane group down HERE, together,
    # right after a barrier (ncclCommDestroy wants all ranks of a node; a gloo group alive at interpreter exit aborts the
    # process); rank 0 then goes on alone with the roofline / CPU-baseline / parity legs.
    rccl_ranks = comm.rccl_ranks()
    gather_us = None
    if not dry and rccl_ranks:
        # what the step's two grouped all-gathers cost on this rank (RCCL group launch + copies), on the launch stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        own_rows = [out[k][own].contiguous() for k in ("embedding", "topk_indices", "topk_values", "preds_LLH", "index")]
        small = [out["refined_LLH"][own].contiguous(), out["refined_geocell"][own].contiguous()] if refiner is not None else None
        for it in range(22):
            if it == 2:
                e0.record()
            comm.gather_many(own_rows)
            if small is not None:
                comm.gather_many(small)
        e1.record()
        torch.cuda.synchronize()
        gather_us = e0.elapsed_time(e1) * 1e3 / 20
    comm.barrier()
    comm.close()
    if rank != 0:
        return
    images_per_step = args.panoramas * 4 * world
    value = images_per_step * args.steps / dt
    step_ms = dt / args.steps * 1e3
    result = {
        "metric": "images/sec end-to-end (ViT+head+refine), 4x336x336",
        "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": step_ms, "per_rank_ms_per_step": [round(x, 3) for x in rank_ms],
        "per_rank_split_ms": {"compute": [round(x, 3) for x in rank_compute], "gather_incl_wait": [round(x, 3) for x in rank_gather],
                              "exact_passes_per_step": [round(x, 3) for x in rank_exact],
                              "queued_panoramas_per_step": [round(x, 2) for x in rank_reenc],
                              "what": "stream time stamps around the two grouped all-gathers of every timed step: compute = encoder + head + "
                                      "certainty + refinement + queueing of the uncertain rows; gather = both collectives including the wait "
                                      "for the slowest rank; exact_passes = the exact tier's passes (one per min_flush queued panoramas, a whole number of pass quanta each, "
                                      "same steps and same slot count on every rank), averaged over the steps"},
        "exact_pass_schedule": {"this_rank": flush_sched, "same_on_every_rank": bool(all(sc == all_sched[0] for sc in all_sched)),
                                "min_flush": pipe.engine.min_flush, "pass_quantum": pipe.engine.pass_quantum, "max_lag": args.max_lag},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32-stub" if dry else enc.mma_dtype, "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: SuperGuessr 4-panorama ViT-L/14-336 + 10k-geocell head + ProtoRefiner "
                               "top-5 over 1Mx1024 bank" + (" (configs[4] shape: sharded over GPUs, all-gather before refinement)" if world > 1 else ""),
                   "panoramas_per_gpu": args.panoramas, "images_per_step": images_per_step, "layers": args.layers,
                   "geocells": args.cells, "prototypes": args.cells * args.protos_per_cell, "topk": args.topk,
                   "parallelism": f"dp{world}", "weights": ("random init seed 0 (HF CLIP init distributions); head centred on the mean embedding" if args.weights == "default" else
                               "NOT the BASELINE line: synthetic.make_vit_weights_spread(seed 31) -- embeddings spread like a trained tower's -- "
                               "and nn.Linear's default head at its natural scale (parity evidence run)"),
                   "resident_pixel_batches": nb, "distinct_argmax_cells_last_step": distinct_cells,
                   "launcher": os.environ.get("PIGEON_BENCH_LAUNCHER", "torchrun" if "TORCHELASTIC_RUN_ID" in os.environ else "direct")},
        "gathered_results": gathered,
    }
    if dry:
        result.update({"dry_run": True, "roofline": None, "cpu_baseline": None,
                       "collective": {"backend": "gloo (CPU stand-in for pg_allgather_many)", "nranks": world}})
        _emit(result)
        return

    from pigeon_amd import _lib
    kernels = {}
    prof_steps = args.steps if args.profile == "all" else max(1, args.profile_steps)
    for name, (cnt, ms) in prof.items():
        if cnt:
            in_region = args.profile == "all" or (args.profile == "dominant" and name == "gemm_fc1")
            kernels[name] = {"launches": cnt, "avg_ms": ms / cnt, "launches_per_step": cnt / (args.steps if in_region else prof_steps)}
    # per-launch rows: the encoder processes <= max_chunk (512) images per internal pass, so a 512-image step launches
    # every layer kernel once with M = 512*577 rows
    chunk_rows = min(args.panoramas * 4, enc.max_chunk) * 577
    for name in GEMM_FLOPS:
        if name in kernels:
            kernels[name]["tflops"] = GEMM_FLOPS[name] * chunk_rows / (kernels[name]["avg_ms"] * 1e-3) / 1e12
    if "attention" in kernels:
        kernels["attention"]["tflops"] = 4.0 * 577 * 577 * 64 * 16 * (chunk_rows / 577) / (kernels["attention"]["avg_ms"] * 1e-3) / 1e12
    dom = max((k for k in kernels if k in GEMM_FLOPS), key=lambda k: kernels[k]["avg_ms"] * kernels[k].get("launches_per_step", 1))
    if args.profile == "dominant":
        dom = "gemm_fc1"                  # the class that was bracketed INSIDE the timed region (it is the dominant one: checked by `all`)
    achieved = kernels[dom]["tflops"]
    traffic, traffic_detail = _committed_traffic(dom, chunk_rows)
    result["mfma_frac_end_to_end"] = value * FLOP_PER_IMAGE / (world * PEAK_MFMA)
    result["graph"] = {"encoder_body_replays_in_timed_region": g1[0] - g0[0], "captures_total": g1[1],
                       "what": "pg_vit_forward replays the ~250 launches between im2col and the token mean from a hipGraph captured at the "
                               "second forward of a (workspace, n_images) key; bit-identical to the eager launches (tests/test_gpu_precise.py)"}
    if ungraphed is not None:
        result["ungraphed_evented"] = ungraphed
    cf = [c[0].float().mean().item() for c in certain_by_batch.values() if c[0] is not None]
    if cf:
        result["certain_frac"] = float(np.mean(cf))               # after the exact re-encode (product mode) / as flagged (--fast)
    if pinned:
        result["config"]["rank0_cores"] = f"{len(pinned)} cores next to GPU {local} (sched_setaffinity)"
    result["roofline"] = {"bound": "mfma", "kernel": f"gemm16 {dom}: M={chunk_rows} rows x {GEMM_FLOPS[dom]} FLOP/row per launch",
                          "achieved": achieved, "peak": PEAK_MFMA / 1e12, "unit": "TFLOP/s", "frac": achieved / (PEAK_MFMA / 1e12),
                          "traffic": traffic, "traffic_detail": traffic_detail,
                          "timing": {"all": "HIP events around every encoder launch on the launch stream, inside the timed region (in-process number)",
                                     "dominant": "HIP events around the dominant class's launches (gemm_fc1) inside the timed region; the other "
                                                 "classes of `kernels` from one extra bracketed step after it",
                                     "graph": f"HIP events around every encoder launch on the launch stream during {prof_steps} un-graphed steps run right "
                                              "after the timed region (same process, same resident inputs); the timed region itself replays the "
                                              "encoder body from its hipGraph and carries no events",
                                     "none": "no events inside the timed region (A/B arm); `kernels` from extra bracketed steps after it"}[args.profile]}
    if traffic_detail and traffic_detail.get("rocprof_avg_ms"):
        # the committed rocprofv3 --kernel-trace --stats average of the same kernel (another box of the pool: +-4 %)
        # (persistent kernel + its small-tile tail launch: one GEMM of the model is both)
        t_roc = traffic_detail["rocprof_avg_ms"] + (traffic_detail.get("rocprof_tail_avg_ms") or 0.0)
        result["roofline"]["frac_rocprof"] = GEMM_FLOPS[dom] * chunk_rows / (t_roc * 1e-3) / PEAK_MFMA
    result["kernels"] = kernels
    result["fp16_range_alarm_rows"] = enc.range_alarm_read()       # always-on: residual rows that came near the fp16 limit (0 = none)
    if rccl_ranks:
        result["rccl"] = {"nranks": rccl_ranks, "version": _lib.load().pg_comm_rccl_version(),
                          "collective": "pg_allgather_many (C ABI, csrc/comm.hip): 5 buffers before refinement + 2 after, one grouped launch each",
                          "inside_timed_region": True, "both_gathers_us_per_step": gather_us,
                          "forced_at_one_rank": bool(world == 1)}
    if rccl_error is not None:
        result["rccl"] = {"error": rccl_error, "forced_at_one_rank": False,
                          "note": "the 1-rank RCCL communicator could not be created; the step ran with the identity gather"}
    if refiner is not None:
        # The refinement against the HBM roofline.  Timed here, back to back on the launch stream (the host runs ahead), over the
        # candidate sets of the resident pixel batches in turn (different cells every launch: ~1 GB of distinct bank rows, nothing
        # comes from the 256 MB Infinity Cache).  Inside the step the exact mode's host synchronisation leaves the launch thread only
        # microseconds ahead of the GPU at this point, so stream events around the step's own refinement would time the launch
        # latency (measured: 0.115 ms against 0.058 ms of kernel time in profiles/r05/bench_kernel_stats.csv).
        sets = [outs_by_batch[j] for j in sorted(outs_by_batch)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        rows, n_it = [], 20
        for it in range(2 + n_it):
            o_ = sets[it % len(sets)]
            if it == 2:
                e0.record()
            refiner(o_["embedding"][own], initial_preds=o_["preds_LLH"][own], candidate_cells=o_["topk_indices"][own],
                    candidate_probs=o_["topk_values"][own], quiet=True)
            if it >= 2:
                rows.append(refiner.last_scratch)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / n_it
        bytes_per_launch = 4096.0 * float(np.mean([float(s_[..., 3].sum()) for s_ in rows]))
        result["roofline_refine"] = {
            "bound": "hbm", "kernel": "refine_candidates_kernel + refine_select_kernel (one refinement of the rank's 128 queries)",
            "achieved": bytes_per_launch / t / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": bytes_per_launch / t / PEAK_HBM,
            "algorithmic_bytes_per_launch": bytes_per_launch, "avg_ms": t * 1e3,
            "traffic": _committed_traffic("refine_candidates", chunk_rows)[0],
            "timing": f"{n_it} refinements back to back on the launch stream after the timed region, the candidate sets of the {len(sets)} resident "
                      "pixel batches in turn",
            "note": "bytes = 4096 B x bank rows streamed (prototypes of the top-k cells + members of the chosen clusters), counted by the "
                    f"kernel; {distinct_cells} distinct argmax cells in the last step.  A 128-query launch (640 blocks, 2.5 per CU) is over "
                    "before it reaches a steady state: the same kernels stream 0.70 of the HBM peak at 1024 queries and 0.75 at 2048 "
                    "(profiles/r05/refine_bandwidth_vs_batch.txt).  `traffic` (2 x FETCH_SIZE of the committed PMC pass) is below the "
                    "algorithmic bytes: that pass runs few steps on few pixel batches, so part of what it streams is still in the "
                    "256 MB Infinity Cache and never reaches the fabric counters"}
    pipe.refine_events = None
    enc.profile_reset()

    if not args.no_extras and world == 1:
        # ---- ingest: uint8 host images -> side-stream H2D -> GPU preprocessing -> step (overlapped), and round 2's fp32 leg ----
        try:
            result["h2d_inclusive"] = ingest_leg(args, dev, pipe, index, step_ms)
        except Exception as e:  # noqa
            result["h2d_inclusive"] = {"error": repr(e)}
        try:
            host = torch.empty((args.panoramas, 12, 336, 336), dtype=torch.float32).pin_memory()
            host.copy_(pixel_batches[0])
            stage = torch.empty_like(pixel_batches[0])

            def h2d_step():
                stage.copy_(host, non_blocking=True)
                pipe.step(stage, index)
            t = _time_gpu(h2d_step, max(2, min(args.steps, 3)), 1)
            result.setdefault("h2d_inclusive", {})["fp32_no_overlap"] = {
                "value": args.panoramas * 4 / t, "unit": "images/s", "ms_per_step": t * 1e3,
                "what": "round 2's leg: pinned host fp32 pixels (694 MB per 128 panoramas) copied H2D on the compute stream, then the step"}
            del host, stage
        except Exception as e:  # noqa
            result.setdefault("h2d_inclusive", {})["fp32_no_overlap"] = {"error": repr(e)}
        # ---- BASELINE configs[0] / configs[1] / configs[2] ----
        oc = []
        try:
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
        if refiner is not None and args.weights == "default":
            try:
                oc.append(spread_tower_leg(args, dev, bank_t, pixel_batches, index))
            except Exception as e:  # noqa
                import traceback
                oc.append({"workload": "spread tower", "error": repr(e), "trace": traceback.format_exc()[-600:]})
        result["other_configs"] = oc

    # ---- the two modes next to each other.  The timed region ran the PRODUCT configuration: exact_top1 (every discrete output --
    # geocell argmax, refined cell and point -- is the fp32 reference's; samples inside the 16-bit path's error band are re-encoded by
    # pg_vit_forward_precise), unless --fast.  The other mode runs here for --fast-steps steps on the same resident batches. ----
    other_outs = {}
    cpu_sample_emb, cpu_per = None, 0
    headline_exact = bool(model.exact_top1)
    n_re = [int(inf["queued"][0]) for inf in info_by_step]
    cert_all = [inf["certain"] for inf in info_by_step]
    result["certainty"] = {
        "mode": "exact_top1 (product default)" if headline_exact else "fast (--fast): certainty reported, nothing re-encoded",
        "rule": model.certainty.describe(), "calibration": model.certainty.stats,
        "reencoded_panoramas_per_step": n_re, "panoramas_per_step": args.panoramas,
        "reencoded_share": float(np.sum(n_re)) / max(1, args.panoramas * len(n_re)),
        "uncertain_after_step": [int((~c).sum()) for c in cert_all],
        "uncertain_by_cause": _cause_counts([inf["cause"] for inf in info_by_step]),
        "boundary_checked": bool(pipe.engine.boundary_checked),
        "rows_that_did_not_fit_the_queue": pipe.engine.check_nothing_dropped()}
    if world == 1 and args.fast_steps > 0:
        try:
            model.exact_top1 = not headline_exact
            if model.exact_top1:
                base.enable_precise(True)
            pipe.refine_events = None
            pipe.step(pixel_batches[0], index)                    # warm-up of the other mode (workspaces, weight copies)
            torch.cuda.synchronize()
            n_re_o = []

            def take_o(done):
                for r in done:
                    other_outs[r["meta"]] = r
                    n_re_o.append(int(r["queued"][0]))
            te = time.perf_counter()
            for i in range(args.fast_steps):
                take_o(pipe.submit(pixel_batches[i % nb], index, meta=i % nb))
            take_o(pipe.flush())
            torch.cuda.synchronize()
            te = (time.perf_counter() - te) / args.fast_steps
            n_timed = len(n_re_o)
            for i in range(nb):                                    # the parity legs below want every resident batch (untimed)
                if i not in other_outs:
                    take_o(pipe.submit(pixel_batches[i], index, meta=i))
            take_o(pipe.flush())
            torch.cuda.synchronize()
            leg = {"value": args.panoramas * 4 / te, "unit": "images/s", "ms_per_step": te * 1e3, "steps": args.fast_steps,
                   "mfma_frac_end_to_end": args.panoramas * 4 / te * FLOP_PER_IMAGE / PEAK_MFMA,
                   "reencoded_panoramas_per_step": n_re_o[:n_timed]}
            if headline_exact:
                leg["what"] = ("SuperGuessr(exact_top1=False) / PIGEON_EXACT_TOP1=0: the 16-bit path alone -- embeddings within 1e-3, discrete "
                               "outputs NOT guaranteed (see the parity legs' `fast_mode` entries); timed after the timed region")
                result["fast_mode"] = leg
                result["exact_cost_vs_fast"] = step_ms / (te * 1e3)
                if isinstance(result.get("h2d_inclusive"), dict) and "ms_per_step" in result["h2d_inclusive"]:
                    result["h2d_inclusive"]["frac_of_fast_mode_resident"] = te * 1e3 / result["h2d_inclusive"]["ms_per_step"]
            else:
                leg["what"] = "SuperGuessr(exact_top1=True), the product default, timed after the timed region (the headline of this run is --fast)"
                result["exact_mode"] = leg
                result["exact_cost_vs_fast"] = te * 1e3 / step_ms
            model.exact_top1 = headline_exact
        except Exception as e:  # noqa
            import traceback
            result["fast_mode" if headline_exact else "exact_mode"] = {"error": repr(e), "trace": traceback.format_exc()[-600:]}
            model.exact_top1 = headline_exact
    other_name = "fast_mode" if headline_exact else "exact_mode"

    if world == 1 and args.cpu_images > 0:
        try:
            # sample: the first panoramas of EVERY resident pixel batch the timed region used, in turn
            used = sorted(outs_by_batch)
            per = max(1, args.cpu_images // 4 // len(used))
            px = torch.cat([pixel_batches[j][:per] for j in used]).cpu()

            def take(outs):
                h = {"embedding": torch.cat([outs[j]["embedding"][:per] for j in used]),
                     "preds_geocell": torch.cat([outs[j]["preds_geocell"][:per] for j in used]),
                     "where": [f"batch {j} #{i}" for j in used for i in range(per)]}
                if refiner is not None:
                    h["refined_geocell"] = torch.cat([outs[j]["refined_geocell"][:per] for j in used])
                    h["refined_LLH"] = torch.cat([outs[j]["refined_LLH"][:per] for j in used])
                return h
            hip = take(outs_by_batch)
            hip["certain"] = torch.cat([certain_by_batch[j][0][:per] for j in used]).cpu()
            cb, o, refined = cpu_baseline(args, model, bank_t if refiner is not None else None, px)
            result["cpu_baseline"] = cb
            rep = parity_report(args, dev, model, o, refined, hip)
            cert = hip["certain"]
            rep["certain"] = f"{int(cert.sum())}/{cert.numel()}"
            rep["flips_among_certain"] = int(sum(1 for r in rep["flipped"] if bool(cert[hip["where"].index(r["panorama"])])))
            if other_outs and all(j in other_outs for j in used):
                rx = parity_report(args, dev, model, o, refined, take(other_outs))
                rep[other_name] = {k: rx[k] for k in ("embedding_rel_err", "logit_abs_err_max", "flips", "flipped", "geocell_argmax_equal",
                                                      "refined_mismatch_unconditional", "refined_cell_equal_where_argmax_equal",
                                                      "refined_lnglat_equal_where_argmax_equal") if k in rx}
            result["parity_vs_oracle_sample"] = rep
            cpu_sample_emb, cpu_per = o["embedding"], per
        except Exception as e:  # noqa
            import traceback
            result["cpu_baseline"] = {"error": str(e), "trace": traceback.format_exc()[-800:]}
    if world == 1 and not args.no_extras:
        try:
            used = sorted(outs_by_batch)
            result["parity_vs_reference_module_gpu_fp32"] = gpu_module_parity(
                args, dev, vit_sd, model, pixel_batches, used, outs_by_batch, certain_by_batch, other_outs, other_name, cpu_sample_emb, cpu_per,
                bank_t if refiner is not None else None)
        except Exception as e:  # noqa
            import traceback
            result["parity_vs_reference_module_gpu_fp32"] = {"error": repr(e), "trace": traceback.format_exc()[-600:]}
    if world == 1 and not args.no_extras:
        del pixel_batches
        torch.cuda.empty_cache()
        result["secondary_baseline"] = secondary_baseline(dev, vit_sd, args.layers)
    _emit(result)


_PROTOCOL_OUT = None


def _claim_stdout():
    """stdout of this command is a protocol -- ONE JSON line, from rank 0.  Everything else that anything in the process writes to
    file descriptor 1 (the product's status prints -- the reference's classes print, they do not log --, RCCL's banner, a library's
    warning) goes to stderr from here on; `_emit` writes the line to the descriptor stdout had."""
    global _PROTOCOL_OUT
    if _PROTOCOL_OUT is not None:
        return
    sys.stdout.flush()
    _PROTOCOL_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = sys.stderr


def _emit(result):
    out = _PROTOCOL_OUT if _PROTOCOL_OUT is not None else sys.stdout
    out.write(json.dumps(result) + "\n")
    out.flush()


def main():
    args = parse()
    if args.cpu_worker:
        cpu_worker_main(args.cpu_worker)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    _claim_stdout()
    worker(args)


if __name__ == "__main__":
    main()
