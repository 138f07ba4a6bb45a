"""bench.py's multi-rank control flow on the CPU (no GPU here): `--dry-run` swaps the HIP model / refiner for stubs and RCCL for
gloo, everything else -- self-launch of the N ranks, interleaved sample ids, the two grouped all-gathers of
PanoramaPipeline.submit / flush (pigeon_amd.deferred), restore_order on rank 0, barrier-bracketed timing with max over ranks, ONE JSON line -- is the code the
GPU run executes (reference launch: accelerate owns process creation, preprocessing/embed.py:55-56,68; collection of all
ranks' results: training/train_eval_loop.py:98-112)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _check(stdout, n, steps, warmup, launcher):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout                                  # exactly ONE JSON line, from rank 0
    r = json.loads(lines[0])
    for k in CONTRACT:
        assert k in r, k
    assert r["n_gpus"] == n and r["steps"] == steps and r["warmup"] == warmup and r["dry_run"] is True
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["unit"] == "images/s" and r["vs_baseline"] is None
    assert r["config"]["images_per_step"] == 6 * 4 * n and r["config"]["parallelism"] == f"dp{n}"
    assert r["config"]["launcher"] == launcher
    assert abs(r["value"] - r["config"]["images_per_step"] / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]
    g = r["gathered_results"]                                       # rank 0 holds the refined output of ALL ranks, in sample order
    assert g == {"panoramas": 6 * n, "ranks": n, "refined_shape": [6 * n, 2], "complete_and_in_sample_order": True}
    assert r["collective"]["nranks"] == n
    return r


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_bench_gpus2_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment: the shape of the driver's 1-GPU command."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--panoramas", "6",
                        "--cells", "50", "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=300, env=_env())
    assert p.returncode == 0, p.stderr[-2000:]
    _check(p.stdout, 2, 3, 1, "self")


def test_bench_under_torchrun():
    """The driver's N>1 command line: python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2."""
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run",
                        "--panoramas", "6", "--cells", "50", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert p.returncode == 0, p.stderr[-2000:]
    _check(p.stdout, 2, 2, 1, "torchrun")


def test_bench_world8_equal_work_under_unequal_uncertain_counts():
    """The 8-rank launch the driver will make on an 8-GPU node, on the CPU: 8 ranks over gloo, the deferred exact tier's protocol with
    UNEQUAL uncertain counts per rank (the stub model of rank r finds r panoramas uncertain per step; an exact pass costs 2 ms per
    SLOT run).  Asserted: one JSON line, every rank's results on rank 0 in sample order, every rank queued what it found -- and every
    rank ran the exact tier in the SAME steps on the SAME number of slots (the longest queue), so that the exact tier's time per rank
    is equal although rank 0 had nothing to re-encode and rank 7 seven panoramas per step."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--panoramas", "12",
                        "--cells", "50", "--steps", "6", "--warmup", "1", "--min-flush", "7", "--pass-quantum", "7", "--pixel-batches", "4"],
                       capture_output=True, text=True, timeout=600, env=_env())
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["config"]["images_per_step"] == 12 * 4 * 8 and r["config"]["parallelism"] == "dp8"
    assert r["gathered_results"] == {"panoramas": 96, "ranks": 8, "refined_shape": [96, 2], "complete_and_in_sample_order": True}
    sp = r["per_rank_split_ms"]
    assert sp["queued_panoramas_per_step"] == [float(i) for i in range(8)]              # unequal uncertain counts ...
    assert len(sp["compute"]) == 8 and len(sp["gather_incl_wait"]) == 8 and len(r["per_rank_ms_per_step"]) == 8
    sched = r["exact_pass_schedule"]
    assert sched["same_on_every_rank"] is True and len(sched["this_rank"]) >= 2          # ... same flush steps, same slots everywhere
    assert sched["min_flush"] == 7 and sched["pass_quantum"] == 7                        # what the GPU defaults to on 256 CUs
    for f in sched["this_rank"]:
        assert f["slots_run"] == max(f["queued_per_rank"]) and f["queued_per_rank"][0] == 0 and f["queued_per_rank"][7] > 0
    assert all(f["slots_run"] % 7 == 0 for f in sched["this_rank"][:-1])                 # whole pass quanta from the queue heads
    # 6 steps x 7 panoramas on the longest queue = 42 slots x 2 ms = 84 ms of exact tier on EVERY rank (14 ms per step)
    ex = sp["exact_passes_per_step"]
    assert sum(f["slots_run"] for f in sched["this_rank"]) == 42
    assert min(ex) > 0.8 * 14.0 and max(ex) < min(ex) * 1.25 + 2.0, ex                   # equal work (slack: a loaded host, gloo)
    # the fast mode of the same launch: nobody queues anything, no exact pass
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--fast", "--panoramas", "12",
                        "--cells", "50", "--steps", "3", "--warmup", "1", "--pixel-batches", "4"], capture_output=True, text=True,
                       timeout=600, env=_env())
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert r["per_rank_split_ms"]["queued_panoramas_per_step"] == [0.0] * 8 and r["exact_pass_schedule"]["this_rank"] == []


def test_bench_single_rank_dry_run_and_world_mismatch():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--panoramas", "6", "--cells", "50",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=300, env=_env())
    assert p.returncode == 0, p.stderr[-2000:]
    _check(p.stdout, 1, 2, 1, "direct")
    # stdout is the protocol: the JSON line and nothing else, whatever the product or a library prints on the way (file descriptor 1
    # is pointed at stderr for the run; PIGEON_BENCH_TEST_NOISE makes the worker print through Python AND through the descriptor)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--panoramas", "6", "--cells", "50",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=300,
                       env=dict(_env(), PIGEON_BENCH_TEST_NOISE="1"))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{") and json.loads(lines[0])["n_gpus"] == 1, p.stdout[:400]
    assert "noise through print" in p.stderr and "noise through fd 1" in p.stderr
    env = dict(_env(), WORLD_SIZE="1", RANK="0")                    # a launcher that disagrees with --gpus: loud, not silent
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True,
                       text=True, timeout=300, env=env)
    assert p.returncode != 0 and "--gpus 2 but WORLD_SIZE is 1" in p.stderr


def test_bench_self_launch_propagates_a_rank_failure():
    """A rank that dies (here: more ranks than the stub allows via an impossible flag) must fail the launcher, not hang it."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--panoramas", "6", "--cells", "3",
                        "--topk", "5"], capture_output=True, text=True, timeout=300, env=_env())   # topk > cells: torch.topk raises
    assert p.returncode != 0


def test_flip_analysis_rule():
    """The rule bench.py's parity leg applies (pure function, no GPU): a top-1 flip is explained only below 2 x the measured
    logit error."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    ref = torch.zeros((3, 6))
    ref[0, 2], ref[0, 4] = 5.0, 4.99          # panorama 0: near-tie, margin 0.01
    ref[1, 1], ref[1, 3] = 5.0, 3.0           # panorama 1: clear winner
    ref[2, 0], ref[2, 5] = 5.0, 4.0           # panorama 2: margin 1.0
    hip = ref.clone()
    hip[0, 4] += 0.02                          # flips the near-tie: |delta| = 0.02 -> bound 0.04 > margin 0.01: explained
    hip[1, 1] -= 0.015
    r, flips = bench.flip_analysis(ref, hip, torch.tensor([4, 1, 0]), ["a", "b", "c"])
    assert r["flips"] == 1 and r["flips_unexplained"] == 0 and flips.tolist() == [True, False, False]
    assert abs(r["logit_abs_err_max"] - 0.02) < 1e-6 and r["flipped"][0]["panorama"] == "a" and abs(r["flipped"][0]["d_top2"] - 0.02) < 1e-6
    assert r["geocell_argmax_equal"] is False and r["smallest_margins"][0]["panorama"] == "a"
    # the same error but a flip where the reference margin is 1.0: NOT explained (a bug, not a near-tie)
    r2, _ = bench.flip_analysis(ref, hip, torch.tensor([4, 1, 5]), ["a", "b", "c"])
    assert r2["flips"] == 2 and r2["flips_unexplained"] == 1
    r3, _ = bench.flip_analysis(ref, ref, torch.tensor([2, 1, 0]), ["a", "b", "c"])
    assert r3["flips"] == 0 and r3["geocell_argmax_equal"] is True and r3["logit_abs_err_max"] == 0.0


def test_cpu_baseline_pool_and_quota(tmp_path):
    """bench.py's CPU-baseline leg without a GPU: the pool of worker processes (disjoint cores and image shards, a common go
    signal, wall time up to the last worker's last image) returns the reference module's embeddings in image order and agrees
    with the oracle restatement run the same way; the pool is sized to what the container may USE (cgroup quota), not to what it
    sees."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from oracle import pigeon_oracle as orc
    from pigeon_amd import synthetic
    q = bench.cpu_quota_cores()
    assert 0 < q <= len(os.sched_getaffinity(0))
    px = synthetic.make_pixels(5, seed=3)
    emb, wall, per, busy = bench.cpu_pool("module", px, 1, 2, 2, weight_seed=0)
    assert emb.shape == (5, 1024) and len(per) == 2 and busy == 4 and 0 < max(per) <= wall + 0.5
    sd = synthetic.make_vit_weights(seed=0, layers=1)
    assert orc.rel_err(emb, orc.clip_embedding(sd, px)) < 1e-5            # image order kept across the shards (3 + 2)
    emb_p, _, _, _ = bench.cpu_pool("port", px, 1, 5, 1, weight_seed=0)    # one image per worker
    assert orc.rel_err(emb_p, emb) < 1e-5
