"""The N>1 path on CPU: two processes, gloo backend.  Checks the Communicator's accelerate-style gather (rank-major),
the single-collective gather_many, batch sharding + index-based order restoration (reference
preprocessing/embed.py:36-37 and dataset_preprocessing.py:296-300), and compute_embeddings' on-disk format."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    from pigeon_amd import distributed
    from pigeon_amd.embed import compute_embeddings
    comm = distributed.init_from_env("gloo", set_device=False)     # host-only, also when the box has a GPU
    assert (comm.rank, comm.world_size) == (rank, world)
    # --- gather: rank-major concat, identical on every rank
    t = torch.arange(6, dtype=torch.float32).view(3, 2) + 100 * rank
    g = comm.gather(t)
    assert g.shape == (3 * world, 2)
    for r in range(world):
        assert torch.equal(g[3 * r:3 * r + 3], torch.arange(6, dtype=torch.float32).view(3, 2) + 100 * r)
    # --- gather_many: mixed dtypes / shapes in ONE collective
    a = torch.full((4, 4, 8), float(rank)); b = torch.arange(4, dtype=torch.int64) + 10 * rank
    c = torch.full((4, 2), rank + 0.5, dtype=torch.float64)
    ga, gb, gc = comm.gather_many([a, b, c])
    assert ga.shape == (8, 4, 8) and gb.dtype == torch.int64 and gc.dtype == torch.float64
    assert torch.equal(gb, torch.cat([torch.arange(4) + 10 * r for r in range(world)]))
    assert torch.equal(gc[:, 0], torch.tensor([0.5] * 4 + [1.5] * 4, dtype=torch.float64))
    assert bool((ga[4:] == 1).all()) and bool((ga[:4] == 0).all())
    # --- sharded embedding loop with a stand-in model: 7 batches of 3 samples (odd count -> wrap-around padding)
    n, bs = 20, 3
    data = [(torch.arange(i, min(i + bs, n), dtype=torch.float32)[:, None].repeat(1, 5), torch.arange(i, min(i + bs, n)))
            for i in range(0, n, bs)]
    data = [d for d in data if d[0].shape[0] == bs]                     # drop ragged last batch (18 samples, 6 batches)
    data.append((torch.arange(18, 20, dtype=torch.float32)[:, None].repeat(1, 5).repeat(2, 1)[:bs], torch.tensor([18, 19, 18])))
    shard = list(distributed.shard_batches(data, rank, world))
    assert len(shard) == 4                                               # 7 batches -> 4 steps per rank (last wraps)
    outs, idxs = compute_embeddings("train", lambda px: px * 2.0, shard, comm, out_dir=outdir)
    comm.barrier()
    if rank == 0:
        emb = np.concatenate(list(np.load(os.path.join(outdir, "train.npy"))), axis=0)
        idx = np.concatenate(list(np.load(os.path.join(outdir, "train_indices.npy"))), axis=0)
        e, = distributed.restore_order(torch.from_numpy(idx.astype(np.int64)), torch.from_numpy(emb.astype(np.float32)))
        assert e.shape == (20, 5)
        assert torch.equal(e[:, 0], torch.arange(20, dtype=torch.float32) * 2.0)
    # --- embed_images end to end with a GENUINELY ragged final batch (23 samples, batch 4, DataLoader drop_last=False):
    #     every rank must see full batches (sample-level wrap-around padding, accelerate's even_batches) or the collective
    #     would run with unequal counts
    from pigeon_amd.embed import embed_images

    class Stub(torch.nn.Module):
        def forward(self, px):
            assert px.shape[0] == 4, px.shape                           # never a short batch under world_size > 1
            return px.flatten(1)[:, :6] * 2.0

    items = [{"image": torch.full((3, 2, 2), float(i)), "index": i} for i in range(23)]
    out2 = os.path.join(outdir, "ragged")
    embed_images(Stub(), {"train": items}, comm, batch_size=4, num_workers=0, out_dir=out2)
    if rank == 0:
        emb = np.load(os.path.join(out2, "train.npy"))                     # plain numeric arrays, no pickle
        idx = np.load(os.path.join(out2, "train_indices.npy"))
        assert emb.shape == (3, 8, 6) and idx.shape == (3, 8)
        e, = distributed.restore_order(torch.from_numpy(idx.reshape(-1)), torch.from_numpy(emb.reshape(-1, 6)))
        assert e.shape == (23, 6) and torch.equal(e[:, 0], torch.arange(23, dtype=torch.float32) * 2.0)
        # the reference reader's reorder (dataset_preprocessing.py:296-300) gives the same rows
        arg = np.argsort(idx.flatten()[:])                                 # duplicates sort next to their originals
        assert np.array_equal(np.unique(idx.flatten()), np.arange(23))
    # --- the data-parallel step (pigeon_amd.deferred behind PanoramaPipeline): ViT + head on the shard -> gather -> refine own slice ->
    # gather; the rows the fast path cannot settle are queued per rank, BOTH ranks run the exact tier in the SAME steps on the SAME
    # number of slots (the longer queue), and every rank ends up with every rank's patched rows.  Scripted stand-ins: tests/_scripted.py
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _scripted import ScriptedModel, ScriptedRefiner, make_pixels
    from oracle import requeue_oracle
    from pigeon_amd.evaluate import PanoramaPipeline
    B = 3
    model, refiner = ScriptedModel(), ScriptedRefiner()
    pipe = PanoramaPipeline(model, refiner, comm, min_flush=3, max_lag=4, ops=requeue_oracle, pass_quantum=2)   # passes of 2, 4 .. slots; the rest stays queued
    index = torch.arange(B) * world + rank                                   # interleaved sample ids, as a sharded loader gives

    def pixels(r, i):
        # rank 0: one uncertain panorama in every second step; rank 1: two in every step -- unequal queues by construction
        hf = ([0, 1, 1] if i % 2 == 0 else [1, 1, 1]) if r == 0 else [0, 0, 1]
        return make_pixels(hf, head_exact=[1, 1, 1], ref_fast=[1, 1, 1 if i != 3 else 0], ref_exact=[1, 1, 1], seed=1000 * r + i)

    got = {}
    n_steps = 7
    for i in range(n_steps):
        for res in pipe.submit(pixels(rank, i), index, meta=i):
            got[res["meta"]] = res
    for res in pipe.flush():
        got[res["meta"]] = res
    assert sorted(got) == list(range(n_steps))
    log = [(f["at_step"], tuple(f["queued"]), f["slots_run"]) for f in pipe.engine.flush_log]
    logs = [None] * world
    torch.distributed.all_gather_object(logs, log)
    assert logs[0] == logs[1] and len(log) >= 2                               # same steps, same queue view, same slots on both ranks
    assert all(slots == max(q) for _, q, slots in log) and any(q[0] != q[1] for _, q, _ in log)
    assert all(slots % 2 == 0 for _, _, slots in log[:-1])                    # whole quanta, except the closing flush
    assert [n for _, n in model.calls] == [slots for _, _, slots in log]       # the exact tier ran on the PADDED size on this rank too
    for i in range(n_steps):
        res = got[i]
        assert torch.equal(res["index"], torch.cat([torch.arange(B) * world + r for r in range(world)]))   # rank-major
        for r in range(world):
            rows = pixels(r, i).reshape(B, -1)
            fast, exact = model._embed(rows, True), model._embed(rows, False)
            hf, hx = model._head(fast), model._head(exact)
            unc = (rows[:, 0] <= 0.5) | (rows[:, 2] <= 0.5)
            sl = slice(r * B, (r + 1) * B)
            assert res["exact"][sl].tolist() == unc.tolist() and bool(res["certain"][sl].all())
            want_emb = torch.where(unc[:, None, None], exact, fast)
            assert torch.equal(res["embedding"][sl], want_emb), (rank, i, r)   # every rank holds every rank's patched rows
            want_cell = torch.where(unc, hx["preds_geocell"], hf["preds_geocell"])
            assert torch.equal(res["preds_geocell"][sl], want_cell)
            want_ref = torch.where(unc[:, None], (hx["preds_LLH"] + 2).float(), (hf["preds_LLH"] + 1).float())
            assert torch.equal(res["refined_LLH"][sl], want_ref)
            assert res["queued"][r] == int(unc.sum())
        oi, oc = distributed.restore_order(res["index"], res["index"], res["preds_geocell"])
        assert oi.tolist() == list(range(B * world))
    # `step` = submit + flush: settled before it returns
    res = pipe.step(pixels(rank, 0), index)
    assert res["embedding"].shape == (B * world, 4, 8) and bool(res["certain"].all())
    # a host tensor among the gathered ones is refused instead of being handed to the collective as a device pointer
    # (exercised with the meta device standing in for "another device")
    try:
        comm.gather_many([torch.zeros(2, 2), torch.zeros(2, device="meta")])
        raise AssertionError("mixed-device gather_many must raise")
    except ValueError:
        pass
    comm.close()                                  # tears down the gloo group init_from_env created (idempotent)
    comm.close()
    assert not torch.distributed.is_initialized()


def test_two_rank_gather_and_embed_loop(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)


def test_single_process_communicator_is_identity():
    sys.path.insert(0, ROOT)
    from pigeon_amd.distributed import Communicator, shard_batches
    c = Communicator()
    t = torch.randn(3, 4)
    assert c.world_size == 1 and c.gather(t) is t and c.gather_many([t])[0] is t
    assert list(shard_batches([1, 2, 3], 0, 1)) == [1, 2, 3]
    assert list(shard_batches(range(5), 1, 2)) == [1, 3, 0]             # batch 4 goes to rank 0; rank 1 wraps to batch 0
    assert list(shard_batches(range(5), 0, 2)) == [0, 2, 4]


def test_shard_batches_takes_the_batch_size_from_a_tensor_leaf():
    """dict batches whose FIRST value is a list of strings (default_collate of file names), namedtuple batches, and a
    ragged final batch: every rank gets full batches, the string leaves are padded with the same wrap-around samples."""
    import collections
    sys.path.insert(0, ROOT)
    from pigeon_amd.distributed import shard_batches, _batch_len

    def batches():
        n = 0
        for bs in (4, 4, 4, 2):
            yield {"name": [f"img{n + i}" for i in range(bs)], "x": torch.arange(n, n + bs).float()[:, None],
                   "idx": torch.arange(n, n + bs), "scale": 2.0}
            n += bs
    assert _batch_len(next(batches())) == 4
    r0, r1 = [list(shard_batches(batches(), r, 2)) for r in range(2)]
    assert [b["idx"].tolist() for b in r0] == [[0, 1, 2, 3], [8, 9, 10, 11]]
    assert [b["idx"].tolist() for b in r1] == [[4, 5, 6, 7], [12, 13, 0, 1]]
    assert r1[1]["name"] == ["img12", "img13", "img0", "img1"] and r1[1]["x"][:, 0].tolist() == [12., 13., 0., 1.]
    NT = collections.namedtuple("NT", "x idx")

    def nt():
        n = 0
        for bs in (3, 3, 1):
            yield NT(torch.arange(n, n + bs).float(), torch.arange(n, n + bs))
            n += bs
    out = [list(shard_batches(nt(), r, 2)) for r in range(2)]
    assert isinstance(out[0][1], NT) and out[0][1].idx.tolist() == [6, 0, 1] and out[1][1].idx.tolist() == [2, 3, 4]
    with pytest.raises(ValueError):
        list(shard_batches([{"a": None}, {"a": None}, {"a": None}], 0, 2))


def test_gpu_cpu_affinity_plan():
    """One process per GPU: each rank's launch thread stays on the cores next to ITS GPU; ranks whose GPUs share a NUMA node split
    that node's cores evenly; an unreadable topology means no pinning (pure function, bench.py applies it with sched_setaffinity)."""
    from pigeon_amd import distributed as d
    assert d.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    lists = ["0-7,16-23", "0-7,16-23", "8-15,24-31", None]
    assert d.gpu_cpu_affinity(lists, 0) == [0, 1, 2, 3, 4, 5, 6, 7]
    assert d.gpu_cpu_affinity(lists, 1) == [16, 17, 18, 19, 20, 21, 22, 23]
    assert d.gpu_cpu_affinity(lists, 2) == [8, 9, 10, 11, 12, 13, 14, 15, 24, 25, 26, 27, 28, 29, 30, 31]
    assert d.gpu_cpu_affinity(lists, 3) is None and d.gpu_cpu_affinity(lists, 7) is None
    all8 = ["0-63,128-191"] * 4 + ["64-127,192-255"] * 4                       # the 8-GPU, 2-socket box
    plans = [d.gpu_cpu_affinity(all8, r) for r in range(8)]
    assert all(len(p) == 32 for p in plans) and len(set(sum(plans, []))) == 256


def test_missing_rank_is_named_within_the_join_timeout(tmp_path):
    """A rank that never shows up: the others fail after PIGEON_JOIN_TIMEOUT_S with its id in the message, instead of sitting
    in torch's 30-minute rendezvous."""
    import subprocess, sys, socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               PIGEON_JOIN_TIMEOUT_S="1.5", PYTHONPATH=ROOT)
    code = "from pigeon_amd import distributed as d; d.init_from_env('gloo', set_device=False)"
    p0 = subprocess.Popen([sys.executable, "-c", code], env=env, stderr=subprocess.PIPE, text=True)
    p2 = subprocess.Popen([sys.executable, "-c", code], env=dict(env, RANK="2", LOCAL_RANK="2"), stderr=subprocess.PIPE, text=True)
    e0, e2 = p0.communicate(timeout=120)[1], p2.communicate(timeout=120)[1]
    assert p0.returncode != 0 and p2.returncode != 0
    # (the rank that gives up first removes its own file on the way out, so the later one may list it as well)
    assert "rank(s) [1] of 3 did not join" in e0 or "rank(s) [1] of 3 did not join" in e2
    for e in (e0, e2):
        assert "did not join within" in e and "1" in e.split("rank(s) [")[1].split("]")[0]
