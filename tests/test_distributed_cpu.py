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
    comm = distributed.init_from_env("gloo")
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
        emb = np.concatenate(list(np.load(os.path.join(outdir, "train.npy"), allow_pickle=True)), axis=0)
        idx = np.concatenate(list(np.load(os.path.join(outdir, "train_indices.npy"), allow_pickle=True)), axis=0)
        e, = distributed.restore_order(torch.from_numpy(idx.astype(np.int64)), torch.from_numpy(emb.astype(np.float32)))
        assert e.shape == (20, 5)
        assert torch.equal(e[:, 0], torch.arange(20, dtype=torch.float32) * 2.0)
    torch.distributed.destroy_process_group()


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
