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
    mk = lambda r: [                                                     # the five buffers PanoramaPipeline.step gathers
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
