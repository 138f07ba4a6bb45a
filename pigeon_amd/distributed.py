"""One-process-per-GPU data parallelism for the inference path.

The reference's only inference-time collective is `accelerator.gather` (preprocessing/embed.py:36-37): an
all-gather, concatenating every rank's tensor along dim 0 in rank order.  Here that is `pg_allgather(_many)` of the C
ABI -- RCCL over xGMI, csrc/comm.hip -- wrapped in a tiny Communicator with accelerate's semantics (torch.distributed
is the control plane only: bootstrap of the RCCL id, barriers, and the gloo data path of the CPU tests), plus the batch
sharding `accelerator.prepare(DataLoader)` performs (whole batches dealt round-robin to ranks, embed.py:68).
"""
from __future__ import annotations

import os
from typing import Iterable, Iterator, List, Optional

# dmabuf IPC: the host driver of these boxes supports no legacy IPC handles, and RCCL's intra-node transport (and any CUDA-tensor
# sharing across processes) fails with "hipIpcGetMemHandle: invalid argument" without it.  It has to be in the environment
# before the HIP runtime initialises, i.e. before the first torch.cuda call of the process -- hence at import time, here and in
# pigeon_amd/__init__.py, not in init_from_env.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


class Communicator:
    """rank / world_size + rank-major all-gather with accelerate's semantics.  world_size == 1 needs no process group.

    Data path: device tensors go through the C ABI (`pg_allgather` / `pg_allgather_many` in libpigeon_hip.so = RCCL over
    xGMI, csrc/comm.hip); the RCCL communicator is created lazily on first use, its unique id travelling from rank 0 over
    the torch.distributed process group, which is only the CONTROL plane here (bootstrap, barrier; "gloo" by default).
    Host tensors (the CPU tests) use that process group directly."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None, force_rccl: Optional[bool] = None):
        """force_rccl (default: env PIGEON_FORCE_RCCL=1): with ONE rank, device tensors still go through a 1-rank RCCL communicator
        (`pg_allgather_many`) instead of being returned as they are -- a single-GPU run then exercises csrc/comm.hip, the RCCL
        binding and the group launch exactly as an N-rank run does (bench.py switches it on: its N = 1 line is the early
        warning for the N > 1 ones)."""
        self.group = group
        self.force_rccl = (os.environ.get("PIGEON_FORCE_RCCL", "0") not in ("", "0")) if force_rccl is None else bool(force_rccl)
        if dist.is_available() and dist.is_initialized():
            self.rank = dist.get_rank(group)
            self.world_size = dist.get_world_size(group)
        else:
            self.rank, self.world_size = 0, 1
        self._rccl = None          # pg_comm handle (ctypes void*)
        self._rccl_device = None
        self._owns_group = False   # set by init_from_env: close() then also tears the control-plane group down

    @property
    def is_main_process(self) -> bool:
        return self.rank == 0

    is_local_main_process = is_main_process

    # ---- RCCL communicator behind the C ABI ----
    def _comm(self, device: torch.device):
        import ctypes as C
        from . import _lib
        if self._rccl is not None:
            if self._rccl_device != device.index:
                raise _lib.PigeonHipError(f"communicator was created on cuda:{self._rccl_device}, tensor is on {device}")
            return self._rccl
        lib = _lib.load()
        ident = [None]
        if self.rank == 0:
            buf = C.create_string_buffer(128)
            _lib.check(lib.pg_comm_unique_id(buf), "pg_comm_unique_id")
            ident = [buf.raw]
        if self.world_size > 1:
            dist.broadcast_object_list(ident, src=0, group=self.group)     # control plane: 128 bytes
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.pg_comm_init_rank(C.byref(h), self.world_size, ident[0], self.rank), "pg_comm_init_rank")
        self._rccl, self._rccl_device = h, device.index
        return h

    def rccl_ranks(self) -> int:
        """Number of ranks as RCCL sees them (0 before the communicator exists)."""
        import ctypes as C
        from . import _lib
        if self._rccl is None:
            return 0
        n = C.c_int()
        _lib.check(_lib.load().pg_comm_count(self._rccl, C.byref(n)), "pg_comm_count")
        return int(n.value)

    def close(self, rccl: bool = True):
        """Release the RCCL communicator and, if init_from_env created it, the control-plane process group.  Call it on EVERY
        rank before the process exits: a gloo group that is still alive when the interpreter unwinds takes the process down
        with `terminate called without an active exception` (its worker threads are still joinable) -- a rank that computed
        everything correctly then exits non-zero and fails its launcher.
        rccl=False (a rank on its way out with an exception): the RCCL communicator is abandoned, not destroyed --
        ncclCommDestroy expects the other ranks of the node, which may be sitting in a collective this rank will never join."""
        if self._rccl is not None:
            if rccl:
                from . import _lib
                _lib.load().pg_comm_destroy(self._rccl)
            self._rccl = None
        if self._owns_group and dist.is_available() and dist.is_initialized():
            self._owns_group = False
            dist.destroy_process_group()

    def gather(self, t: torch.Tensor) -> torch.Tensor:
        """accelerate.Accelerator.gather: (n, ...) on every rank -> (world*n, ...) rank-major, on every rank."""
        return self.gather_many([t])[0]

    def gather_many(self, tensors: List[torch.Tensor], out: Optional[List[torch.Tensor]] = None) -> List[torch.Tensor]:
        """Gather several tensors (same leading dimension on every rank) in ONE grouped collective; every result is
        rank-major and contiguous -- no packing, no unpacking copies.  `out`: contiguous (world * n, ...) destinations (e.g. slabs of a
        result ring) written in place of fresh tensors; with one rank and no forced RCCL they receive plain copies."""
        if out is not None:
            if len(out) != len(tensors):
                raise ValueError("gather_many: `out` must hold one destination per tensor")
            for t, o in zip(tensors, out):
                if (o.shape[0] != self.world_size * t.shape[0] or tuple(o.shape[1:]) != tuple(t.shape[1:]) or o.dtype != t.dtype
                        or o.device != t.device or not o.is_contiguous()):
                    raise ValueError(f"gather_many: destination {tuple(o.shape)} {o.dtype} on {o.device} does not fit {self.world_size} x "
                                     f"{tuple(t.shape)} {t.dtype} on {t.device} (contiguous)")
        if self.world_size == 1 and not (self.force_rccl and len(tensors) and all(t.is_cuda for t in tensors)):
            if out is None:
                return list(tensors)
            for t, o in zip(tensors, out):
                o.copy_(t)
            return list(out)
        devs = {t.device for t in tensors}
        if len(devs) != 1:
            # a host tensor (or one on another GPU) among device tensors would hand RCCL a pointer it cannot read
            raise ValueError(f"gather_many: all tensors must live on ONE device, got {sorted(str(d) for d in devs)}")
        srcs = [t.contiguous() for t in tensors]
        outs = list(out) if out is not None else [
            torch.empty((self.world_size * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in srcs]
        if srcs[0].is_cuda:
            import ctypes as C
            from . import _lib
            comm = self._comm(srcs[0].device)
            n = len(srcs)
            send = (C.c_void_p * n)(*[t.data_ptr() for t in srcs])
            recv = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
            nbytes = (C.c_size_t * n)(*[t.numel() * t.element_size() for t in srcs])
            stream = C.c_void_p(torch.cuda.current_stream(srcs[0].device).cuda_stream)
            _lib.check(_lib.load().pg_allgather_many(comm, n, send, recv, nbytes, stream), "pg_allgather_many")
        else:
            for t, o in zip(srcs, outs):                                      # gloo (CPU tests): list form
                if t.shape[0] == 0:
                    continue
                dist.all_gather(list(o.chunk(self.world_size, dim=0)), t, group=self.group)
        return outs

    def all_values(self, value: float) -> List[float]:
        """Control-plane all-gather of one host scalar per rank (bench: every rank's step time, not only the maximum)."""
        if self.world_size == 1:
            return [float(value)]
        out: List = [None] * self.world_size
        dist.all_gather_object(out, float(value), group=self.group)
        return [float(v) for v in out]

    def max_over_ranks(self, value: float) -> float:
        """Control-plane MAX reduction of a host scalar (bench timing)."""
        if self.world_size == 1:
            return value
        t = torch.tensor([value], dtype=torch.float64)
        if dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def barrier(self):
        if self.world_size > 1:
            dist.barrier(group=self.group)

    wait_for_everyone = barrier


def init_from_env(backend: Optional[str] = None, set_device: bool = True, join_timeout_s: float = 300.0) -> Communicator:
    """Initialise the control-plane process group from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun's env);
    no-op for a single process.  Default backend "gloo": the data path does not use it (see Communicator).
    set_device=False: do not bind this process to cuda:LOCAL_RANK (host-only runs: the CPU tests, `bench.py --dry-run`)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if set_device and torch.cuda.is_available():
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if local >= torch.cuda.device_count():
            raise RuntimeError(f"LOCAL_RANK {local} but this node has {torch.cuda.device_count()} GPU(s): one process per GPU")
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        rank = int(os.environ.get("RANK", "0"))
        _await_all_ranks(rank, world, float(os.environ.get("PIGEON_JOIN_TIMEOUT_S", join_timeout_s)))
        rdzv = os.environ.get("PIGEON_RDZV_FILE")
        if rdzv:
            # a launcher that owns its ranks (bench.py self_launch) hands them a FILE store: no fixed TCP port to lose a race for
            # between the launcher's "find a free port" and rank 0's bind (gloo's own pair connections use ephemeral ports)
            dist.init_process_group(backend=backend or "gloo", init_method=f"file://{rdzv}", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend=backend or "gloo")
        comm = Communicator()
        comm._owns_group = True
        return comm
    return Communicator()


def _await_all_ranks(rank: int, world: int, timeout_s: float) -> None:
    """Bounded wait that NAMES who is missing, in front of torch's own rendezvous.  torch's env:// rendezvous just blocks (30
    minutes by default) when a rank never shows up -- on an 8-GPU launch the useful message is "rank(s) [5] did not join".
    ONE NODE ONLY (the contract of this path: one process per GPU of one node): the check runs when WORLD_SIZE equals
    LOCAL_WORLD_SIZE (or LOCAL_WORLD_SIZE is unset); on a multi-node launch the ranks of the other nodes can never show up in this
    node's /tmp, so it is skipped.  Every rank drops a file `rank_<r>` into a per-user directory named after MASTER_PORT (mode 0700,
    owned by this uid, not a symlink; the file is created with O_EXCL | O_NOFOLLOW, so nothing a stranger planted there is followed
    or truncated) and waits until all `world` files are there or `timeout_s` is over.  Files are removed at exit and on SIGTERM (what
    torchrun and bench.py's launcher send); a stale file of a killed run only weakens the check.  It must never break a launch that
    would have worked: any filesystem trouble skips it, and it only RAISES when it has seen at least one other rank's file (proof
    that the ranks share the directory) -- a rank that sees nobody (ranks in separate containers / TMPDIRs) prints a note and lets
    torch's rendezvous do its own waiting."""
    import atexit
    import signal
    import tempfile
    import time
    local_world = os.environ.get("LOCAL_WORLD_SIZE")
    if local_world is not None and local_world.isdigit() and int(local_world) != world:
        return                                                 # multi-node: the other nodes' ranks are not visible here
    try:
        uid = os.getuid()
        d = os.path.join(tempfile.gettempdir(), f"pigeon_join_{uid}_{os.environ.get('MASTER_PORT', '0')}")
        try:
            os.mkdir(d, 0o700)
        except FileExistsError:
            pass
        st = os.lstat(d)
        import stat as _stat
        if not _stat.S_ISDIR(st.st_mode) or st.st_uid != uid or (st.st_mode & 0o077):
            return                                             # not ours / a symlink / open to others: do not touch it
        mine = os.path.join(d, f"rank_{rank}")
        try:
            os.unlink(mine)                                    # a stale file of an earlier run of this user on this port
        except FileNotFoundError:
            pass
        fd = os.open(mine, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
        with os.fdopen(fd, "w") as f:
            f.write(str(os.getpid()))

        def _cleanup():
            try:
                os.remove(mine)
                os.rmdir(d)
            except OSError:
                pass
        atexit.register(_cleanup)
        try:                                                   # SIGTERM skips atexit: remove the file, then die as before
            prev = signal.getsignal(signal.SIGTERM)
            if prev in (signal.SIG_DFL, None):
                def _on_term(signum, frame):
                    _cleanup()
                    signal.signal(signal.SIGTERM, signal.SIG_DFL)
                    os.kill(os.getpid(), signal.SIGTERM)
                signal.signal(signal.SIGTERM, _on_term)
        except (ValueError, OSError):                          # not the main thread
            pass
    except OSError:
        return
    t_end = time.time() + timeout_s
    seen_other = False
    while True:
        missing = [r for r in range(world) if not os.path.exists(os.path.join(d, f"rank_{r}"))]
        if not missing:
            return
        seen_other = seen_other or len(missing) < world - 1
        if time.time() > t_end:
            if not seen_other:                                 # nobody else was ever visible: no proof of a shared directory
                import sys
                print(f"[pigeon_amd] rank {rank}: saw no other rank in {d} within {timeout_s:.0f} s; leaving the wait to torch's "
                      f"rendezvous", file=sys.stderr)
                return
            raise TimeoutError(f"rank {rank}: rank(s) {missing} of {world} did not join within {timeout_s:.0f} s "
                               f"(MASTER_PORT {os.environ.get('MASTER_PORT')})")
        time.sleep(0.05)


def gpu_cpu_affinity(local_cpulists: List[Optional[str]], local_rank: int) -> Optional[List[int]]:
    """Host cores for the process that drives GPU `local_rank`: the cores of the GPU's NUMA node (`local_cpulist` of its PCI
    device), split evenly between the local ranks whose GPUs hang off the same node.  local_cpulists[r] is rank r's sysfs string
    ("0-31,128-159") or None (unknown -> no pinning).  Pure function: tests/test_distributed_cpu.py."""
    mine = local_cpulists[local_rank] if 0 <= local_rank < len(local_cpulists) else None
    if not mine:
        return None
    cpus = parse_cpulist(mine)
    peers = [r for r, s in enumerate(local_cpulists) if s and parse_cpulist(s) == cpus]
    if not cpus or local_rank not in peers:
        return None
    k, n = peers.index(local_rank), len(peers)
    per = len(cpus) // n
    if per == 0:
        return cpus
    return cpus[k * per:(k + 1) * per]


def parse_cpulist(s: str) -> List[int]:
    out: List[int] = []
    for part in s.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return sorted(set(out))


def gpu_local_cpulist(device_index: int) -> Optional[str]:
    """sysfs `local_cpulist` of the PCI device behind cuda:<device_index> (None when it cannot be read)."""
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as f:
            return f.read().strip() or None
    except Exception:  # noqa
        return None


def pin_to_gpu_numa(local_rank: int, local_world: int) -> Optional[List[int]]:
    """sched_setaffinity of this process to its share of the cores next to its GPU; returns the cores (None: not pinned)."""
    try:
        lists = [gpu_local_cpulist(r) for r in range(local_world)]
        cpus = gpu_cpu_affinity(lists, local_rank)
        if cpus:
            allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
            if allowed:
                os.sched_setaffinity(0, allowed)
                return allowed
    except Exception:  # noqa
        pass
    return None


def _is_sample_list(b) -> bool:
    """A per-sample python sequence leaf, e.g. the list of file names default_collate puts next to the tensors."""
    return isinstance(b, (list, tuple)) and not hasattr(b, "_fields") and len(b) > 0 and not any(
        torch.is_tensor(x) or isinstance(x, (dict, list, tuple)) for x in b)


def _tree_map(fn, *bs):
    """Apply fn(*leaves) to the per-sample leaves of parallel batches: tensors, and python lists of per-sample scalars /
    strings (both support `+`-style concatenation and slicing, which is all the padding below needs).  Containers (dict,
    list, tuple, namedtuple) are rebuilt; anything else (python scalars, None) is carried over from the first batch."""
    b = bs[0]
    if torch.is_tensor(b) or _is_sample_list(b):
        return fn(*bs)
    if isinstance(b, dict):
        return {k: _tree_map(fn, *[x[k] for x in bs]) for k in b}
    if isinstance(b, (list, tuple)):
        items = [_tree_map(fn, *[x[i] for x in bs]) for i in range(len(b))]
        return type(b)(*items) if hasattr(b, "_fields") else type(b)(items)            # namedtuple batches
    return b                                                           # python scalars, None, strings: per-batch, not per-sample


def _cat(*leaves):
    if torch.is_tensor(leaves[0]):
        return torch.cat([x.to(leaves[0].dtype) for x in leaves], dim=0)
    out = []
    for x in leaves:
        out.extend(x)
    return type(leaves[0])(out)


def _first_tensor(b):
    """First tensor leaf anywhere in a (nested dict / list / tuple) batch, or None."""
    if torch.is_tensor(b):
        return b
    if isinstance(b, dict):
        it = b.values()
    elif isinstance(b, (list, tuple)):
        it = b
    else:
        return None
    for v in it:
        t = _first_tensor(v)
        if t is not None:
            return t
    return None


def _batch_len(b) -> int:
    """Samples in a batch = leading dimension of its first TENSOR leaf (a dict whose first value is a list of strings,
    or a scalar, must not decide it); a flat list of non-tensors counts its items; an opaque object is dealt whole."""
    t = _first_tensor(b)
    if t is not None:
        return int(t.shape[0]) if t.dim() > 0 else 1
    if isinstance(b, (list, tuple)):
        return len(b)
    return 1


def shard_batches(batches: Iterable, rank: int, world_size: int, even: bool = True) -> Iterator:
    """Deal whole batches round-robin to ranks, as accelerate's BatchSamplerShard does for
    `accelerator.prepare(DataLoader)` with split_batches=False (batch i -> rank i % world_size; reference
    preprocessing/embed.py:68).

    even=True (accelerate's even_batches default): every rank runs the same number of steps AND every batch it
    yields has the full batch size -- the all-gather needs equal per-rank row counts.  A short final batch
    (DataLoader drop_last=False) is completed at SAMPLE level with samples taken from the start of the data, and
    the ranks left without a batch in the last round get filler batches cut from the same wrap-around stream,
    exactly as BatchSamplerShard does; the duplicates are removed downstream by the gathered sample indices
    (reference preprocessing/dataset_preprocessing.py:299-300 sorts by index and keeps the first num_samples)."""
    if world_size == 1:
        yield from batches
        return
    head: List = []            # the first `world_size` batches = the wrap-around sample stream (accelerate: initial_data)
    group: List = []
    bs = None
    for b in batches:
        if bs is None:
            bs = _batch_len(b)
        if len(head) < world_size:
            head.append(b)
        group.append(b)
        if len(group) == world_size and _batch_len(b) == bs:
            yield group[rank]
            group = []
    if not group:
        return
    if not even:
        if rank < len(group):
            yield group[rank]
        return
    if _first_tensor(head[0]) is None and not _is_sample_list(head[0]):
        if isinstance(head[0], (dict, list, tuple)):
            raise ValueError("shard_batches(even=True): the batch holds no tensor leaf to take the batch size from; "
                             "pass tensors (or a flat list of samples), or even=False")
        i = 0                                                            # opaque batches: whole-batch wrap-around
        while len(group) < world_size:
            group.append(head[i % len(head)])
            i += 1
        yield group[rank]
        return
    pool = _tree_map(_cat, *head)
    while _batch_len(pool) < bs * (world_size + 1):                 # enough samples for one padded + W-1 filler batches
        pool = _tree_map(_cat, pool, pool)
    cursor = 0

    def take(n):
        nonlocal cursor
        piece = _tree_map(lambda t: t[cursor:cursor + n], pool)
        cursor += n
        return piece

    last = group[-1]
    short = bs - _batch_len(last)
    if short > 0:
        group[-1] = _tree_map(_cat, last, take(short))
    while len(group) < world_size:
        group.append(take(bs))
    yield group[rank]


def restore_order(gathered_indices: torch.Tensor, *tensors: torch.Tensor):
    """Undo rank interleaving with the gathered sample indices (dataset_preprocessing.py:299-300): stable argsort
    of the indices, first occurrence kept for wrapped-around duplicates."""
    idx = gathered_indices.cpu()
    order = torch.argsort(idx, stable=True)
    sorted_idx = idx[order]
    keep = torch.ones_like(sorted_idx, dtype=torch.bool)
    keep[1:] = sorted_idx[1:] != sorted_idx[:-1]
    order = order[keep]
    return [t.cpu()[order] for t in tensors]
