"""One-process-per-GPU data parallelism for the inference path.

The reference's only inference-time collective is `accelerator.gather` (preprocessing/embed.py:36-37): an
all-gather, concatenating every rank's tensor along dim 0 in rank order.  Here that is
`torch.distributed.all_gather_into_tensor` on the process group -- backend "nccl" is RCCL over xGMI on ROCm,
"gloo" on CPU for tests -- wrapped in a tiny Communicator with accelerate's semantics, plus the batch sharding
`accelerator.prepare(DataLoader)` performs (whole batches dealt round-robin to ranks, embed.py:68).
"""
from __future__ import annotations

import os
from typing import Iterable, Iterator, List, Optional

import torch
import torch.distributed as dist


class Communicator:
    """rank / world_size + rank-major all-gather.  world_size == 1 needs no process group."""

    def __init__(self, group: Optional[dist.ProcessGroup] = None):
        self.group = group
        if dist.is_available() and dist.is_initialized():
            self.rank = dist.get_rank(group)
            self.world_size = dist.get_world_size(group)
        else:
            self.rank, self.world_size = 0, 1

    @property
    def is_main_process(self) -> bool:
        return self.rank == 0

    is_local_main_process = is_main_process

    def gather(self, t: torch.Tensor) -> torch.Tensor:
        """accelerate.Accelerator.gather: (n, ...) on every rank -> (world*n, ...) rank-major, on every rank."""
        if self.world_size == 1:
            return t
        t = t.contiguous()
        out = torch.empty((self.world_size * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        if t.is_cuda:
            dist.all_gather_into_tensor(out, t, group=self.group)       # one RCCL all-gather over xGMI
        else:
            parts = list(out.chunk(self.world_size, dim=0))             # gloo: list form
            dist.all_gather(parts, t, group=self.group)
        return out

    def gather_many(self, tensors: List[torch.Tensor]) -> List[torch.Tensor]:
        """Gather several tensors with the same leading dimension in ONE collective: they are packed as raw bytes
        into a single (n, bytes_per_row) buffer, gathered once, and unpacked.  (Per step the payload is tiny --
        2 MiB of embeddings + a few KiB of indices -- so one launch beats four.)"""
        if self.world_size == 1:
            return list(tensors)
        n = tensors[0].shape[0]
        flat = [t.contiguous().view(n, -1).view(torch.uint8) for t in tensors]
        widths = [f.shape[1] for f in flat]
        packed = torch.cat(flat, dim=1)
        g = self.gather(packed)
        outs, off = [], 0
        for t, w in zip(tensors, widths):
            piece = g[:, off:off + w].contiguous().view(t.dtype).view((self.world_size * n,) + tuple(t.shape[1:]))
            outs.append(piece)
            off += w
        return outs

    def barrier(self):
        if self.world_size > 1:
            dist.barrier(group=self.group)

    wait_for_everyone = barrier


def init_from_env(backend: Optional[str] = None) -> Communicator:
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun's env).
    No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend=backend)
    return Communicator()


def shard_batches(batches: Iterable, rank: int, world_size: int, even: bool = True) -> Iterator:
    """Deal whole batches round-robin to ranks, as accelerate's BatchSamplerShard does for
    `accelerator.prepare(DataLoader)` with split_batches=False (batch i -> rank i % world_size).
    even=True (accelerate's even_batches default) wraps around to the first batches so that every rank runs
    the same number of steps -- required for the collective; the duplicates are removed downstream by the
    gathered sample indices (reference preprocessing/dataset_preprocessing.py:299-300 argsorts by index)."""
    if world_size == 1:
        yield from batches
        return
    head: List = []
    group: List = []
    for b in batches:
        if len(head) < world_size:
            head.append(b)
        group.append(b)
        if len(group) == world_size:
            yield group[rank]
            group = []
    if group:
        if not even:
            if rank < len(group):
                yield group[rank]
            return
        i = 0
        while len(group) < world_size:
            group.append(head[i % len(head)])
            i += 1
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
