"""Multi-GPU batch embedding, behind the reference's call surface (preprocessing/embed.py).

`compute_embeddings(name, model, data, accelerator)` and `embed_images(loaded_model, dataset)` keep the
reference's names, arguments and on-disk outputs (`data/landmark_embeddings/{name}.npy` = (steps, world*batch, 1024)
float32 array of the gathered batches, `{name}_indices.npy` = (steps, world*batch) int64; embed.py:41-43).  `accelerate.Accelerator` is replaced by
`pigeon_amd.distributed.Communicator` (same `.gather`, `.is_local_main_process`, `.wait_for_everyone`), one
process per GPU, RCCL all-gather over xGMI.
"""
from __future__ import annotations

import logging
import os
from typing import Any, Iterable, Optional

import numpy as np
import torch

from .config import EMBED_BATCH_SIZE_PER_GPU
from .distributed import Communicator, shard_batches

logging.basicConfig(level=logging.INFO)
logger = logging.getLogger('embed')


class EmbedDataset:
    """reference dataset_creation/finetune/embed_dataset.py:6-25: yields (pixel_values (3,336,336), index).
    Items whose 'image' is already a tensor are passed through; PIL images go through the restated CLIP
    preprocessing (pigeon_amd.clip_embedder.clip_preprocess)."""

    def __init__(self, dataset):
        self.dataset = dataset

    def __getitem__(self, idx):
        data = self.dataset[idx]
        image = data['image']
        if not torch.is_tensor(image):
            from .clip_embedder import clip_preprocess
            image = clip_preprocess(image)
        return image.squeeze(), data['index']

    def __len__(self):
        return len(self.dataset)


def _stack_padded(batches, fill):
    """np.stack of per-step arrays; a shorter last step is padded to the common length (fill None: indices -> int max so
    they sort last)."""
    if not batches:
        return np.zeros((0,))
    n = max(b.shape[0] for b in batches)
    out = []
    for b in batches:
        if b.shape[0] < n:
            pad_shape = (n - b.shape[0],) + b.shape[1:]
            v = np.iinfo(b.dtype).max if fill is None else fill
            b = np.concatenate([b, np.full(pad_shape, v, dtype=b.dtype)], axis=0)
        out.append(b)
    return np.stack(out)


def compute_embeddings(name: str, model: Any, data: Iterable, accelerator: Communicator,
                       out_dir: str = 'data/landmark_embeddings', save: bool = True):
    """reference preprocessing/embed.py:16-43.  `data` yields (pixels, index) batches ALREADY sharded for this
    rank (see embed_images).  Every step: output = model(pixels); all-gather index and output (rank-major, one
    collective); rank 0 keeps the numpy copies and finally np.save's them."""
    logger.warning(f'Starting {name} embedding ...')
    all_outputs, all_indices = [], []
    for pixels, index in data:
        output = model(pixels)
        index = torch.as_tensor(index).to(output.device)
        all_indic, all_output = accelerator.gather_many([index, output])       # embed.py:36-37 in one collective
        all_outputs.append(all_output.cpu().detach().numpy())
        all_indices.append(all_indic.cpu().detach().numpy())
    if accelerator.is_local_main_process and save:
        os.makedirs(out_dir, exist_ok=True)
        # plain numeric arrays (steps, world*batch, 1024) / (steps, world*batch): what the reference's np.save of a list of
        # equal-shape batches produces (:41-43) and what its reader np.load's WITHOUT allow_pickle
        # (preprocessing/dataset_preprocessing.py:294-300).  A ragged last batch (single process, drop_last=False) is
        # padded with zero rows whose index is INT64_MAX, which sort behind every real sample in that reader's argsort
        # -- the reference itself cannot save that case with numpy >= 1.24.
        np.save(f'{out_dir}/{name}.npy', _stack_padded(all_outputs, 0.0))
        np.save(f'{out_dir}/{name}_indices.npy', _stack_padded(all_indices, None))
    return all_outputs, all_indices


def embed_images(loaded_model: Any, dataset, accelerator: Optional[Communicator] = None,
                 batch_size: int = EMBED_BATCH_SIZE_PER_GPU, num_workers: int = 8,
                 out_dir: str = 'data/landmark_embeddings'):
    """reference preprocessing/embed.py:45-83: wrap every split in EmbedDataset, one DataLoader per split
    (bs 512 per GPU, shuffle False), shard batches over ranks, embed train/val/test with a barrier between."""
    from torch.utils.data import DataLoader
    accelerator = accelerator or Communicator()
    results = {}
    loaded_model.eval()
    for split_name in ('train', 'val', 'test'):
        if split_name not in dataset:
            continue
        loader = DataLoader(EmbedDataset(dataset[split_name]), batch_size, shuffle=False, num_workers=num_workers)
        sharded = shard_batches(loader, accelerator.rank, accelerator.world_size)
        results[split_name] = compute_embeddings(split_name, loaded_model, sharded, accelerator, out_dir)
        accelerator.wait_for_everyone()
    return results
