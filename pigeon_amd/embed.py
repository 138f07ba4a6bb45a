"""Multi-GPU batch embedding, behind the reference's call surface (preprocessing/embed.py).

`compute_embeddings(name, model, data, accelerator)` and `embed_images(loaded_model, dataset)` keep the
reference's names, arguments and on-disk outputs (`data/landmark_embeddings/{name}.npy` = list of gathered
batches, `{name}_indices.npy`; embed.py:41-43).  `accelerate.Accelerator` is replaced by
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
        np.save(f'{out_dir}/{name}.npy', np.array(all_outputs, dtype=object), allow_pickle=True)
        np.save(f'{out_dir}/{name}_indices.npy', np.array(all_indices, dtype=object), allow_pickle=True)
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
