"""Rank-strided dataset sharding.

Produces, index for index, the sequence ``torch.utils.data.DistributedSampler`` yields
(which is what reference distributed.py:105-108 returns): optional per-epoch permutation
from ``Generator(seed + epoch)``, padding to a multiple of the world size by repeating
the head of the list (or truncation with ``drop_last``), then ``indices[rank::world]``.
Rank and world size come from this package's process-group state, not from c10d.
"""
from __future__ import annotations

import math
from typing import Iterator, Optional, Sized

import torch
from torch.utils.data import Sampler


class ShardedSampler(Sampler[int]):
    def __init__(self, dataset: Sized, num_replicas: Optional[int] = None, rank: Optional[int] = None, shuffle: bool = True, seed: int = 0, drop_last: bool = False):
        from . import process_group as pg

        if num_replicas is None:
            num_replicas = pg.get_world_size()
        if rank is None:
            rank = pg.get_rank()
        if rank >= num_replicas or rank < 0:
            raise ValueError(f"Invalid rank {rank}, rank should be in the interval [0, {num_replicas - 1}]")
        self.dataset = dataset
        self.num_replicas = int(num_replicas)
        self.rank = int(rank)
        self.epoch = 0
        self.drop_last = bool(drop_last)
        n = len(dataset)
        if self.drop_last and n % self.num_replicas != 0:
            self.num_samples = math.ceil((n - self.num_replicas) / self.num_replicas)
        else:
            self.num_samples = math.ceil(n / self.num_replicas)
        self.total_size = self.num_samples * self.num_replicas
        self.shuffle = bool(shuffle)
        self.seed = int(seed)

    def global_indices(self):
        """The padded / truncated index list before striding (same on every rank)."""
        n = len(self.dataset)
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            indices = torch.randperm(n, generator=g).tolist()
        else:
            indices = list(range(n))
        if not self.drop_last:
            pad = self.total_size - len(indices)
            if pad > 0:
                if pad <= len(indices):
                    indices += indices[:pad]
                else:
                    indices += (indices * math.ceil(pad / len(indices)))[:pad]
        else:
            indices = indices[: self.total_size]
        assert len(indices) == self.total_size
        return indices

    def __iter__(self) -> Iterator[int]:
        shard = self.global_indices()[self.rank : self.total_size : self.num_replicas]
        assert len(shard) == self.num_samples
        return iter(shard)

    def __len__(self) -> int:
        return self.num_samples

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)
