"""The image partition of the data-parallel run: rank r of W takes `indices[r::W]` of ONE infinite index stream
`shuffle(range(size)) + shuffle(range(size)) + ...` that every rank generates from the same seed
(uwsod/detectron2/data/samplers/distributed_sampler.py:12-55 `TrainingSampler`); the seed, when the config does not fix one, is
rank 0's draw handed to everybody (uwsod/detectron2/utils/comm.py:220-231 `shared_random_seed`).

No data-path collective: the ranks agree because they run the same generator, not because they talk.  The stream is torch's CPU
`randperm` under a seeded `torch.Generator` — the reference's own call, so a run resumed here visits the reference's images in the
reference's order (pinned by tests/golden/sampler.npz, written by running the reference's class)."""
import itertools
from typing import Iterator, Optional

import numpy as np
import torch
import torch.distributed as dist


def shared_random_seed(group=None) -> int:
    """comm.py:220-231: every rank draws `np.random.randint(2 ** 31)`, rank 0's value wins (the reference all_gathers python ints over
    gloo and takes element 0; here one 8-byte broadcast from rank 0 — on the device when the group's backend needs device tensors).
    All ranks must call it."""
    ints = int(np.random.randint(2 ** 31))
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return ints
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.tensor([ints], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return int(t.item())


class TrainingSampler(torch.utils.data.Sampler):
    """distributed_sampler.py:12-55.  `rank` / `world_size` default to the process group's (1 process: the whole stream)."""

    def __init__(self, size: int, shuffle: bool = True, seed: Optional[int] = None, rank: Optional[int] = None,
                 world_size: Optional[int] = None):
        assert size > 0
        self._size, self._shuffle = int(size), bool(shuffle)
        if seed is None:
            seed = shared_random_seed()
        self._seed = int(seed)
        live = dist.is_available() and dist.is_initialized()
        self._rank = int(rank) if rank is not None else (dist.get_rank() if live else 0)
        self._world_size = int(world_size) if world_size is not None else (dist.get_world_size() if live else 1)
        assert 0 <= self._rank < self._world_size

    def __iter__(self) -> Iterator[int]:
        yield from itertools.islice(self._infinite_indices(), self._rank, None, self._world_size)

    def _infinite_indices(self) -> Iterator[int]:
        g = torch.Generator()
        g.manual_seed(self._seed)
        while True:
            if self._shuffle:
                yield from torch.randperm(self._size, generator=g).tolist()
            else:
                yield from range(self._size)

    def take(self, n: int, skip: int = 0):
        """this rank's indices number skip .. skip + n - 1 (a resumed run skips what its earlier iterations consumed)"""
        return list(itertools.islice(iter(self), skip, skip + n))


def sharded_batches(dataset, images_per_rank: int, sampler: TrainingSampler, mapper=None, start_iter: int = 0):
    """The infinite per-rank batch stream of build_detection_train_loader (uwsod/detectron2/data/build.py:343-381) without worker
    processes: `images_per_rank` consecutive indices of this rank's shard per iteration -> [mapper(dataset[i]) ...]."""
    it = itertools.islice(iter(sampler), start_iter * images_per_rank, None)
    while True:
        idx = list(itertools.islice(it, images_per_rank))
        yield [mapper(dataset[i]) if mapper is not None else dataset[i] for i in idx]
