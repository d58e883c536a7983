"""Generate tests/golden/sampler.npz by RUNNING the reference's own TrainingSampler
(/root/reference/uwsod/detectron2/data/samplers/distributed_sampler.py:12-55), loaded by path in THIS container with `detectron2.utils.comm`
replaced by a two-function stand-in (get_rank / get_world_size: the sampler reads nothing else once a seed is given).  The fixture
holds inputs (size, seed, world size, shuffle) and each rank's first indices: data, no source text.

    python tests/golden/make_sampler_golden.py
"""
import importlib.util
import itertools
import os
import sys
import types

import numpy as np

REF = "/root/reference/uwsod/detectron2/data/samplers/distributed_sampler.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sampler.npz")

# (name, size, seed, world, shuffle, indices per rank)
CASES = [
    ("voc_w4", 5011, 1234, 4, True, 3000),          # VOC07 trainval, SEED 1234: more than two epochs of the stream over 4 ranks
    ("small_w8", 37, 7, 8, True, 40),
    ("w1", 11, 99, 1, True, 30),
    ("noshuffle_w3", 10, 5, 3, False, 12),
]


def main():
    comm = types.ModuleType("detectron2.utils.comm")
    state = {"rank": 0, "world": 1}
    comm.get_rank = lambda: state["rank"]
    comm.get_world_size = lambda: state["world"]
    comm.shared_random_seed = lambda: 0
    for name in ("detectron2", "detectron2.utils"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["detectron2.utils.comm"] = comm
    sys.modules["detectron2.utils"].comm = comm
    spec = importlib.util.spec_from_file_location("ref_distributed_sampler", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    for name, size, seed, world, shuffle, n in CASES:
        rows = []
        for r in range(world):
            state["rank"], state["world"] = r, world
            s = mod.TrainingSampler(size, shuffle=shuffle, seed=seed)
            rows.append([int(v) for v in itertools.islice(iter(s), n)])
        out[f"{name}/cfg"] = np.asarray([size, seed, world, int(shuffle), n], np.int64)
        out[f"{name}/idx"] = np.asarray(rows, np.int64)
    np.savez(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items() if k.endswith("idx")})


if __name__ == "__main__":
    main()
