"""The image partition of the data-parallel run (uwsod/detectron2/data/samplers/distributed_sampler.py:12-55,
uwsod/detectron2/utils/comm.py:220-231): against the fixture written by running the reference's TrainingSampler, and across four real
processes over gloo (seed agreement, disjoint shards that cover the shared permutation)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampler.npz")


def test_training_sampler_matches_the_reference_run():
    import sos_wsod_amd  # noqa: F401
    from sos_wsod_amd.samplers import TrainingSampler
    g = np.load(GOLDEN)
    names = sorted({k.split("/")[0] for k in g.files})
    assert names
    for name in names:
        size, seed, world, shuffle, n = (int(v) for v in g[f"{name}/cfg"])
        for r in range(world):
            got = TrainingSampler(size, shuffle=bool(shuffle), seed=seed, rank=r, world_size=world).take(n)
            assert got == g[f"{name}/idx"][r].tolist(), (name, r)


def test_resume_skips_what_was_consumed_and_batches_follow_the_shard():
    import sos_wsod_amd  # noqa: F401
    from sos_wsod_amd.samplers import TrainingSampler, sharded_batches
    s = TrainingSampler(37, seed=7, rank=3, world_size=8)
    full = s.take(40)
    assert s.take(10, skip=25) == full[25:35]
    data = [{"id": i} for i in range(37)]
    it = sharded_batches(data, 2, s, mapper=lambda d: d["id"], start_iter=5)
    assert next(it) == full[10:12] and next(it) == full[12:14]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    import sos_wsod_amd  # noqa: F401
    from sos_wsod_amd.samplers import TrainingSampler, shared_random_seed
    from sos_wsod_amd.trainer import init_distributed
    init_distributed(backend="gloo")
    np.random.seed(1000 + rank)                       # every rank draws something else: rank 0's draw must win (comm.py:229-231)
    first = shared_random_seed()
    s = TrainingSampler(23)                           # seed=None: the constructor asks the group itself (the second collective draw)
    seed = s._seed
    mine = torch.tensor(s.take(46), dtype=torch.int64)
    rows = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(rows, mine)
    seeds = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(seeds, torch.tensor([seed], dtype=torch.int64))
    if rank == 0:
        rs = np.random.RandomState(1000)
        want_first, want_second = int(rs.randint(2 ** 31)), int(rs.randint(2 ** 31))
        torch.save({"rows": torch.stack(rows), "seeds": torch.cat(seeds), "seed0": want_second, "first_ok": first == want_first}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_four_ranks_agree_on_the_seed_and_split_one_permutation_stream(tmp_path):
    """world 4 over gloo: (1) shared_random_seed hands rank 0's draw to every rank; (2) rank r's stream is stream[r::4] of ONE sequence
    shuffle(range(n)) + shuffle(range(n)) + ...: the ranks' index sets are disjoint inside an epoch window and together cover every
    image exactly once per epoch."""
    out = str(tmp_path / "s.pt")
    world = 4
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r = torch.load(out)
    assert r["first_ok"] and (r["seeds"] == r["seed0"]).all(), r["seeds"]
    rows = r["rows"].numpy()                                   # (4, 46)
    stream = rows.T.reshape(-1)                                # interleave: position p of the shared stream = rows[p % 4, p // 4]
    n = 23
    for e in range(len(stream) // n):
        assert sorted(stream[e * n:(e + 1) * n].tolist()) == list(range(n)), e          # every epoch of the stream is a permutation
    assert stream[:n].tolist() != list(range(n))                                        # shuffled
    first = [set(rows[k, :5].tolist()) for k in range(world)]                           # the first 20 positions: inside epoch 0
    assert all(first[a].isdisjoint(first[b]) for a in range(world) for b in range(a + 1, world))
    # and it is the sequence a single process generates from the same seed
    import sos_wsod_amd  # noqa: F401
    from sos_wsod_amd.samplers import TrainingSampler
    assert TrainingSampler(n, seed=int(r["seed0"]), rank=0, world_size=1).take(len(stream)) == stream.tolist()
