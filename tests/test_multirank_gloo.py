"""N>1 path on CPU: world_size-2 gloo ranks exercise the sharding / barrier / max-over-ranks logic bench.py uses
(the GPU work itself is per-rank and independent: no collective on the data path)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import time
    import __graft_entry__ as g
    g.load_package()
    from tamcmc_c_amd import shard, synth
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    r, lr, w = shard.rank_info()
    mine = shard.stars_of_rank(8, r, w)
    star = synth.make_c2_star(seed=20240229 + r, nx=512)   # every rank builds ITS star (different seed)
    steps = 50

    def work():
        time.sleep(0.05 * (rank + 1))   # uneven ranks: the slowest one sets the time
        return float(star.params.sum())

    elapsed, chk = shard.timed_region(work, dist=dist)
    value = shard.aggregate_rate(steps, w, elapsed)
    sums = [None] * w
    dist.all_gather_object(sums, chk)
    q.put((r, mine, elapsed, value, sums))
    dist.destroy_process_group()


def test_two_ranks_shard_and_time():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, e0, v0, sums0), (r1, s1, e1, v1, sums1) = out
    assert s0 == [0, 2, 4, 6] and s1 == [1, 3, 5, 7]           # stars round-robin over ranks, disjoint, complete
    assert e0 == pytest.approx(e1) and e0 >= 0.1                # both ranks report the MAX (slowest rank: 0.1 s)
    assert v0 == pytest.approx(2 * 50 / e0)                     # whole-job rate = all ranks' units / max time
    assert sums0 == sums1 and sums0[0] != sums0[1]              # ranks really worked on different stars
