"""N>1 path of bench.py on CPU: clips are partitioned over ranks with no data-path collective; one
all_gather of a small stats vector at the end (replaces the reference's mp.Queue, evaluator.py:507-531)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    clips = bench.shard_clips(7, rank, world)
    stats = torch.tensor([float(len(clips)), float(sum(clips)), 1.0 + rank], dtype=torch.float64)
    allst = bench.gather_stats(stats, world)
    dist.barrier()
    q.put((rank, clips, allst.tolist()))
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (r0, c0, s0), (r1, c1, s1) = res
    assert sorted(c0 + c1) == list(range(7)) and not set(c0) & set(c1)      # disjoint cover
    assert abs(len(c0) - len(c1)) <= 1
    assert s0 == s1 and len(s0) == 2                                        # every rank sees both stat rows
    assert s0[0][2] == 1.0 and s0[1][2] == 2.0
