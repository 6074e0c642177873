"""-m 'not gpu': the N>1 path (one process per device, clip sharding + result gather) on CPU with gloo, world_size 2."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from evoworld_amd import distributed as D
    r, w, _ = D.init(backend="gloo")
    mine = D.shard_clips(5, r, w)
    x = torch.full((1, 3, 4, 2, 2), float(r + 1))
    got = D.gather_results(x)
    wts = [torch.arange(6, dtype=torch.float32) * (1 if r == 0 else 0)]
    D.broadcast_tensors(wts, src=0)
    t = D.max_over_ranks(1.0 + r, "cpu")
    D.barrier()
    q.put((r, mine, [float(g.mean()) for g in got], wts[0].tolist(), t))


def test_two_rank_gloo_shard_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]
    for r in res:
        assert r[2] == [1.0, 2.0] and r[3] == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0] and r[4] == 2.0


def test_single_process_helpers_are_noops():
    from evoworld_amd import distributed as D
    x = torch.ones(2)
    assert D.gather_results(x)[0] is x and D.shard_clips(3, 0, 1) == [0, 1, 2] and D.max_over_ranks(3.0, "cpu") == 3.0
