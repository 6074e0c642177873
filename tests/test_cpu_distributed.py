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
    per_rank = D.gather_floats(10.0 + r, "cpu")
    info = D.world_info()
    D.barrier()
    q.put((r, mine, [float(g.mean()) for g in got], wts[0].tolist(), t, per_rank, info))


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
        assert r[5] == [10.0, 11.0] and r[6] == (2, "gloo")      # bench.py's per_rank_ms_per_step / rccl_world + dist_backend fields


def test_single_process_helpers_are_noops():
    from evoworld_amd import distributed as D
    x = torch.ones(2)
    assert D.gather_results(x)[0] is x and D.shard_clips(3, 0, 1) == [0, 1, 2] and D.max_over_ranks(3.0, "cpu") == 3.0
    assert D.gather_floats(2.5, "cpu") == [2.5] and D.world_info() == (1, None)


def _cfg_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from evoworld_amd import distributed as D
    from oracle.reproject_ref import euler_cfg_step_ref
    r, w, _ = D.init(backend="gloo")
    grp = D.CfgGroup(r, w, size=2)
    T, h, ww = 3, 4, 8
    rows = T * h * ww
    g = torch.Generator().manual_seed(5)
    eps_full = torch.randn(2, rows, 4, generator=g).half()                     # what a B=2 forward would return (synthetic)
    lat = torch.randn(1, T, 4, h, ww, generator=g)
    guid = torch.linspace(1.0, 3.0, T)
    outs = []
    for step in range(3):                                                      # a few steps: the buffer is reused
        eps_all = torch.zeros(2, rows, 4, dtype=torch.float16)
        mine = None
        for row in grp.rows():
            mine = (eps_full[row].float() * (step + 1)).half()                  # "this rank's forward output" of its row
            eps_all[row].copy_(mine)
        grp.all_gather_rows(eps_all, mine)
        e = eps_all.float().reshape(2, T, h, ww, 4).permute(0, 1, 4, 2, 3)
        lat = euler_cfg_step_ref(e[0:1], e[1:2], lat, guid, 10.0 / (step + 1), 5.0 / (step + 1))
        outs.append(eps_all.clone())
    q.put((r, grp.rows(), grp.pair, grp.n_pairs, torch.stack(outs).numpy(), lat.numpy()))   # numpy: no fd passing after exit


def test_cfg_pair_exchange_and_combine_match_single_rank():
    """CFG-pair axis (north_star 'denoising-step batch'): rank 0 owns the unconditional row, rank 1 the conditional one; after
    the per-step all_gather both ranks hold the full eps batch and the replicated CFG + Euler combine is bit-identical to the
    single-rank result on the same eps."""
    from oracle.reproject_ref import euler_cfg_step_ref
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_cfg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=120) for _ in ps), key=lambda t: t[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0] and res[1][1] == [1] and res[0][2] == res[1][2] == 0 and res[0][3] == 1
    T, h, ww = 3, 4, 8
    g = torch.Generator().manual_seed(5)
    eps_full = torch.randn(2, T * h * ww, 4, generator=g).half()
    lat = torch.randn(1, T, 4, h, ww, generator=g)
    guid = torch.linspace(1.0, 3.0, T)
    for step in range(3):
        e16 = (eps_full.float() * (step + 1)).half()
        for r in res:
            assert torch.equal(torch.from_numpy(r[4][step]), e16)                                 # both members hold both rows, in CFG order
        e = e16.float().reshape(2, T, h, ww, 4).permute(0, 1, 4, 2, 3)
        lat = euler_cfg_step_ref(e[0:1], e[1:2], lat, guid, 10.0 / (step + 1), 5.0 / (step + 1))
    assert torch.equal(torch.from_numpy(res[0][5]), lat) and torch.equal(torch.from_numpy(res[1][5]), lat)          # replicated combine == single-rank combine, bit-exact


def test_cfg_group_degenerate_and_validation():
    import pytest
    from evoworld_amd import distributed as D
    g1 = D.CfgGroup(0, 1, size=1)
    assert g1.rows() == [0, 1] and g1.n_pairs == 1
    x = torch.arange(8.0).reshape(2, 2, 2)
    assert g1.all_gather_rows(x, x[1]) is x                                    # no process group: nothing to exchange
    with pytest.raises(ValueError):
        D.CfgGroup(0, 3, size=2)
    with pytest.raises(ValueError):
        D.CfgGroup(0, 4, size=4)


def _cfg4_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from evoworld_amd import distributed as D
    r, w, _ = D.init(backend="gloo")
    grp = D.CfgGroup(r, w, size=2)
    members = dist.get_process_group_ranks(grp.group)
    rows = 24
    outs = []
    for step in range(2):
        # "this rank's forward output": depends on the CLIP (= pair) and on the CFG row, not on the rank id directly
        eps_all = torch.zeros(2, rows, 4, dtype=torch.float16)
        (row,) = grp.rows()
        mine = torch.full((rows, 4), float(100 * grp.pair + 10 * row + step), dtype=torch.float16)
        eps_all[row].copy_(mine)
        grp.all_gather_rows(eps_all, mine)
        outs.append(eps_all.clone())
    # one clip per pair: the clip sharding of the cfg2 x dp(N/2) layout
    clips = D.shard_clips(4, grp.pair, grp.n_pairs)
    D.barrier()
    q.put((r, grp.pair, grp.member, grp.n_pairs, members, grp.rows(), clips, torch.stack(outs).numpy()))


def test_cfg_two_pairs_world4_gloo():
    """cfg2 x dp2 (what `bench.py --gpus 4 --split cfg` builds, and cfg2 x dp4 at 8 GPUs): ranks (0,1) and (2,3) form two pairs, `new_group`
    is called for every pair on every rank in the same order, the members of a pair end up bit-identical, different pairs carry different
    clips and never see each other's rows."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_cfg4_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=180) for _ in ps), key=lambda t: t[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r, pair, member, n_pairs, members, rows, clips, outs in res:
        assert pair == r // 2 and member == r % 2 and n_pairs == 2
        assert members == [2 * pair, 2 * pair + 1]                 # the group this rank kept is its own pair's
        assert rows == [member]                                    # rank 2p: unconditional row, 2p+1: conditional row
        assert clips == [pair, pair + 2]                           # clips are sharded over PAIRS
        for step in range(2):
            want = torch.stack([torch.full((24, 4), float(100 * pair + 10 * row + step)) for row in range(2)]).half()
            assert torch.equal(torch.from_numpy(outs[step]), want)
    assert (res[0][7] == res[1][7]).all() and (res[2][7] == res[3][7]).all()      # pair members bit-identical
    assert (res[0][7] != res[2][7]).any()                                          # different pairs, different clips
