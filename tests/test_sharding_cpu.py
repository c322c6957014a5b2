"""Multi-process path on CPU (gloo, world_size 2): shard arithmetic, the single all-gather of episode returns,
and independence of the results from the world size (RNG streams keyed by the GLOBAL env id).  The per-shard
numbers come from the CPU oracle here (this is a test; on the GPU box the same code gathers the HIP env's returns)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gym_sbr2_amd.sharding import gather_returns, gather_returns_into, local_device, shard_range


def test_shard_ranges_partition_the_batch():
    for n, w in [(262144, 8), (10, 3), (7, 8), (65536, 1), (0, 2)]:
        r = [shard_range(n, k, w) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        sizes = [b - a for a, b in r]
        assert max(sizes) - min(sizes) <= 1
    assert shard_range(262144, 3, 8) == (3 * 32768, 4 * 32768)
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _shard_returns(n_global, start, stop, steps=30):
    """Episode-return-like numbers for global env ids [start, stop) from the oracle: device-style Philox influent
    noise (seed 5) and random policy (seed 9), both keyed by global env id."""
    from oracle import sbr_oracle as O
    t = np.load(os.path.join(os.path.dirname(__file__), "golden", "influent_tables.npz"))
    n = stop - start
    b = O.OracleBatch(n, first_env_id=start)
    scen = (np.arange(start, stop) % 8).astype(np.int32)
    b.reset(b.mix(t["means"], t["stds"], scen, b.normals(5)))
    return b.rollout(steps, 9)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, n_global, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        start, stop = shard_range(n_global, rank, world)
        local = torch.from_numpy(_shard_returns(n_global, start, stop)).to(torch.float32)
        full = gather_returns(local, n_global)
        if n_global % world == 0:          # the allocation-free form bench.py uses: float64 row -> float32 send -> preallocated recv
            row = torch.from_numpy(_shard_returns(n_global, start, stop))
            send, recv = torch.empty(stop - start, dtype=torch.float32), torch.full((n_global,), float("nan"), dtype=torch.float32)
            addr = recv.data_ptr()
            out = gather_returns_into(row, send, recv)
            assert out is recv and recv.data_ptr() == addr and torch.equal(recv, full)
            with pytest.raises(ValueError):
                gather_returns_into(row, send, torch.empty(n_global + 1, dtype=torch.float32))
        dist.barrier()
        np.save(os.path.join(out_dir, "rank%d.npy" % rank), full.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_global", [64, 37])          # equal shards, and ragged shards (padded gather)
def test_world_size_2_gather_matches_single_process(n_global, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, n_global, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / ("rank%d.npy" % r)) for r in range(world)]
    single = _shard_returns(n_global, 0, n_global).astype(np.float32)
    assert np.array_equal(got[0], got[1])                # every rank holds the same collated vector
    assert np.array_equal(got[0], single)                # and it does not depend on the world size
    assert got[0].shape == (n_global,) and np.isfinite(got[0]).all()


def test_gather_without_process_group_is_identity():
    v = torch.arange(5, dtype=torch.float32)
    assert torch.equal(gather_returns(v, 5), v)
    with pytest.raises(ValueError):
        gather_returns(v, 6)


def test_gather_into_without_process_group_is_a_cast():
    row = torch.arange(5, dtype=torch.float64) / 3
    send, recv = torch.empty(5, dtype=torch.float32), torch.empty(5, dtype=torch.float32)
    assert gather_returns_into(row, send, recv) is recv and torch.equal(recv, row.to(torch.float32))
    with pytest.raises(ValueError):
        gather_returns_into(row, send, torch.empty(10, dtype=torch.float32))


def test_local_device_follows_local_rank():
    assert local_device(env={"LOCAL_RANK": "3"}, n_devices=8) == 3
    assert local_device(env={"LOCAL_RANK": "5"}, n_devices=4) == 1       # more ranks than visible devices: wrap, never fail
    assert local_device(env={"LOCAL_RANK": "0"}, n_devices=0) == 0
    assert local_device(env={}, n_devices=8) == 0                        # no launcher, no GPU here: torch's current device


def _device_worker(rank, world, port, out_dir):
    """Each spawned rank builds the recommended class with no `device` argument, the way torch.distributed.run would start
    it (LOCAL_RANK in the environment, every GPU visible); SbrOSVec is replaced by a recorder - no GPU is touched."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gym_sbr2_amd.sharding as S
        import gym_sbr2_amd.vec_env as V
        seen = {}

        class Recorder:
            def __init__(self, num_envs, device=0, first_env_id=0, **kw):
                seen.update(num_envs=num_envs, device=device, first_env_id=first_env_id)
        V.SbrOSVec = Recorder
        real_count = torch.cuda.device_count
        torch.cuda.device_count = lambda: 8
        try:
            sh = S.ShardedSbrOS(1000)
        finally:
            torch.cuda.device_count = real_count
        np.save(os.path.join(out_dir, "dev%d.npy" % rank), np.array([seen["device"], seen["first_env_id"], seen["num_envs"], sh.device]))
    finally:
        dist.destroy_process_group()


def test_sharded_env_puts_rank_r_on_device_r(tmp_path):
    world = 2
    mp.spawn(_device_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / ("dev%d.npy" % r)).tolist() for r in range(world)]
    assert got[0] == [0, 0, 500, 0] and got[1] == [1, 500, 500, 1]
