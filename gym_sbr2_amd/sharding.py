"""Multi-GPU: environments are independent, so the N global envs are split into contiguous blocks, one
process (and one SbrOSVec handle) per GPU, with NO collective on the data path.  The only exchange is one
all-gather of the per-env episode returns per episode (RCCL over xGMI on the GPU box: backend "nccl";
"gloo" in the CPU tests).  Random streams and scenario assignment are keyed by the GLOBAL env id
(first_env_id in the C ABI), so results do not depend on the world size.

The reference has nothing distributed; this is the MI355X-side design for
BASELINE.json configs[3].
"""
import torch
import torch.distributed as dist


def shard_range(n_global, rank, world):
    """[start, stop) of the contiguous block of global env ids owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world) or n_global < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(int(n_global), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_returns(local, n_global, group=None):
    """All-gather the per-env episode returns of every rank into one [n_global] tensor (same on every rank),
    ordered by global env id.  One collective: all_gather_into_tensor when the shards are equal, else a padded one."""
    if not (dist.is_available() and dist.is_initialized()):
        if local.numel() != n_global:
            raise ValueError("no process group: local shard must be the whole batch")
        return local.clone()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    start, stop = shard_range(n_global, rank, world)
    if local.numel() != stop - start:
        raise ValueError("rank %d holds %d envs, expected %d" % (rank, local.numel(), stop - start))
    local = local.contiguous()
    if n_global % world == 0:
        out = torch.empty(n_global, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    width = -(-n_global // world)
    padded = torch.zeros(width, dtype=local.dtype, device=local.device)
    padded[: local.numel()] = local
    out = torch.empty(width * world, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    parts = []
    for r in range(world):
        a, b = shard_range(n_global, r, world)
        parts.append(out[r * width: r * width + (b - a)])
    return torch.cat(parts)


def gather_returns_into(row, send, recv, group=None):
    """all_gather_into_tensor of equal shards through caller-owned buffers: `row` [n_local] (any float dtype) is cast into `send`
    [n_local] and gathered into `recv` [world * n_local]; nothing is allocated.  Without a process group `recv` must be
    [n_local] and receives the cast."""
    if dist.is_available() and dist.is_initialized():
        world = dist.get_world_size(group)
        if send.numel() * world != recv.numel() or row.numel() != send.numel():
            raise ValueError("recv must hold world x n_local values (%d x %d != %d)" % (world, send.numel(), recv.numel()))
        send.copy_(row)
        dist.all_gather_into_tensor(recv, send, group=group)
    else:
        if recv.numel() != row.numel():
            raise ValueError("no process group: this rank holds a shard, not the whole batch")
        recv.copy_(row)
    return recv


def local_device(env=None, n_devices=None):
    """The GPU this process should use when none is named: LOCAL_RANK (what torch.distributed.run exports for one process
    per GPU), modulo the number of visible devices; without a launcher, torch's current device.  Every rank defaulting to
    device 0 would put all shards on one GPU - correct results, no scaling."""
    import os
    env = os.environ if env is None else env
    if n_devices is None:
        n_devices = torch.cuda.device_count()
    if "LOCAL_RANK" in env:
        return int(env["LOCAL_RANK"]) % max(int(n_devices), 1)
    return torch.cuda.current_device() if torch.cuda.is_available() else 0


class ShardedSbrOS:
    """This rank's block of a global batch of SBROS-v1 envs (one process per GPU)."""

    def __init__(self, n_global, rank=None, world=None, device=None, **kw):
        from .vec_env import SbrOSVec
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        if world is None:
            world = dist.get_world_size() if dist.is_initialized() else 1
        self.n_global, self.rank, self.world = int(n_global), rank, world
        self.start, self.stop = shard_range(n_global, rank, world)
        self.device = local_device() if device is None else device
        self.env = SbrOSVec(self.stop - self.start, device=self.device, first_env_id=self.start, **kw)

    @property
    def global_ids(self):
        return torch.arange(self.start, self.stop)

    def reset(self, seed=0, scenario_of=lambda gid: gid % 8, **kw):
        return self.env.reset(seed=seed, scenario=scenario_of(self.global_ids).to(torch.int32), **kw)

    def step(self, action):
        return self.env.step(action)

    def rollout(self, n_steps, policy_seed=0):
        return self.env.rollout(n_steps, policy_seed)

    def gather_buffers(self, dtype=torch.float32):
        """Caller-owned buffers for gather_episode_returns_into(): (float64 row [n_local], send [n_local] dtype, recv [n_global]
        dtype) on this rank's device.  Only equal shards can be gathered without staging (all_gather_into_tensor)."""
        if self.n_global % self.world != 0:
            raise ValueError("gather_buffers needs equal shards (n_global % world == 0); use gather_episode_returns()")
        dev, n = self.env.device, self.stop - self.start
        return (torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=dtype, device=dev),
                torch.empty(self.n_global, dtype=dtype, device=dev))

    def gather_episode_returns_into(self, bufs, group=None):
        """The single collective of the path with NOTHING allocated: the returns row is copied into bufs[0] (float64), cast
        into bufs[1] and all-gathered into bufs[2] ([n_global], ordered by global env id), all on the current stream.  Returns
        bufs[2].  Without a process group (world 1) the cast lands in bufs[2] directly."""
        self.env.episode_returns(out=bufs[0])
        return gather_returns_into(*bufs, group=group)

    def gather_episode_returns(self, dtype=torch.float32, group=None):
        """[n_global] episode returns on every rank: the single collective of the path (allocates its result; ragged shards
        are padded).  gather_episode_returns_into() is the allocation-free form for equal shards."""
        return gather_returns(self.env.episode_returns().to(dtype), self.n_global, group=group)

    def close(self):
        self.env.close()
