"""Multi-GPU plumbing: one process per GPU, contiguous global env-id ranges,
NO per-step communication.  Because every Philox counter carries the GLOBAL env id, the
trajectories of env g are identical for every world size / shard layout.

The only collective on the path is the all-gather of rollout buffers for league-style
self-play (BASELINE.json configs[4]); it is a torch.distributed call (backend "nccl" = RCCL
over xGMI on the GPUs, "gloo" in the CPU tests) issued on a side stream so that it overlaps
the next rollout's simulation.
"""
import torch


def shard_range(n_global, rank, world):
    """Contiguous shard [offset, offset+count) of rank; sizes differ by at most one."""
    if not (0 <= rank < world) or n_global < world:
        raise ValueError('need 0 <= rank < world <= n_global')
    base, rem = divmod(n_global, world)
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def make_sharded_vec_env(n_global, rank, world, device=None, **kwargs):
    """The local shard of a global batch of n_global envs as a Soccer2DVecEnv."""
    from .vec_env import Soccer2DVecEnv
    offset, count = shard_range(n_global, rank, world)
    if device is None:
        device = f'cuda:{rank % max(1, torch.cuda.device_count())}'
    return Soccer2DVecEnv(count, device=device, env_id_offset=offset, **kwargs)


def all_gather_rollout(rollout, group=None, time_major=True):
    """All-gather a local rollout dict {name: tensor[T, N_local, ...]} over the env axis.

    Every rank must hold the same N_local (use equal shards for league play).  Returns
    {name: tensor[T, world*N_local, ...]} if time_major (one permute copy) else the raw
    gathered slabs {name: tensor[world, T, N_local, ...]} (no copy after the collective).
    One collective per field; fields are small in number and large in bytes, which is the
    shape direct xGMI all-gathers want (7 links x ~153 GB/s per GPU)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):      # single process: the local shard is the whole batch
        # (copies, like the collective: the caller may reuse its rollout buffers)
        return {k: (v.clone() if time_major else v.clone().unsqueeze(0)) for k, v in rollout.items() if v is not None}
    world = dist.get_world_size(group)
    out = {}
    for name, t in rollout.items():
        if t is None:
            continue
        t = t.contiguous()
        slab = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(slab.view(-1), t.view(-1), group=group)
        if time_major:
            T, n = t.shape[0], t.shape[1]
            slab = slab.movedim(0, 1).reshape((T, world * n) + tuple(t.shape[2:]))
        out[name] = slab
    return out


def all_reduce_stats(stats, group=None):
    """Sum the int64 episode counters [env_steps, Goal, Out, Timeout, ...] over ranks."""
    import torch.distributed as dist
    s = stats.clone()
    dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
    return s


class LeagueRolloutExchange:
    """Double-buffered rollout exchange for league self-play: while rollout k+1 is being
    simulated on the compute stream, rollout k is all-gathered on a side stream."""

    def __init__(self, env, n_steps, group=None):
        self.env, self.T, self.group = env, int(n_steps), group
        self.bufs = [env.engine.alloc_rollout(self.T), env.engine.alloc_rollout(self.T)]
        self.side = torch.cuda.Stream(env.device)
        self.k = 0
        self._pending = None

    def step(self, actions=None):
        """Simulate T cycles into buffer k, start gathering it, return the PREVIOUS gathered
        rollout (None on the first call)."""
        buf = self.bufs[self.k & 1]
        cur = torch.cuda.current_stream(self.env.device)
        self.env.engine.rollout(self.T, actions=actions, out=buf)
        ready = torch.cuda.Event()
        ready.record(cur)
        prev = self._pending
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            gathered = all_gather_rollout(buf, group=self.group, time_major=False)
            done = torch.cuda.Event()
            done.record(self.side)
        self._pending = (gathered, done)
        self.k += 1
        if prev is None:
            return None
        prev[1].synchronize()
        return prev[0]

    def flush(self):
        if self._pending is None:
            return None
        g, ev = self._pending
        ev.synchronize()
        self._pending = None
        return g
