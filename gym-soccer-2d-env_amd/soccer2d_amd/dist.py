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


FIELDS = ('obs', 'action', 'reward', 'done', 'result')


LEAGUE_FIELDS = ('action', 'reward', 'done', 'result')     # what a league needs of the others' rollouts: 10 of the record's 50 bytes


def _pack(rollout, fields=None):
    """One contiguous uint8 slab holding the chosen fields of a rollout dict (+ its layout); fields=None: every field.
    Slab-backed rollouts (Engine.alloc_rollout(slab=True, fields=...)) are used as they are -- the slab holds what it was carved
    for --; others are packed with one copy."""
    if rollout.get('_slab') is not None:
        if fields is not None and tuple(n for n, *_ in rollout['_layout']) != tuple(f for f in FIELDS if f in fields):
            raise ValueError(f"this slab was carved for {[n for n, *_ in rollout['_layout']]}, not for {list(fields)}")
        return rollout['_slab'], rollout['_layout']
    layout, off = [], 0
    for name in FIELDS:
        t = rollout.get(name)
        if t is None or (fields is not None and name not in fields):
            continue
        nb = t.numel() * t.element_size()
        layout.append((name, t.dtype, tuple(t.shape), off, nb))
        off += (nb + 255) // 256 * 256
    some = next(v for k, v in rollout.items() if k in FIELDS and v is not None)
    slab = torch.empty(off, dtype=torch.uint8, device=some.device)
    for name, dt, shape, o, nb in layout:
        slab[o:o + nb].copy_(rollout[name].contiguous().view(-1).view(torch.uint8))
    return slab, tuple(layout)


def _views(gathered, layout, world, time_major):
    """{name: tensor} views of a gathered slab [world, slab_bytes]."""
    out = {}
    for name, dt, shape, o, nb in layout:
        v = gathered[:, o:o + nb].view(dt).view((world,) + tuple(shape))       # [world, T, N_local, ...]
        if time_major:
            T, n = shape[0], shape[1]
            v = v.movedim(0, 1).reshape((T, world * n) + tuple(shape[2:]))     # one permute copy
        out[name] = v
    return out


def all_gather_rollout(rollout, group=None, time_major=True, out=None, fields=None):
    """All-gather a local rollout dict {name: tensor[T, N_local, ...]} over the env axis with ONE collective.
    `fields`: the subset of the record that travels (None = all five).  The full record is 50 B per env-step -- at T = 256 and
    65 536 envs per rank 838.9 MB out of every rank per exchange, 5.5 ms of a 153 GB/s xGMI link against a 0.18 ms rollout: a
    league that gathers whole records is link-bound at ~1/30 of the simulation rate.  LEAGUE_FIELDS (action, reward, done, result:
    10 B per env-step) is what rating and opponent sampling need; observations stay on the rank that trains on them.

    The five fields of a rollout record (obs f32, action i32 / f32, reward f32, done u8, result u8) are carved out of one
    contiguous uint8 slab (Engine.alloc_rollout(slab=True): the rollout kernel writes into it directly, no packing copy)
    and the slab travels in a single `all_gather_into_tensor` -- on the GPUs RCCL over xGMI; every rank holds the same
    N_local (equal shards for league play).  Per exchange and rank: T * N_local * 50 bytes sent (+ < 1.3 KB of alignment
    padding), world times that received: 209.7 MB out / 1.68 GB in at T = 64, N_local = 65 536, world = 8 -- one large
    message per peer, the shape direct xGMI all-gathers want (7 links x ~153 GB/s per GPU); round 1 issued five
    collectives per exchange, two of them 4 MB slabs where launch latency dominates.
    Returns {name: tensor[T, world*N_local, ...]} if time_major (one permute copy per field) else the raw gathered views
    {name: tensor[world, T, N_local, ...]} (no copy after the collective).  `out`: optional preallocated uint8 tensor
    [world, slab_bytes] to gather into (LeagueRolloutExchange double-buffers it)."""
    import torch.distributed as dist
    slab, layout = _pack(rollout, fields)
    if not (dist.is_available() and dist.is_initialized()):      # single process: the local shard is the whole batch
        g = slab.clone().unsqueeze(0) if out is None else out    # (copies, like the collective: the caller may reuse its buffers)
        if out is not None:
            out[0].copy_(slab)
        return _views(g, layout, 1, time_major)
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world, slab.numel()), dtype=torch.uint8, device=slab.device)
    if slab.is_cuda and dist.get_backend(group) != 'nccl':
        # rehearsal only (several ranks on one GPU, gloo): gloo gathers host memory, so the slab is staged through the host.
        # On the GPUs of a node the backend is "nccl" (= RCCL) and the slab travels device to device over xGMI, below.
        host = torch.empty((world, slab.numel()), dtype=torch.uint8)
        dist.all_gather_into_tensor(host.view(-1), slab.cpu(), group=group)
        out.copy_(host)
        return _views(out, layout, world, time_major)
    dist.all_gather_into_tensor(out.view(-1), slab, group=group)
    return _views(out, layout, world, time_major)


def all_reduce_stats(stats, group=None):
    """Sum the int64 episode counters [env_steps, Goal, Out, Timeout, ...] over ranks."""
    import torch.distributed as dist
    s = stats.clone()
    dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
    return s


class LeagueRolloutExchange:
    """Double-buffered rollout exchange for league self-play: while rollout k+1 is being simulated on the compute stream,
    rollout k is all-gathered (one collective) on a side stream.  Local rollout slabs AND gathered slabs are allocated
    once, up front, two of each: nothing is allocated on the side stream, so the caching allocator never hands a block that
    compute-stream kernels still read to the next gather."""

    def __init__(self, env, n_steps, group=None, timing=False, fields=None):
        """fields: the part of the record that is gathered (None = all five arrays; dist.LEAGUE_FIELDS = action, reward, done,
        result).  The kernel always writes the whole record; arrays outside `fields` stay local tensors of the same buffers."""
        import torch.distributed as dist
        self.env, self.T, self.group = env, int(n_steps), group
        self.fields = None if fields is None else tuple(f for f in FIELDS if f in fields)
        self.timings = [] if timing else None                 # (start, end) HIP events of every gather, on the side stream
        self.bufs = [env.engine.alloc_rollout(self.T, slab=True, fields=self.fields) for _ in range(2)]
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        nbytes = self.bufs[0]['_slab'].numel()
        self.gathered = [torch.empty((self.world, nbytes), dtype=torch.uint8, device=env.device) for _ in range(2)]
        self.bytes_per_exchange = {'sent': nbytes, 'received': self.world * nbytes, 'collectives': 1}
        self.side = torch.cuda.Stream(env.device)
        self.k = 0
        self._pending = None

    def step(self, actions=None):
        """Simulate T cycles into buffer k, start gathering it, return the PREVIOUS gathered rollout (None on the first
        call): {name: tensor[world, T, N_local, ...]} views of a gathered slab that stays valid until the call after next."""
        b = self.k & 1
        buf = self.bufs[b]
        cur = torch.cuda.current_stream(self.env.device)
        self.env.engine.rollout(self.T, actions=actions, out=buf)
        ready = torch.cuda.Event()
        ready.record(cur)
        prev = self._pending
        with torch.cuda.stream(self.side):
            # `ready` also orders this gather behind every read the caller queued on the compute stream from gathered[b],
            # which was handed out two calls ago
            self.side.wait_event(ready)
            if self.timings is not None:
                t0 = torch.cuda.Event(enable_timing=True)
                t0.record(self.side)
            gathered = all_gather_rollout(buf, group=self.group, time_major=False, out=self.gathered[b])
            done = torch.cuda.Event(enable_timing=self.timings is not None)
            done.record(self.side)
            if self.timings is not None:
                self.timings.append((t0, done))
        self._pending = (gathered, done, b)
        self.k += 1
        if prev is None:
            return None
        return self._hand_out(prev, cur)

    def _hand_out(self, pending, cur):
        g, ev, b = pending
        cur.wait_event(ev)                                    # consumers on the compute stream see the finished gather
        return g

    def flush(self):
        if self._pending is None:
            return None
        cur = torch.cuda.current_stream(self.env.device)
        g = self._hand_out(self._pending, cur)
        self._pending = None
        return g
