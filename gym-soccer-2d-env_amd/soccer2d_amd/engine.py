"""Engine: thin owner of one libs2d_hip.so handle plus zero-copy torch views of its arena.

PyTorch is plumbing here (device memory, streams): the arena is ONE torch uint8 tensor that
the C library carves up; every state / output array is a view of it, so nothing is copied
between the engine and PyTorch consumers.  All launches go to torch's current HIP stream.
"""
import ctypes as C

import torch

from . import _capi
from ._capi import (ACT_COMMAND, ACT_CONTINUOUS, ACT_DISCRETE_I32, ACT_DISCRETE_I64, ACT_RANDOM, ACT_TURNING,
                    BUFFER_FIELDS, S2D_OBS_DIM, WORLD_MODEL_FIELDS)

_TORCH_DTYPES = {'float32': torch.float32, 'int32': torch.int32, 'uint8': torch.uint8, 'int64': torch.int64}
_ITEM = {'float32': 4, 'int32': 4, 'uint8': 1, 'int64': 8}

# reach_ball_env.py:26-36 -- same names, same defaults
TASK_KWARGS = dict(change_ball_position=True, change_ball_velocity=False, ball_position_x=0, ball_position_y=0,
                   ball_speed=0, ball_direction=0, min_distance_to_ball=5.0, max_steps=200,
                   use_continuous_action=True, action_space_size=16, use_turning=False)


def make_config(seed=0x5EED, env_id_offset=0, auto_reset=True, noise=True, server_params=None, **kwargs):
    """S2DConfig from ReachBallEnv-style kwargs (+ optional ServerParam overrides by
    idl/service.proto field name).  Unknown names raise ValueError.  noise=True (player_rand / ball_rand on) is the
    default because the reference's rcssserver runs with its stock noise (soccer_2d_env.py:363-368); noise=False is the
    explicit opt-in for deterministic dynamics."""
    lib = _capi.load_library()
    cfg = _capi.S2DConfig()
    lib.s2d_default_config(C.byref(cfg))
    for k, v in kwargs.items():
        if k not in TASK_KWARGS:
            raise ValueError(f"unknown ReachBallEnv kwarg {k!r}")
        setattr(cfg.task, k, type(getattr(cfg.task, k))(v))
    for k, v in (server_params or {}).items():
        if not hasattr(cfg.sp, k):
            raise ValueError(f"unknown ServerParam field {k!r}")
        setattr(cfg.sp, k, float(v))
    cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    cfg.env_id_offset = int(env_id_offset)
    cfg.auto_reset = int(bool(auto_reset))
    cfg.noise = int(bool(noise))
    _capi.check(lib, lib.s2d_validate_config(C.byref(cfg)), 's2d_validate_config')
    return cfg


class Engine:
    """One batched simulator instance on one GPU."""

    def __init__(self, num_envs, device='cuda:0', cfg=None, **kwargs):
        self.lib = _capi.load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("the s2d HIP engine needs a GPU (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise ValueError(f"device must be a cuda/HIP device, got {device!r}")
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.cfg = cfg if cfg is not None else make_config(**kwargs)
        self.num_envs = int(num_envs)
        if self.num_envs <= 0:
            raise ValueError("num_envs must be positive")
        nbytes = self.lib.s2d_arena_bytes(C.byref(self.cfg), self.num_envs)
        if nbytes == 0:
            raise ValueError("s2d_arena_bytes rejected the configuration")
        # +256 so that a 256-byte aligned base always exists inside the tensor
        self._arena_raw = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        shift = (-self._arena_raw.data_ptr()) % 256
        self.arena = self._arena_raw[shift:shift + nbytes]
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.s2d_create(C.byref(self.cfg), self.num_envs, self.device.index, self.arena.data_ptr(),
                                     nbytes, self._stream(), C.byref(h))
        _capi.check(self.lib, rc, 's2d_create')
        self._h = h
        self._make_views(nbytes)
        self._wm = None
        self._ro_cache = {}

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _make_views(self, nbytes):
        n = self.num_envs
        off = (C.c_int64 * 28)()
        _capi.check(self.lib, self.lib.s2d_buffer_offsets(self._h, off, 28), 's2d_buffer_offsets')
        assert off[0] == nbytes
        self.buffers = {}
        for k, (name, _ct, dt, trail) in enumerate(BUFFER_FIELDS):
            o = off[k + 1]
            if trail is None:          # stats: [S2D_STATS_ROWS(n)][8], one row per group of 64 envs
                rows = max(64, (n + 63) // 64)
                count, shape = rows * 8, (rows, 8)
                name = 'stats_striped'
            else:
                count = n
                for d in trail:
                    count *= d
                shape = (n,) + tuple(trail)
            view = self.arena[o:o + count * _ITEM[dt]].view(_TORCH_DTYPES[dt]).view(shape)
            self.buffers[name] = view
            setattr(self, name, view)

    def close(self):
        if getattr(self, '_h', None):
            self.lib.s2d_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ hot path
    def _action_arg(self, actions, leading=None):
        """Validate a caller action tensor and return (tensor_kept_alive, pointer, kind)."""
        t = self.cfg.task
        if actions is None:
            return None, None, ACT_RANDOM
        if not torch.is_tensor(actions):
            actions = torch.as_tensor(actions, device=self.device)
        if actions.device != self.device:
            actions = actions.to(self.device, non_blocking=True)
        lead = (self.num_envs,) if leading is None else (leading, self.num_envs)
        if not t.use_continuous_action:
            if actions.dtype not in (torch.int32, torch.int64):
                actions = actions.to(torch.int64)
            if tuple(actions.shape) == lead + (1,):
                actions = actions.reshape(lead)
            if tuple(actions.shape) != lead:
                raise ValueError(f"discrete actions must have shape {lead}, got {tuple(actions.shape)}")
            kind = ACT_DISCRETE_I32 if actions.dtype == torch.int32 else ACT_DISCRETE_I64
        else:
            width = 4 if t.use_turning else 1
            actions = actions.to(torch.float32)
            if width == 1 and tuple(actions.shape) == lead:
                actions = actions.reshape(lead + (1,))
            if tuple(actions.shape) != lead + (width,):
                raise ValueError(f"continuous actions must have shape {lead + (width,)}, got {tuple(actions.shape)}")
            kind = ACT_TURNING if t.use_turning else ACT_CONTINUOUS
        actions = actions.contiguous()
        return actions, C.c_void_p(actions.data_ptr()), kind

    def reset(self, mask=None):
        """Reset all envs (mask None) or those where mask != 0.  Stream-ordered, async."""
        ptr = None
        if mask is not None:
            mask = torch.as_tensor(mask, device=self.device)
            if tuple(mask.shape) != (self.num_envs,):
                raise ValueError(f"mask must have shape ({self.num_envs},)")
            mask = mask.to(torch.uint8).contiguous()
            ptr = C.c_void_p(mask.data_ptr())
        _capi.check(self.lib, self.lib.s2d_reset(self._h, ptr, self._stream()), 's2d_reset')
        self._keep = mask
        return self.obs

    def step(self, actions=None):
        """One cycle for every env.  actions None = in-kernel uniform random policy."""
        keep, ptr, kind = self._action_arg(actions)
        _capi.check(self.lib, self.lib.s2d_step(self._h, ptr, kind, self._stream()), 's2d_step')
        self._keep = keep
        return self.obs, self.reward, self.done, self.result

    def step_commands(self, commands):
        """One cycle with a decoded body command per env: float32 [N, 4] = (S2D_CMD_* 0 none / 1 dash / 2 turn, power, relative
        direction, 0), executed as it is (S2D_ACT_COMMAND) -- the boundary of the reference's `action_to_rpc_actions` hook."""
        c = torch.as_tensor(commands, device=self.device).to(torch.float32).contiguous()
        if tuple(c.shape) != (self.num_envs, 4):
            raise ValueError(f"commands must have shape ({self.num_envs}, 4), got {tuple(c.shape)}")
        _capi.check(self.lib, self.lib.s2d_step(self._h, C.c_void_p(c.data_ptr()), ACT_COMMAND, self._stream()), 's2d_step')
        self._keep = c
        return self.obs, self.reward, self.done, self.result

    def rollout(self, n_steps, actions=None, out=None, with_obs=True):
        """n_steps fused cycles in one launch.  Returns dict of time-major tensors
        obs [T,N,10], action [T,N]/[T,N,1]/[T,N,4], reward [T,N], done [T,N], result [T,N].
        A caller-owned `out` dict is validated once and its pointer block cached (keyed by the dict's identity and the
        tensors' addresses), so that re-launching into the same buffers costs one ctypes call: at 65 536 envs a launch of
        64 cycles takes ~48 us on the device, and the host must not be the slower side."""
        T, n = int(n_steps), self.num_envs
        keep, ptr, kind = self._action_arg(actions, leading=T)
        fresh = out is None
        if fresh:                  # a buffer nobody will pass again: never cached (the cache holds `out` alive)
            out = self.alloc_rollout(T, with_obs=with_obs)
        cached = None if fresh else self._ro_cache.get(id(out))
        key = tuple(None if out.get(k) is None else out[k].data_ptr() for k in ('obs', 'action', 'reward', 'done', 'result'))
        if cached is None or cached[0] != key or cached[1] < T:
            ro = _capi.S2DRollout()
            t_min = None
            for name in ('obs', 'action', 'reward', 'done', 'result'):
                v = out.get(name)
                if v is not None:
                    if not v.is_contiguous() or v.device != self.device or v.shape[0] < T or v.shape[1] != n:
                        raise ValueError(f"rollout buffer {name!r} must be a contiguous [T>={T},{n},...] tensor on {self.device}")
                    setattr(ro, name, v.data_ptr())
                    t_min = v.shape[0] if t_min is None else min(t_min, v.shape[0])
            cached = (key, T if t_min is None else t_min, ro, out)
            if not fresh:
                if len(self._ro_cache) >= 16:
                    self._ro_cache.clear()
                self._ro_cache[id(out)] = cached
        _capi.check(self.lib, self.lib.s2d_rollout(self._h, T, ptr, kind, C.byref(cached[2]), self._stream()), 's2d_rollout')
        self._keep = (keep, out)
        return out

    def step_k(self, k, actions=None, out=None):
        """k cycles of the per-step API in ONE launch (s2d_step_k; 1 <= k <= 64): for a learner that holds its actions for k steps
        ahead (action repeat, open-loop chunks).  actions [k, N, ...] as for rollout() (None = in-kernel random policy); returns the
        per-step record {obs [k,N,10], action, reward, done, result}; self.obs / reward / done / result hold the last step."""
        k = int(k)
        keep, ptr, kind = self._action_arg(actions, leading=k)
        if out is None:
            out = self.alloc_rollout(k)
        ro = _capi.S2DRollout()
        for name in ('obs', 'action', 'reward', 'done', 'result'):
            v = out.get(name)
            if v is not None:
                if not v.is_contiguous() or v.device != self.device or v.shape[0] < k or v.shape[1] != self.num_envs:
                    raise ValueError(f"record buffer {name!r} must be a contiguous [K>={k},{self.num_envs},...] tensor on {self.device}")
                setattr(ro, name, v.data_ptr())
        _capi.check(self.lib, self.lib.s2d_step_k(self._h, k, ptr, kind, C.byref(ro), self._stream()), 's2d_step_k')
        self._keep = (keep, out)
        return out

    def alloc_rollout(self, T, with_obs=True, slab=False, fields=None):
        """Caller-owned rollout buffers for `rollout(..., out=)`.  slab=True carves the fields out of ONE contiguous
        uint8 tensor (256-byte aligned fields; returned under the key '_slab'), so that a rollout record travels in a
        single collective (dist.all_gather_rollout) without a packing copy: the kernel writes straight into the slab.
        fields (with slab=True): the names that go into the slab -- what travels --; the others are plain local tensors."""
        n, t, dev = self.num_envs, self.cfg.task, self.device
        slab_fields = None if fields is None else set(fields)
        if slab_fields is not None and not slab:
            raise ValueError('fields= selects what goes into the slab: it needs slab=True')
        act_dt, act_trail = (torch.int32, ()) if not t.use_continuous_action else (torch.float32, (4 if t.use_turning else 1,))
        fields = [('obs', torch.float32, (S2D_OBS_DIM,))] if with_obs else []
        fields += [('action', act_dt, act_trail), ('reward', torch.float32, ()), ('done', torch.uint8, ()), ('result', torch.uint8, ())]
        if not slab:
            out = {name: torch.empty((T, n) + trail, dtype=dt, device=dev) for name, dt, trail in fields}
            out.setdefault('obs', None)
            return out
        in_slab = [f for f in fields if slab_fields is None or f[0] in slab_fields]
        local = {name: torch.empty((T, n) + trail, dtype=dt, device=dev) for name, dt, trail in fields if slab_fields is not None and name not in slab_fields}
        layout, off = [], 0
        for name, dt, trail in in_slab:
            cnt = T * n
            for d in trail:
                cnt *= d
            nbytes = cnt * torch.empty((), dtype=dt).element_size()
            layout.append((name, dt, (T, n) + trail, off, nbytes))
            off += (nbytes + 255) // 256 * 256
        raw = torch.empty(off + 256, dtype=torch.uint8, device=dev)
        shift = (-raw.data_ptr()) % 256
        slab_t = raw[shift:shift + off]
        out = {name: slab_t[o:o + nb].view(dt).view(shape) for name, dt, shape, o, nb in layout}
        out.update(local)
        out.setdefault('obs', None)
        out['_slab'], out['_layout'] = slab_t, tuple(layout)
        return out

    # ------------------------------------------------------------------ state access
    def world_model_derived(self):
        """Derived protobuf-mirroring fields (dist/angle members of RpcVector2D etc.)."""
        if self._wm is None:
            self._wm = {k: torch.empty(self.num_envs, dtype=torch.float32, device=self.device)
                        for k in WORLD_MODEL_FIELDS}
        wm = _capi.S2DWorldModel()
        for k, v in self._wm.items():
            setattr(wm, k, v.data_ptr())
        _capi.check(self.lib, self.lib.s2d_world_model(self._h, C.byref(wm), self._stream()), 's2d_world_model')
        return self._wm

    VALIDATE_NAMES = ('non_finite', 'angle_range', 'stamina_range', 'effort_recovery_range', 'counters', 'obs_non_finite')

    def validate_state(self):
        """Debug guard (s2d_validate_state): dict of violation counts; all zero for every state the engine produces."""
        counts = torch.zeros(8, dtype=torch.int32, device=self.device)
        _capi.check(self.lib, self.lib.s2d_validate_state(self._h, C.c_void_p(counts.data_ptr()), self._stream()), 's2d_validate_state')
        c = counts.cpu().tolist()
        return dict(zip(self.VALIDATE_NAMES, c))

    def set_seed(self, seed):
        """New Philox key for all later draws (takes effect at the next launch; ordered on torch's current stream)."""
        _capi.check(self.lib, self.lib.s2d_set_seed(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF, self._stream()), 's2d_set_seed')
        self.cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF

    @property
    def stats(self):
        """int64[8]: env-steps, Goal, Out, Timeout, ... (sum over the device-side stripes)."""
        return self.stats_striped.sum(dim=0)

    def stats_reset(self):
        _capi.check(self.lib, self.lib.s2d_stats_reset(self._h, self._stream()), 's2d_stats_reset')

    def kernel_name(self):
        s = self.lib.s2d_kernel_name(self._h)
        return s.decode() if s else ''

    def state_dict(self):
        """Exact-resume checkpoint: the arena bytes + config (Philox is counter-based, so
        there is no generator state beyond `cycle`, which lives in the arena)."""
        return {'arena': self.arena.clone(), 'num_envs': self.num_envs,
                'config': bytes(memoryview(self.cfg))}

    def load_state_dict(self, sd):
        if sd['num_envs'] != self.num_envs or sd['config'] != bytes(memoryview(self.cfg)):
            raise ValueError("state_dict belongs to a different engine configuration")
        self.arena.copy_(sd['arena'].to(self.device))
