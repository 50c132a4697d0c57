"""GoToCenter surrogate task on the GPU (include/s2d_gtc.h): the reference's kinematic stand-in
for reach_ball (python_sample_soccer_env.py:46-255) as a batched env.  Device tensors in/out."""
import ctypes as C

import torch

from . import _capi

GTC_OBS_DIM = 4


class S2DGtcConfig(C.Structure):
    _fields_ = [('abi_version', C.c_uint32), ('struct_bytes', C.c_uint32),
                ('x_min', C.c_double), ('x_max', C.c_double), ('y_min', C.c_double), ('y_max', C.c_double),
                ('min_distance_to_center', C.c_double), ('max_steps', C.c_int32), ('continuous', C.c_int32),
                ('seed', C.c_uint64), ('env_id_offset', C.c_int64), ('auto_reset', C.c_int32),
                ('turn', C.c_int32), ('use_turn', C.c_int32), ('actor_out_size', C.c_int32)]


class S2DGtcRollout(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('obs', 'action', 'reward', 'done', 'result')]


GTC_PROTOTYPES = (
    ('s2d_gtc_default_config', None, (C.POINTER(S2DGtcConfig),)),
    ('s2d_gtc_arena_bytes', C.c_size_t, (C.POINTER(S2DGtcConfig), C.c_int64)),
    ('s2d_gtc_create', C.c_int, (C.POINTER(S2DGtcConfig), C.c_int64, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p))),
    ('s2d_gtc_destroy', None, (C.c_void_p,)),
    ('s2d_gtc_buffer_offsets', C.c_int, (C.c_void_p, C.POINTER(C.c_int64), C.c_int)),
    ('s2d_gtc_reset', C.c_int, (C.c_void_p, C.c_void_p, C.c_void_p)),
    ('s2d_gtc_step', C.c_int, (C.c_void_p, C.c_void_p, C.c_void_p)),
    ('s2d_gtc_step_u', C.c_int, (C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)),
    ('s2d_gtc_rollout', C.c_int, (C.c_void_p, C.c_int, C.POINTER(S2DGtcRollout), C.c_void_p)),
)
_FIELDS = (('x', 'float32', ()), ('y', 'float32', ()), ('body', 'float32', ()), ('prev_distance', 'float32', ()),
           ('prev_angle_diff', 'float32', ()), ('step_count', 'int32', ()), ('episode', 'int32', ()),
           ('obs', 'float32', (4,)), ('reward', 'float32', ()), ('done', 'uint8', ()), ('result', 'uint8', ()),
           ('terminal_obs', 'float32', (4,)), ('stats_striped', 'int64', None))
_TD = {'float32': (torch.float32, 4), 'int32': (torch.int32, 4), 'uint8': (torch.uint8, 1), 'int64': (torch.int64, 8)}


def bind(lib):
    for name, res, args in GTC_PROTOTYPES:
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, list(args)
    return lib


def make_gtc_config(seed=0x5EED, env_id_offset=0, auto_reset=True, **kw):
    lib = bind(_capi.load_library())
    cfg = S2DGtcConfig()
    lib.s2d_gtc_default_config(C.byref(cfg))
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise ValueError(f"unknown GoToCenter parameter {k!r}")
        setattr(cfg, k, type(getattr(cfg, k))(v))
    cfg.seed, cfg.env_id_offset, cfg.auto_reset = int(seed), int(env_id_offset), int(bool(auto_reset))
    return cfg


class GoToCenterVecEnv:
    def __init__(self, num_envs, device='cuda:0', cfg=None, **kw):
        self.lib = bind(_capi.load_library())
        if not torch.cuda.is_available():
            raise RuntimeError("the s2d HIP engine needs a GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.device = torch.device(device)
        self.cfg = cfg if cfg is not None else make_gtc_config(**kw)
        self.num_envs = n = int(num_envs)
        nbytes = self.lib.s2d_gtc_arena_bytes(C.byref(self.cfg), n)
        self._raw = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        shift = (-self._raw.data_ptr()) % 256
        self.arena = self._raw[shift:shift + nbytes]
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.s2d_gtc_create(C.byref(self.cfg), n, self.device.index or 0, self.arena.data_ptr(), nbytes, self._stream(), C.byref(h))
        _capi.check(self.lib, rc, 's2d_gtc_create')
        self._h = h
        off = (C.c_int64 * 14)()
        _capi.check(self.lib, self.lib.s2d_gtc_buffer_offsets(self._h, off, 14), 's2d_gtc_buffer_offsets')
        for k, (name, dt, trail) in enumerate(_FIELDS):
            td, item = _TD[dt]
            shape = (64, 8) if trail is None else (n,) + tuple(trail)
            cnt = 1
            for d in shape:
                cnt *= d
            setattr(self, name, self.arena[off[k + 1]:off[k + 1] + cnt * item].view(td).view(shape))
        from .spaces import Box, Discrete
        import numpy as np
        # python_sample_soccer_env.py:70-86
        self.turn_mode = bool(self.cfg.turn and self.cfg.continuous)
        self.action_dim = int(self.cfg.actor_out_size) if self.turn_mode else 1
        if self.turn_mode:
            self.action_space = Box(low=-1.0, high=1.0, shape=(self.action_dim,), dtype=np.float32)
        else:
            self.action_space = Box(low=-1.0, high=1.0, shape=(1,), dtype=np.float32) if self.cfg.continuous else Discrete(16)
        self.observation_space = Box(low=-1.0, high=1.0, shape=(4,), dtype=np.float32)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @property
    def stats(self):
        return self.stats_striped.sum(dim=0)

    def reset(self, mask=None):
        ptr = None
        if mask is not None:
            mask = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
            ptr = C.c_void_p(mask.data_ptr())
        _capi.check(self.lib, self.lib.s2d_gtc_reset(self._h, ptr, self._stream()), 's2d_gtc_reset')
        self._keep = mask
        return self.obs

    def step(self, actions=None, select_u=None):
        """actions: int[N] (discrete) / float[N] (continuous) / float[N, actor_out_size] (turn mode) or None
        (in-engine uniform random policy).  select_u: optional float[N] uniforms for the turn / dash selection
        (python_sample_soccer_env.py:151) in place of the engine's Philox stream."""
        ptr, uptr = None, None
        if actions is not None:
            a = torch.as_tensor(actions, device=self.device)
            if self.cfg.continuous:
                a = a.to(torch.float32).reshape(self.num_envs, self.action_dim).contiguous()
            else:
                a = a.to(torch.int32).reshape(self.num_envs).contiguous()
            ptr, self._keep = C.c_void_p(a.data_ptr()), a
        if select_u is not None:
            u = torch.as_tensor(select_u, device=self.device).to(torch.float32).reshape(self.num_envs).contiguous()
            uptr, self._keep_u = C.c_void_p(u.data_ptr()), u
            _capi.check(self.lib, self.lib.s2d_gtc_step_u(self._h, ptr, uptr, self._stream()), 's2d_gtc_step_u')
        else:
            _capi.check(self.lib, self.lib.s2d_gtc_step(self._h, ptr, self._stream()), 's2d_gtc_step')
        return self.obs, self.reward, self.done, {'result': self.result, 'terminal_observation': self.terminal_obs}

    def rollout(self, n_steps, with_obs=True):
        T, n, dev = int(n_steps), self.num_envs, self.device
        out = dict(obs=torch.empty((T, n, 4), device=dev) if with_obs else None,
                   action=(torch.empty((T, n, self.action_dim) if self.turn_mode else (T, n), dtype=torch.float32, device=dev)
                           if self.cfg.continuous else torch.empty((T, n), dtype=torch.int32, device=dev)),
                   reward=torch.empty((T, n), device=dev), done=torch.empty((T, n), dtype=torch.uint8, device=dev),
                   result=torch.empty((T, n), dtype=torch.uint8, device=dev))
        ro = S2DGtcRollout()
        for k, v in out.items():
            if v is not None:
                setattr(ro, k, v.data_ptr())
        _capi.check(self.lib, self.lib.s2d_gtc_rollout(self._h, T, C.byref(ro), self._stream()), 's2d_gtc_rollout')
        self._keep = out
        return out

    def close(self):
        if getattr(self, '_h', None):
            self.lib.s2d_gtc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
