"""MatchEngine / Soccer2DMatchVecEnv: N lockstep 11v11 matches on one MI355X (include/s2d_match.h).

The reference has no 11v11 task env; what is mirrored is the protobuf schema its agents see
and speak: per-cycle body commands {Dash, Turn, Kick, Tackle} (idl/service.proto:380-402) in,
WorldModel-shaped tensors (ball, teammates/opponents tables, game_mode_type, scores, cycle;
idl/service.proto:144-175, 306-349) out.  Device tensors end to end, zero-copy views of the
engine arena, launches on torch's current stream.  No CPU fallback.
"""
import ctypes as C

import torch

from . import _capi, _capi_match as M

_TD = {'float32': torch.float32, 'int32': torch.int32, 'uint8': torch.uint8, 'int64': torch.int64}
_ITEM = {'float32': 4, 'int32': 4, 'uint8': 1, 'int64': 8}


def make_match_config(seed=0x5EED, env_id_offset=0, auto_reset=True, noise=False, server_params=None,
                      hetero_seed=None, player_type_id=None, player_types=None, **match_params):
    """S2DMatchConfig.  Heterogeneous players: `hetero_seed` draws the 17 non-default PlayerTypes the way
    rcssserver does (s2d_match_generate_player_types); `player_types` = {type id: {field: value}} overrides;
    `player_type_id` = 22 type ids, one per player slot (DoChangePlayerType, idl/service.proto:1393-1433;
    default all 0 = homogeneous)."""
    lib = M.bind(_capi.load_library())
    cfg = M.S2DMatchConfig()
    lib.s2d_match_default_config(C.byref(cfg))
    if server_params or match_params:
        for k, v in (server_params or {}).items():
            if not hasattr(cfg.sp, k):
                raise ValueError(f"unknown ServerParam field {k!r}")
            setattr(cfg.sp, k, float(v))
        for k, v in match_params.items():
            if not hasattr(cfg.mp, k):
                raise ValueError(f"unknown match parameter {k!r}")
            setattr(cfg.mp, k, type(getattr(cfg.mp, k))(v))
        # the default type follows the (possibly overridden) server parameters
        sp, mp = cfg.sp, cfg.mp
        base = dict(player_speed_max=sp.player_speed_max, stamina_inc_max=sp.stamina_inc_max, player_decay=sp.player_decay,
                    inertia_moment=sp.inertia_moment, dash_power_rate=sp.dash_power_rate, player_size=sp.player_size,
                    kickable_margin=mp.kickable_margin, kick_rand=mp.kick_rand, extra_stamina=sp.extra_stamina,
                    effort_max=sp.effort_init, effort_min=sp.effort_min, kick_power_rate=mp.kick_power_rate,
                    catchable_area_l_stretch=1.0)
        for t in range(M.MATCH_PLAYER_TYPES):
            for k, v in base.items():
                setattr(cfg.player_types[t], k, float(v))
    if hetero_seed is not None:
        _capi.check(lib, lib.s2d_match_generate_player_types(C.byref(cfg), None, int(hetero_seed) & 0xFFFFFFFFFFFFFFFF),
                    's2d_match_generate_player_types')
    for t, vals in (player_types or {}).items():
        for k, v in vals.items():
            if k not in M.PLAYER_TYPE_FIELDS:
                raise ValueError(f"unknown PlayerType field {k!r}")
            setattr(cfg.player_types[int(t)], k, float(v))
    if player_type_id is not None:
        ids = [int(v) for v in player_type_id]
        if len(ids) != M.MATCH_PLAYERS:
            raise ValueError("player_type_id needs 22 entries")
        for i, v in enumerate(ids):
            cfg.player_type_id[i] = v
    cfg.seed, cfg.env_id_offset = int(seed) & 0xFFFFFFFFFFFFFFFF, int(env_id_offset)
    cfg.auto_reset, cfg.noise = int(bool(auto_reset)), int(bool(noise))
    _capi.check(lib, lib.s2d_match_validate_config(C.byref(cfg)), 's2d_match_validate_config')
    return cfg


class MatchEngine:
    def __init__(self, num_envs, device='cuda:0', cfg=None, **kwargs):
        self.lib = M.bind(_capi.load_library())
        if not torch.cuda.is_available():
            raise RuntimeError("the s2d HIP engine needs a GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.cfg = cfg if cfg is not None else make_match_config(**kwargs)
        self.num_envs = int(num_envs)
        if self.num_envs <= 0:
            raise ValueError("num_envs must be positive")
        nbytes = self.lib.s2d_match_arena_bytes(C.byref(self.cfg), self.num_envs)
        self._raw = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        shift = (-self._raw.data_ptr()) % 256
        self.arena = self._raw[shift:shift + nbytes]
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.s2d_match_create(C.byref(self.cfg), self.num_envs, self.device.index, self.arena.data_ptr(), nbytes,
                                           self._stream(), C.byref(h))
        _capi.check(self.lib, rc, 's2d_match_create')
        self._h = h
        off = (C.c_int64 * 32)()
        _capi.check(self.lib, self.lib.s2d_match_buffer_offsets(self._h, off, 32), 's2d_match_buffer_offsets')
        n = self.num_envs
        for k, (name, _ct, dt, trail) in enumerate(M.MATCH_BUFFER_FIELDS):
            o = off[k + 1]
            shape = (64, 8) if trail is None else (n,) + tuple(trail)
            count = 1
            for d in shape:
                count *= d
            setattr(self, 'stats_striped' if trail is None else name,
                    self.arena[o:o + count * _ITEM[dt]].view(_TD[dt]).view(shape))

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @property
    def stats(self):
        """int64[8]: env-steps, goals left/right, matches, kicks, tackles, offsides, ball-outs."""
        return self.stats_striped.sum(dim=0)

    def close(self):
        if getattr(self, '_h', None):
            self.lib.s2d_match_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _actions(self, actions, T=None):
        if actions is None:
            return None, None
        a = torch.as_tensor(actions, device=self.device).to(torch.float32).contiguous()
        want = (self.num_envs, M.MATCH_PLAYERS, 3) if T is None else (T, self.num_envs, M.MATCH_PLAYERS, 3)
        if tuple(a.shape) != want:
            raise ValueError(f"actions must have shape {want}, got {tuple(a.shape)}")
        return a, C.c_void_p(a.data_ptr())

    def reset(self, mask=None):
        ptr = None
        if mask is not None:
            mask = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
            if tuple(mask.shape) != (self.num_envs,):
                raise ValueError(f"mask must have shape ({self.num_envs},)")
            ptr = C.c_void_p(mask.data_ptr())
        _capi.check(self.lib, self.lib.s2d_match_reset(self._h, ptr, self._stream()), 's2d_match_reset')
        self._keep = mask

    def step(self, actions=None):
        """actions float[N,22,3] = (command, a, b) per player, None = random policy."""
        keep, ptr = self._actions(actions)
        _capi.check(self.lib, self.lib.s2d_match_step(self._h, ptr, self._stream()), 's2d_match_step')
        self._keep = keep
        return self.reward_left, self.done

    def alloc_rollout(self, T, with_obs=True):
        n, dev = self.num_envs, self.device
        return dict(obs=torch.empty((T, n, M.MATCH_SLOTS, M.MATCH_OBJ_WORDS), dtype=torch.float32, device=dev) if with_obs else None,
                    reward=torch.empty((T, n), dtype=torch.float32, device=dev),
                    mode=torch.empty((T, n), dtype=torch.int32, device=dev),
                    done=torch.empty((T, n), dtype=torch.uint8, device=dev))

    def rollout(self, n_steps, actions=None, out=None, with_obs=True):
        T = int(n_steps)
        keep, ptr = self._actions(actions, T)
        if out is None:
            out = self.alloc_rollout(T, with_obs)
        ro = M.S2DMatchRollout()
        for name in ('obs', 'reward', 'mode', 'done'):
            v = out.get(name)
            if v is not None:
                if not v.is_contiguous() or v.shape[0] < T or v.shape[1] != self.num_envs:
                    raise ValueError(f"rollout buffer {name!r} must be contiguous [T>={T},{self.num_envs},...]")
                setattr(ro, name, v.data_ptr())
        _capi.check(self.lib, self.lib.s2d_match_rollout(self._h, T, ptr, C.byref(ro), self._stream()), 's2d_match_rollout')
        self._keep = (keep, out)
        return out

    def kernel_name(self):
        """which instantiation of the cycle kernel this engine launches (`<stock>`: rules and physics folded into the code)"""
        return self.lib.s2d_match_kernel_name(self._h).decode()

    def relative_tables(self):
        """(dist, angle) float32 [N,22,23]: Player.dist_from_self / angle_from_self (and the ball's, column 22)
        as seen by each of the 22 agents (idl/service.proto:84-85, 155-156)."""
        if getattr(self, '_rel', None) is None:
            shape = (self.num_envs, M.MATCH_PLAYERS, M.MATCH_BALL + 1)
            self._rel = (torch.empty(shape, dtype=torch.float32, device=self.device),
                         torch.empty(shape, dtype=torch.float32, device=self.device))
        d, a = self._rel
        _capi.check(self.lib, self.lib.s2d_match_relative(self._h, d.data_ptr(), a.data_ptr(), self._stream()), 's2d_match_relative')
        return d, a

    def egocentric_tables(self):
        """Per-agent view for policies that act from the player's own frame: dict of device tensors
        dist [N,22,23], bearing [N,22,23] (direction of object j seen from agent p RELATIVE to p's body,
        degrees in (-180, 180]; 0 on the diagonal), teammate [22,23] bool (same side; the ball column is
        False).  Built from the relative-tables kernel (Player.dist_from_self / angle_from_self) and the body
        directions; nothing leaves the device."""
        d, a = self.relative_tables()
        body = self.body[:, :M.MATCH_PLAYERS].unsqueeze(2)
        bearing = a - body
        bearing = torch.where(bearing > 180.0, bearing - 360.0, torch.where(bearing <= -180.0, bearing + 360.0, bearing))
        eye = torch.eye(M.MATCH_PLAYERS, M.MATCH_BALL + 1, dtype=torch.bool, device=self.device)
        bearing = torch.where(eye.unsqueeze(0), torch.zeros_like(bearing), bearing)
        side = torch.arange(M.MATCH_BALL + 1, device=self.device) < 11
        teammate = (side[:M.MATCH_PLAYERS, None] == side[None, :])
        teammate[:, M.MATCH_BALL] = False
        return {'dist': d, 'bearing': bearing, 'teammate': teammate}

    def world_model(self):
        """dict proto-path -> device tensor (left team's point of view = absolute coordinates)."""
        P = M.MATCH_PLAYERS
        wm = {'world_model.cycle': self.cycle, 'world_model.stoped_cycle': self.stopped_cycle,
              'world_model.game_mode_type': self.mode, 'world_model.game_mode_side': self.mode_side,
              'world_model.left_team_score': self.score_left, 'world_model.right_team_score': self.score_right,
              'world_model.last_kick_side': self.last_touch_side,
              'world_model.ball.position.x': self.x[:, M.MATCH_BALL], 'world_model.ball.position.y': self.y[:, M.MATCH_BALL],
              'world_model.ball.velocity.x': self.vx[:, M.MATCH_BALL], 'world_model.ball.velocity.y': self.vy[:, M.MATCH_BALL]}
        # the penalty shoot-out (WorldModel.is_penalty_kick_mode, PenaltyKickState: idl/service.proto:336, 130-138), decoded from the
        # set-play word the engine keeps it in (include/s2d_match.h); "our" = the left team
        pen = (self.mode >= M.GM_PENALTY_SETUP) & (self.mode <= 29)
        w = torch.where(pen, self.set_play_taker, torch.zeros_like(self.set_play_taker))
        wm.update({'world_model.is_penalty_kick_mode': pen,
                   'world_model.penalty_kick_state.on_field_side': torch.where(pen, 2, 0),
                   'world_model.penalty_kick_state.current_taker_side': torch.where(pen & (self.mode != M.GM_PENALTY_ONFIELD), self.mode_side, 0),
                   'world_model.penalty_kick_state.our_taker_counter': (w >> 12) & 15,
                   'world_model.penalty_kick_state.their_taker_counter': (w >> 16) & 15,
                   'world_model.penalty_kick_state.our_score': (w >> 20) & 15,
                   'world_model.penalty_kick_state.their_score': (w >> 24) & 15})
        for team, sl in (('teammates', slice(0, 11)), ('opponents', slice(11, P))):
            for f, t in (('position.x', self.x), ('position.y', self.y), ('velocity.x', self.vx), ('velocity.y', self.vy),
                         ('body_direction', self.body), ('stamina', self.stamina), ('is_tackling', self.tackle_cycles)):
                wm[f'world_model.{team}.{f}'] = t[:, sl]
        return wm


class Soccer2DMatchVecEnv:
    """gym-style surface over MatchEngine for BASELINE.json configs[3] ("11v11 full-match env"):
    ``reset() -> obs``, ``step(actions) -> (obs, reward, done, info)`` with device tensors.

    obs     float32 [N, 23, 5]  (x, y, vx, vy, body) of the 22 players and the ball (row 22), absolute
            coordinates (left team attacks +x); a zero-copy view of the engine's last rollout row.
    actions float32 [N, 22, 3]  (command, a, b) per player, commands of include/s2d_match.h.
    reward  float32 [N]         +1 when the left team scores, -1 when the right team scores
            (zero-sum: the right team's reward is the negative).
    done    uint8 [N]           1 when a match reached TimeOver (auto-restart follows the VecEnv convention).
    info    dict of tensors     game_mode_type, game_mode_side, scores, cycle, nearest player per team.
    """

    def __init__(self, num_envs, device='cuda:0', **kwargs):
        import numpy as np
        from .spaces import Box
        self.engine = MatchEngine(num_envs, device, **kwargs)
        self.num_envs, self.device = self.engine.num_envs, self.engine.device
        self.observation_space = Box(low=-200.0, high=200.0, shape=(23, 5), dtype=np.float32)
        self.action_space = Box(low=-180.0, high=180.0, shape=(22, 3), dtype=np.float32)
        self._ro = self.engine.alloc_rollout(1)

    def _obs(self):
        e = self.engine
        return torch.stack([e.x[:, :23], e.y[:, :23], e.vx[:, :23], e.vy[:, :23], e.body[:, :23]], dim=2)

    def reset(self, mask=None):
        self.engine.reset(mask)
        return self._obs()

    def step(self, actions=None):
        a = None if actions is None else torch.as_tensor(actions, device=self.device).to(torch.float32).reshape(1, self.num_envs, 22, 3)
        self.engine.rollout(1, actions=a, out=self._ro)
        e = self.engine
        info = {'game_mode_type': e.mode, 'game_mode_side': e.mode_side, 'left_team_score': e.score_left,
                'right_team_score': e.score_right, 'cycle': e.cycle, 'nearest_left': e.nearest_left, 'nearest_right': e.nearest_right}
        return self._ro['obs'][0, :, :23], e.reward_left, e.done, info

    def close(self):
        self.engine.close()
