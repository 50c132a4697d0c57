"""ctypes wrapper around the 11v11 match oracle (oracle/_build/libs2d_match_oracle_f32.so).
TEST INFRASTRUCTURE (same rules as tests/oracle.py)."""
import ctypes as C
import os

import numpy as np

import oracle as O
from soccer2d_amd import _capi, _capi_match as M

MATCH_DEFAULTS = dict(
    kick_power_rate=0.027, kickable_margin=0.7, kick_rand=0.1, max_power=100.0, min_power=-100.0,
    tackle_dist=2.0, tackle_back_dist=0.0, tackle_width=1.25, tackle_power_rate=0.027,
    max_tackle_power=100.0, max_back_tackle_power=0.0,
    goal_width=14.02, offside_active_area_size=2.5, free_kick_distance=9.15,
    tackle_cycles=10, half_time_cycles=3000, nr_normal_halfs=2, drop_ball_time=100, use_offside=1, catch_ban_cycle=5,
    catchable_area_l=1.2, catch_area_w=1.0, catch_probability=1.0, max_catch_angle=90.0, min_catch_angle=-90.0,
    penalty_area_length=16.5, penalty_area_half_width=20.16, goalie_max_moves=2, after_goal_wait=50,
    kick_off_wait=0, back_passes=1, free_kick_faults=1, stopped_clock=1, announce_wait=30, foul_cycles=5,
    foul_detect_probability=0.5, nr_extra_halfs=2, extra_half_cycles=1000, golden_goal=0,
    penalty_shoot_outs=1, pen_before_setup_wait=10, pen_ready_wait=10, pen_taken_wait=150, pen_nr_kicks=5, pen_max_extra_kicks=5,
    pen_dist_x=42.5, illegal_defense_number=0, illegal_defense_duration=20, illegal_defense_dist_x=16.5, illegal_defense_width=40.32,
    pen_allow_mult_kicks=1, pen_random_winner=0)


def default_player_type(sp, mp):
    """PlayerType 0 = the ServerParam values (the test suite's own statement of it)."""
    return dict(player_speed_max=sp['player_speed_max'], stamina_inc_max=sp['stamina_inc_max'],
                player_decay=sp['player_decay'], inertia_moment=sp['inertia_moment'],
                dash_power_rate=sp['dash_power_rate'], player_size=sp['player_size'],
                kickable_margin=mp['kickable_margin'], kick_rand=mp['kick_rand'], extra_stamina=sp['extra_stamina'],
                effort_max=sp['effort_init'], effort_min=sp['effort_min'], kick_power_rate=mp['kick_power_rate'],
                catchable_area_l_stretch=1.0)

OBJ_FIELDS = ('x', 'y', 'vx', 'vy', 'body', 'stamina', 'effort', 'recovery', 'stamina_capacity', 'tackle_cycles')
EXTRA_OBJ_FIELDS = {'catch_ban': 22, 'card': 29}      # s2dmo_get field ids beyond the contiguous block
EXTRA_ENV_FIELDS = {'ball_holder': 23, 'goalie_moves': 24, 'set_play_taker': 25, 'last_kicker': 26, 'stopped_cycle': 27, 'tick': 28}
ENV_FIELDS = ('cycle', 'mode', 'mode_side', 'score_left', 'score_right', 'last_touch_side', 'setplay_timer',
              'offside_mask', 'reward_left', 'done', 'nearest_left', 'nearest_right')


def make_match_config(seed=0x5EED, env_id_offset=0, auto_reset=1, noise=0, server=None, player_types=None,
                      player_type_id=None, **mp):
    """player_types: {type id: {field: value}} overrides of the default type; player_type_id: 22 ints."""
    cfg = M.S2DMatchConfig()
    cfg.abi_version = _capi.S2D_ABI_VERSION
    cfg.struct_bytes = C.sizeof(M.S2DMatchConfig)
    sp = dict(O.SERVER_DEFAULTS)
    sp.update(server or {})
    for k, v in sp.items():
        setattr(cfg.sp, k, float(v))
    d = dict(MATCH_DEFAULTS)
    for k in mp:
        if k not in d:
            raise KeyError(k)
    d.update(mp)
    for k, v in d.items():
        setattr(cfg.mp, k, type(getattr(cfg.mp, k))(v))
    cfg.seed, cfg.env_id_offset, cfg.auto_reset, cfg.noise = seed, env_id_offset, auto_reset, noise
    base = default_player_type(sp, d)
    for t in range(M.MATCH_PLAYER_TYPES):
        vals = dict(base)
        vals.update((player_types or {}).get(t, {}))
        for k, v in vals.items():
            setattr(cfg.player_types[t], k, float(v))
    for i in range(22):
        cfg.player_type_id[i] = int(player_type_id[i]) if player_type_id is not None else 0
    return cfg


_lib = None


def lib():
    global _lib
    if _lib is None:
        O.build_oracle()
        path = os.path.join(O.ORACLE_DIR, '_build', 'libs2d_match_oracle_f32.so')
        if not os.path.exists(path):
            import subprocess
            subprocess.run(['make', '-C', O.ORACLE_DIR], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        L = C.CDLL(path)
        cfgp = C.POINTER(M.S2DMatchConfig)
        L.s2dmo_create.argtypes = [cfgp, C.c_int64]; L.s2dmo_create.restype = C.c_void_p
        L.s2dmo_destroy.argtypes = [C.c_void_p]; L.s2dmo_destroy.restype = None
        L.s2dmo_reset.argtypes = [C.c_void_p, C.c_void_p]; L.s2dmo_reset.restype = None
        L.s2dmo_step.argtypes = [C.c_void_p, C.c_void_p]; L.s2dmo_step.restype = None
        L.s2dmo_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]; L.s2dmo_get.restype = C.c_int
        L.s2dmo_set_obj.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_double)]; L.s2dmo_set_obj.restype = C.c_int
        L.s2dmo_set_game.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]; L.s2dmo_set_game.restype = C.c_int
        L.s2dmo_set_touch.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int]; L.s2dmo_set_touch.restype = C.c_int
        L.s2dmo_set_card.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int]; L.s2dmo_set_card.restype = C.c_int
        L.s2dmo_stats.argtypes = [C.c_void_p]; L.s2dmo_stats.restype = C.POINTER(C.c_ulonglong)
        L.s2dmo_random_actions.argtypes = [C.c_void_p, C.c_void_p]; L.s2dmo_random_actions.restype = None
        L.s2dmo_relative.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]; L.s2dmo_relative.restype = None
        _lib = L
    return _lib


class MatchOracle:
    def __init__(self, cfg, n):
        self.L, self.cfg, self.n = lib(), cfg, int(n)
        self.h = self.L.s2dmo_create(C.byref(cfg), self.n)
        assert self.h

    def __del__(self):
        try:
            if self.h:
                self.L.s2dmo_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def reset(self, mask=None):
        if mask is None:
            self.L.s2dmo_reset(self.h, None)
        else:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            self.L.s2dmo_reset(self.h, m.ctypes.data)

    def step(self, actions=None):
        if actions is None:
            self.L.s2dmo_step(self.h, None)
        else:
            a = np.ascontiguousarray(actions, dtype=np.float32)
            assert a.shape == (self.n, 22, 3)
            self.L.s2dmo_step(self.h, a.ctypes.data)

    def random_actions(self):
        a = np.zeros((self.n, 22, 3), dtype=np.float32)
        self.L.s2dmo_random_actions(self.h, a.ctypes.data)
        return a

    def get(self, name):
        if name in OBJ_FIELDS or name in EXTRA_OBJ_FIELDS:
            idx = OBJ_FIELDS.index(name) if name in OBJ_FIELDS else EXTRA_OBJ_FIELDS[name]
            out = np.zeros((self.n, 24))
            assert self.L.s2dmo_get(self.h, idx, out.ctypes.data_as(C.POINTER(C.c_double))) == 0
            return out.astype(np.int32 if name in ('tackle_cycles', 'catch_ban', 'card') else np.float32)
        idx = EXTRA_ENV_FIELDS[name] if name in EXTRA_ENV_FIELDS else 10 + ENV_FIELDS.index(name)
        out = np.zeros(self.n)
        assert self.L.s2dmo_get(self.h, idx, out.ctypes.data_as(C.POINTER(C.c_double))) == 0
        if name == 'reward_left':
            return out.astype(np.float32)
        if name == 'done':
            return out.astype(np.uint8)
        return out.astype(np.int32)

    def stats(self):
        return np.ctypeslib.as_array(self.L.s2dmo_stats(self.h), shape=(8,)).astype(np.int64)

    def relative(self):
        d = np.zeros((self.n, 22, 23), dtype=np.float32)
        a = np.zeros((self.n, 22, 23), dtype=np.float32)
        self.L.s2dmo_relative(self.h, d.ctypes.data, a.ctypes.data)
        return d, a

    def set_obj(self, e, slot, **kw):
        cur = [float(self.get(f)[e, slot]) for f in OBJ_FIELDS]
        for k, v in kw.items():
            cur[OBJ_FIELDS.index(k)] = float(v)
        a = np.asarray(cur, dtype=np.float64)
        assert self.L.s2dmo_set_obj(self.h, e, slot, a.ctypes.data_as(C.POINTER(C.c_double))) == 0

    def set_game(self, e, **kw):
        touch = {k: kw.pop(k) for k in ('set_play_taker', 'last_kicker') if k in kw}
        if touch:
            cur_t = [int(self.get('set_play_taker')[e]), int(self.get('last_kicker')[e])]
            cur_t[0] = int(touch.get('set_play_taker', cur_t[0])); cur_t[1] = int(touch.get('last_kicker', cur_t[1]))
            assert self.L.s2dmo_set_touch(self.h, e, cur_t[0], cur_t[1]) == 0
        names = ENV_FIELDS[:8]
        cur = [int(self.get(f)[e]) for f in names]
        for k, v in kw.items():
            cur[names.index(k)] = int(v)
        a = (C.c_int32 * 8)(*cur)
        assert self.L.s2dmo_set_game(self.h, e, a) == 0
