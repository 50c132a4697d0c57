"""ctypes mirror of include/s2d_match.h (11v11 match engine entry points of libs2d_hip.so)."""
import ctypes as C

from . import _capi

MATCH_PLAYERS, MATCH_SLOTS, MATCH_BALL, MATCH_OBJ_WORDS = 22, 24, 22, 5
MCMD_NONE, MCMD_DASH, MCMD_TURN, MCMD_KICK, MCMD_TACKLE, MCMD_CATCH, MCMD_MOVE = 0, 1, 2, 3, 4, 5, 6
MATCH_PLAYER_TYPES, GOALIE_LEFT, GOALIE_RIGHT = 18, 0, 11
GM_TIME_OVER, GM_PLAY_ON, GM_KICK_OFF, GM_KICK_IN, GM_FREE_KICK, GM_CORNER_KICK, GM_GOAL_KICK, GM_AFTER_GOAL, GM_OFF_SIDE = 1, 2, 3, 4, 5, 6, 7, 8, 9
GM_BEFORE_KICK_OFF, GM_BACK_PASS, GM_FREE_KICK_FAULT = 0, 18, 19          # idl/service.proto:268, 286-287
GM_FIRST_HALF_OVER, GM_FOUL_CHARGE, GM_CATCH_FAULT, GM_IND_FREE_KICK, GM_GOALIE_CATCH, GM_EXTEND_HALF = 11, 14, 20, 21, 30, 31   # :279, 282, 288-289, 298-299
GM_PENALTY_SETUP, GM_PENALTY_READY, GM_PENALTY_TAKEN, GM_PENALTY_MISS, GM_PENALTY_SCORE, GM_PENALTY_ONFIELD, GM_PENALTY_FOUL = 22, 23, 24, 25, 26, 28, 29   # :290-297
GM_FOUL_PUSH, GM_FOUL_MULTIPLE_ATTACKER, GM_FOUL_BALL_OUT = 15, 16, 17   # :283-285 (an operator's calls: played like FoulCharge_)
GM_PAUSE, GM_HUMAN = 12, 13                            # :280-281 (an operator's: written into eng.mode to hold a match)
GM_ILLEGAL_DEFENSE = 27                                # :295 (off in the stock server)
GM_PENALTY_KICK = 10                                   # :278 (a foul inside the offender's own penalty area; the shoot-out modes are not built)
GM_NAMES = {0: 'BeforeKickOff', 1: 'TimeOver', 2: 'PlayOn', 3: 'KickOff_', 4: 'KickIn_', 5: 'FreeKick_', 6: 'CornerKick_', 7: 'GoalKick_',
            8: 'AfterGoal_', 9: 'OffSide_', 11: 'FirstHalfOver', 12: 'Pause', 13: 'Human', 14: 'FoulCharge_', 15: 'FoulPush_', 16: 'FoulMultipleAttacker_', 17: 'FoulBallOut_', 18: 'BackPass_', 19: 'FreeKickFault_',
            20: 'CatchFault_', 21: 'IndFreeKick_', 22: 'PenaltySetup_', 23: 'PenaltyReady_', 24: 'PenaltyTaken_',
            25: 'PenaltyMiss_', 26: 'PenaltyScore_', 27: 'IllegalDefense_', 28: 'PenaltyOnfield_', 29: 'PenaltyFoul_', 30: 'GoalieCatch_', 31: 'ExtendHalf'}
CARD_NONE, CARD_YELLOW, CARD_RED = 0, 1, 2


class S2DMatchParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        'kick_power_rate', 'kickable_margin', 'kick_rand', 'max_power', 'min_power',
        'tackle_dist', 'tackle_back_dist', 'tackle_width', 'tackle_power_rate',
        'max_tackle_power', 'max_back_tackle_power',
        'goal_width', 'offside_active_area_size', 'free_kick_distance')] + [(n, C.c_int32) for n in (
            'tackle_cycles', 'half_time_cycles', 'nr_normal_halfs', 'drop_ball_time', 'use_offside', 'catch_ban_cycle')] + [
                (n, C.c_double) for n in ('catchable_area_l', 'catch_area_w', 'catch_probability', 'max_catch_angle',
                                          'min_catch_angle', 'penalty_area_length', 'penalty_area_half_width')] + [
                    ('goalie_max_moves', C.c_int32), ('after_goal_wait', C.c_int32), ('kick_off_wait', C.c_int32),
                    ('back_passes', C.c_int32), ('free_kick_faults', C.c_int32), ('stopped_clock', C.c_int32),
                    ('announce_wait', C.c_int32), ('foul_cycles', C.c_int32), ('nr_extra_halfs', C.c_int32),
                    ('foul_detect_probability', C.c_double), ('extra_half_cycles', C.c_int32), ('golden_goal', C.c_int32),
                    ('penalty_shoot_outs', C.c_int32), ('pen_before_setup_wait', C.c_int32), ('pen_ready_wait', C.c_int32),
                    ('pen_taken_wait', C.c_int32), ('pen_nr_kicks', C.c_int32), ('pen_max_extra_kicks', C.c_int32),
                    ('pen_dist_x', C.c_double), ('illegal_defense_number', C.c_int32), ('illegal_defense_duration', C.c_int32),
                    ('illegal_defense_dist_x', C.c_double), ('illegal_defense_width', C.c_double),
                    ('pen_allow_mult_kicks', C.c_int32), ('pen_random_winner', C.c_int32)]


PLAYER_TYPE_FIELDS = ('player_speed_max', 'stamina_inc_max', 'player_decay', 'inertia_moment', 'dash_power_rate',
                      'player_size', 'kickable_margin', 'kick_rand', 'extra_stamina', 'effort_max', 'effort_min',
                      'kick_power_rate', 'catchable_area_l_stretch')


class S2DPlayerType(C.Structure):          # idl/service.proto:1697-1732 (members that enter the dynamics)
    _fields_ = [(n, C.c_double) for n in PLAYER_TYPE_FIELDS]


class S2DPlayerParams(C.Structure):        # idl/service.proto:1664-1695
    _fields_ = [(n, C.c_double) for n in (
        'player_speed_max_delta_min', 'player_speed_max_delta_max', 'stamina_inc_max_delta_factor',
        'player_decay_delta_min', 'player_decay_delta_max', 'inertia_moment_delta_factor',
        'dash_power_rate_delta_min', 'dash_power_rate_delta_max', 'player_size_delta_factor',
        'kickable_margin_delta_min', 'kickable_margin_delta_max', 'kick_rand_delta_factor',
        'extra_stamina_delta_min', 'extra_stamina_delta_max', 'effort_max_delta_factor', 'effort_min_delta_factor',
        'new_dash_power_rate_delta_min', 'new_dash_power_rate_delta_max', 'new_stamina_inc_max_delta_factor',
        'kick_power_rate_delta_min', 'kick_power_rate_delta_max',
        'catchable_area_l_stretch_min', 'catchable_area_l_stretch_max')]


class S2DMatchConfig(C.Structure):
    _fields_ = [('abi_version', C.c_uint32), ('struct_bytes', C.c_uint32),
                ('sp', _capi.S2DServerParams), ('mp', S2DMatchParams),
                ('seed', C.c_uint64), ('env_id_offset', C.c_int64),
                ('auto_reset', C.c_int32), ('noise', C.c_int32), ('reserved', C.c_int32 * 4),
                ('player_types', S2DPlayerType * MATCH_PLAYER_TYPES), ('player_type_id', C.c_int32 * MATCH_SLOTS)]


_F, _I, _U8 = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
# (name, ctypes type, torch dtype, per-env trailing shape; None = stats[8])
MATCH_BUFFER_FIELDS = tuple(
    [(n, _F, 'float32', (MATCH_SLOTS,)) for n in ('x', 'y', 'vx', 'vy', 'body', 'stamina', 'effort', 'recovery', 'stamina_capacity')]
    + [('tackle_cycles', _I, 'int32', (MATCH_SLOTS,)), ('catch_ban', _I, 'int32', (MATCH_SLOTS,))]
    + [(n, _I, 'int32', ()) for n in ('cycle', 'mode', 'mode_side', 'score_left', 'score_right', 'last_touch_side',
                                      'setplay_timer', 'offside_mask', 'ball_holder', 'goalie_moves', 'set_play_taker',
                                      'last_kicker', 'stopped_cycle', 'tick')]
    + [('card', _I, 'int32', (MATCH_SLOTS,))]
    + [('reward_left', _F, 'float32', ()), ('done', _U8, 'uint8', ()),
       ('nearest_left', _I, 'int32', ()), ('nearest_right', _I, 'int32', ()),
       ('stats', C.POINTER(C.c_ulonglong), 'int64', None)])


class S2DMatchBuffers(C.Structure):
    _fields_ = [('n_envs', C.c_int64)] + [(n, t) for (n, t, _, _) in MATCH_BUFFER_FIELDS]


class S2DMatchRollout(C.Structure):
    _fields_ = [('obs', C.c_void_p), ('reward', C.c_void_p), ('mode', C.c_void_p), ('done', C.c_void_p)]


MATCH_PROTOTYPES = (
    ('s2d_match_default_config', None, (C.POINTER(S2DMatchConfig),)),
    ('s2d_match_default_player_params', None, (C.POINTER(S2DPlayerParams),)),
    ('s2d_match_generate_player_types', C.c_int, (C.POINTER(S2DMatchConfig), C.POINTER(S2DPlayerParams), C.c_uint64)),
    ('s2d_match_validate_config', C.c_int, (C.POINTER(S2DMatchConfig),)),
    ('s2d_match_arena_bytes', C.c_size_t, (C.POINTER(S2DMatchConfig), C.c_int64)),
    ('s2d_match_create', C.c_int, (C.POINTER(S2DMatchConfig), C.c_int64, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p,
                                   C.POINTER(C.c_void_p))),
    ('s2d_match_destroy', None, (C.c_void_p,)),
    ('s2d_match_buffers', C.c_int, (C.c_void_p, C.POINTER(S2DMatchBuffers))),
    ('s2d_match_buffer_offsets', C.c_int, (C.c_void_p, C.POINTER(C.c_int64), C.c_int)),
    ('s2d_match_reset', C.c_int, (C.c_void_p, C.c_void_p, C.c_void_p)),
    ('s2d_match_step', C.c_int, (C.c_void_p, C.c_void_p, C.c_void_p)),
    ('s2d_match_rollout', C.c_int, (C.c_void_p, C.c_int, C.c_void_p, C.POINTER(S2DMatchRollout), C.c_void_p)),
    ('s2d_match_relative', C.c_int, (C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)),
    ('s2d_match_kernel_name', C.c_char_p, (C.c_void_p,)),
)


def bind(lib):
    """Attach restype/argtypes of the match entry points (idempotent)."""
    for name, res, args in MATCH_PROTOTYPES:
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = list(args)
    return lib
