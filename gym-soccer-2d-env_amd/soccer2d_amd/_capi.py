"""ctypes mirror of include/s2d.h (the C ABI of libs2d_hip.so) and the loader.

No compute happens here: structs, prototypes and error translation only.  The library is
loaded AFTER ``import torch`` so that its ``libamdhip64.so.7`` dependency resolves to the
HIP runtime torch already mapped (one runtime per process: torch's streams and device
memory are then valid in the engine).

The product path has no CPU fallback: if the HIP library is missing or does not load,
``load_library()`` raises ``S2DLibraryError``.
"""
import ctypes as C
import os

S2D_ABI_VERSION = 4
S2D_OBS_DIM = 10

# error codes
S2D_OK, S2D_EINVAL, S2D_EHIP, S2D_ENOMEM, S2D_ENODEV = 0, -1, -2, -3, -4
# enums (include/s2d.h)
MODE_BEFORE_KICK_OFF, MODE_TIME_OVER, MODE_PLAY_ON = 0, 1, 2
SIDE_UNKNOWN, SIDE_LEFT, SIDE_RIGHT = 0, 1, 2
RESULT_NONE, RESULT_GOAL, RESULT_OUT, RESULT_TIMEOUT = 0, 1, 2, 3
RESULT_NAMES = (None, 'Goal', 'Out', 'Timeout')   # info['result'], reach_ball_env.py:126-150
CMD_NONE, CMD_DASH, CMD_TURN, CMD_FREEZE = 0, 1, 2, -1
ACT_DISCRETE_I32, ACT_DISCRETE_I64, ACT_CONTINUOUS, ACT_TURNING, ACT_RANDOM, ACT_COMMAND = 0, 1, 2, 3, 4, 5


class S2DLibraryError(RuntimeError):
    """libs2d_hip.so is missing / failed to load.  There is no fallback path."""


class S2DServerParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        'pitch_half_length', 'pitch_half_width',
        'player_size', 'player_decay', 'player_rand', 'player_speed_max', 'player_accel_max',
        'inertia_moment',
        'stamina_max', 'stamina_inc_max', 'stamina_capacity', 'extra_stamina',
        'recover_init', 'recover_dec_thr', 'recover_min', 'recover_dec',
        'effort_init', 'effort_dec_thr', 'effort_min', 'effort_dec', 'effort_inc_thr', 'effort_inc',
        'dash_power_rate', 'max_dash_power', 'min_dash_power',
        'max_dash_angle', 'min_dash_angle', 'dash_angle_step', 'side_dash_rate', 'back_dash_rate',
        'max_moment', 'min_moment',
        'ball_size', 'ball_decay', 'ball_rand', 'ball_speed_max', 'ball_accel_max',
        'collision_vel_rate')]


class S2DReachBallParams(C.Structure):
    _fields_ = [
        ('change_ball_position', C.c_int32), ('change_ball_velocity', C.c_int32),
        ('ball_position_x', C.c_double), ('ball_position_y', C.c_double),
        ('ball_speed', C.c_double), ('ball_direction', C.c_double),
        ('min_distance_to_ball', C.c_double),
        ('max_steps', C.c_int32), ('use_continuous_action', C.c_int32),
        ('action_space_size', C.c_int32), ('use_turning', C.c_int32),
        ('reset_ball_decay', C.c_double)]


class S2DConfig(C.Structure):
    _fields_ = [
        ('abi_version', C.c_uint32), ('struct_bytes', C.c_uint32),
        ('sp', S2DServerParams), ('task', S2DReachBallParams),
        ('seed', C.c_uint64), ('env_id_offset', C.c_int64),
        ('auto_reset', C.c_int32), ('noise', C.c_int32), ('reserved', C.c_int32 * 4)]


_F = C.POINTER(C.c_float)
_U8 = C.POINTER(C.c_uint8)
_I32 = C.POINTER(C.c_int32)

# (name, ctypes pointer type, torch dtype name, trailing dims) in S2DBuffers order
BUFFER_FIELDS = (
    ('player_x', _F, 'float32', ()), ('player_y', _F, 'float32', ()),
    ('player_vx', _F, 'float32', ()), ('player_vy', _F, 'float32', ()),
    ('player_body', _F, 'float32', ()),
    ('stamina', _F, 'float32', ()), ('effort', _F, 'float32', ()),
    ('recovery', _F, 'float32', ()), ('stamina_capacity', _F, 'float32', ()),
    ('ball_x', _F, 'float32', ()), ('ball_y', _F, 'float32', ()),
    ('ball_vx', _F, 'float32', ()), ('ball_vy', _F, 'float32', ()),
    ('prev_dist', _F, 'float32', ()), ('prev_angle', _F, 'float32', ()),
    ('step_number', _I32, 'int32', ()), ('cycle', _I32, 'int32', ()),
    ('policy_step', _I32, 'int32', ()), ('episode', _I32, 'int32', ()),
    ('obs', _F, 'float32', (S2D_OBS_DIM,)), ('reward', _F, 'float32', ()),
    ('done', _U8, 'uint8', ()), ('result', _U8, 'uint8', ()),
    ('terminal_obs', _F, 'float32', (S2D_OBS_DIM,)),
    ('action_dir', _F, 'float32', ()), ('action_cmd', _U8, 'uint8', ()),
    ('stats', C.POINTER(C.c_ulonglong), 'int64', None),   # [8], not per-env
)
STATE_FIELDS = tuple(f[0] for f in BUFFER_FIELDS[:19])


class S2DBuffers(C.Structure):
    _fields_ = [('n_envs', C.c_int64)] + [(n, t) for (n, t, _, _) in BUFFER_FIELDS]


class S2DRollout(C.Structure):
    _fields_ = [('obs', C.c_void_p), ('action', C.c_void_p), ('reward', C.c_void_p),
                ('done', C.c_void_p), ('result', C.c_void_p)]


WORLD_MODEL_FIELDS = (
    'ball_dist_from_self', 'ball_angle_from_self', 'ball_relative_x', 'ball_relative_y',
    'ball_pos_dist', 'ball_pos_angle', 'ball_vel_dist', 'ball_vel_angle',
    'self_pos_dist', 'self_pos_angle', 'self_vel_dist', 'self_vel_angle',
    'self_dist_from_ball', 'self_angle_from_ball')


class S2DWorldModel(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in WORLD_MODEL_FIELDS]


# every symbol include/s2d.h declares: (name, restype, argtypes)
PROTOTYPES = (
    ('s2d_version', C.c_char_p, ()),
    ('s2d_last_error', C.c_char_p, ()),
    ('s2d_default_config', None, (C.POINTER(S2DConfig),)),
    ('s2d_validate_config', C.c_int, (C.POINTER(S2DConfig),)),
    ('s2d_arena_bytes', C.c_size_t, (C.POINTER(S2DConfig), C.c_int64)),
    ('s2d_create', C.c_int, (C.POINTER(S2DConfig), C.c_int64, C.c_int, C.c_void_p, C.c_size_t,
                             C.c_void_p, C.POINTER(C.c_void_p))),
    ('s2d_destroy', None, (C.c_void_p,)),
    ('s2d_buffers', C.c_int, (C.c_void_p, C.POINTER(S2DBuffers))),
    ('s2d_buffer_offsets', C.c_int, (C.c_void_p, C.POINTER(C.c_int64), C.c_int)),
    ('s2d_reset', C.c_int, (C.c_void_p, C.c_void_p, C.c_void_p)),
    ('s2d_step', C.c_int, (C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)),
    ('s2d_rollout', C.c_int, (C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(S2DRollout), C.c_void_p)),
    ('s2d_step_k', C.c_int, (C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(S2DRollout), C.c_void_p)),
    ('s2d_world_model', C.c_int, (C.c_void_p, C.POINTER(S2DWorldModel), C.c_void_p)),
    ('s2d_stats_reset', C.c_int, (C.c_void_p, C.c_void_p)),
    ('s2d_kernel_name', C.c_char_p, (C.c_void_p,)),
    ('s2d_validate_state', C.c_int, (C.c_void_p, C.c_void_p, C.c_void_p)),
    ('s2d_set_seed', C.c_int, (C.c_void_p, C.c_uint64, C.c_void_p)),
    ('s2d_debug_eval', C.c_int, (C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)),
)

PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(PKG_ROOT, 'lib', 'libs2d_hip.so')

_lib = None


def load_library(path=None):
    """Load libs2d_hip.so and bind every prototype.  Raises S2DLibraryError on failure."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    import torch  # noqa: F401  -- maps torch's libamdhip64.so.7 first (see module docstring)
    p = path or os.environ.get('S2D_LIB', LIB_PATH)
    if not os.path.exists(p):
        raise S2DLibraryError(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C gym-soccer-2d-env_amd/csrc`). There is no CPU fallback.")
    try:
        lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    except OSError as e:
        raise S2DLibraryError(f"cannot load {p}: {e}") from e
    for name, res, args in PROTOTYPES:
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise S2DLibraryError(f"{p} does not export {name}") from e
        fn.restype = res
        fn.argtypes = list(args)
    _lib = lib
    return lib


def check(lib, rc, what):
    """Translate a C return code into the Python exception the boundary promises
    (SURVEY.md 8b 'Errors': ValueError for bad arguments, RuntimeError for HIP errors)."""
    if rc == S2D_OK:
        return
    msg = lib.s2d_last_error()
    msg = msg.decode() if msg else ''
    text = f"{what} failed ({rc}): {msg}"
    if rc == S2D_EINVAL:
        raise ValueError(text)
    if rc == S2D_ENOMEM:
        raise MemoryError(text)
    raise RuntimeError(text)
