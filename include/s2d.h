/*
 * s2d.h -- C ABI of the MI355X-native batched 2D-soccer engine (libs2d_hip.so).
 *
 * This is the drop-in boundary for the `reach_ball` hot path of
 * CLSFramework/gym-soccer-2d-env.  The reference has no C ABI / FFI of its own (its
 * boundary is the Python gym.Env surface); each entry point below therefore cites the
 * reference *Python* interface it replaces (file:line under /root/reference):
 *
 *   s2d_create / s2d_destroy   Soccer2DEnv.__init__ / close      soccer_2d_env.py:30-95, 280-299
 *                              (spawn rcssserver + proxy + gRPC  -> one in-process engine)
 *   s2d_reset                  Soccer2DEnv.reset -> abs_reset -> env_reset
 *                              soccer_2d_env.py:179-224, reach_ball_env.py:163-218
 *   s2d_step                   Soccer2DEnv.step                  soccer_2d_env.py:226-269
 *                              + ReachBallEnv hooks              reach_ball_env.py:53-161
 *   s2d_rollout                SB3 collect_rollouts loop over step()  dqn_stable_baselines3.py:41-55
 *                              (T fused steps, random policy or caller actions)
 *   s2d_world_model            protobuf State/WorldModel fields  idl/service.proto:22-27, 68-86,
 *                              144-223, 306-349 (returned as device arrays, not wire bytes)
 *   S2DConfig                  ReachBallEnv kwargs               reach_ball_env.py:26-36
 *                              + ServerParam / PlayerType names  idl/service.proto:1435-1732
 *
 * Plain C: pointers and sizes only, no torch / C++ types.  All `*_dev` pointers are HIP
 * device pointers; `stream` is a hipStream_t passed as void* (NULL = the null stream).
 * Every launch function is stream-ordered, asynchronous and hipGraph-capturable (no
 * allocation, no synchronisation inside).  One engine per (process, GPU); a handle is not
 * thread-safe, different handles are independent.
 */
#ifndef S2D_H_
#define S2D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S2D_ABI_VERSION 4

/* ---- error codes (0 = ok, negative = failure; text via s2d_last_error()) ------------ */
enum {
  S2D_OK = 0,
  S2D_EINVAL = -1,   /* bad argument / config (Python raises ValueError)           */
  S2D_EHIP = -2,     /* a HIP runtime call failed (Python raises RuntimeError)      */
  S2D_ENOMEM = -3,   /* arena too small / allocation failed                         */
  S2D_ENODEV = -4    /* no usable gfx950 device                                     */
};

/* ---- enums mirroring the reference ---------------------------------------------------- */
/* GameModeType values, idl/service.proto:267-301 (only the ones the path can reach).    */
enum { S2D_MODE_BEFORE_KICK_OFF = 0, S2D_MODE_TIME_OVER = 1, S2D_MODE_PLAY_ON = 2 };
/* Side, idl/service.proto:88-92 */
enum { S2D_SIDE_UNKNOWN = 0, S2D_SIDE_LEFT = 1, S2D_SIDE_RIGHT = 2 };
/* info['result'] labels, reach_ball_env.py:126-150 (None / 'Goal' / 'Out' / 'Timeout') */
enum { S2D_RESULT_NONE = 0, S2D_RESULT_GOAL = 1, S2D_RESULT_OUT = 2, S2D_RESULT_TIMEOUT = 3 };
/* low-level body commands = PlayerAction oneof members, idl/service.proto:380-397 */
enum { S2D_CMD_NONE = 0, S2D_CMD_DASH = 1, S2D_CMD_TURN = 2,
       S2D_CMD_FREEZE = -1 /* S2D_ACT_COMMAND only: this env does not take part in the cycle (state and outputs stay as they are) --
                            * how a host mirror lets ONE env of a batch consume its reset cycle (soccer_2d_env.py:187-197) */ };

/* how `actions_dev` of s2d_step / s2d_rollout is laid out (reach_ball_env.py:39-47, 53-85) */
enum {
  S2D_ACT_DISCRETE_I32 = 0, /* int32[N]    Discrete(n)                                   */
  S2D_ACT_DISCRETE_I64 = 1, /* int64[N]    same, torch's default integer dtype           */
  S2D_ACT_CONTINUOUS = 2,   /* float[N][1] Box(-1,1,(1,))  rel_dir = a*180 (not clipped) */
  S2D_ACT_TURNING = 3,      /* float[N][4] Box(-1,1,(4,))  [turn_p, turn_a, dash_p, dash_a] */
  S2D_ACT_RANDOM = 4,       /* NULL: uniform random policy drawn in-kernel (Philox)      */
  /* float[N][4] = {S2D_CMD_*, power, relative direction, 0}: one decoded PlayerAction body command per env, executed as it is
   * (what the proxy does with `ignore_preprocess=True`, server.py:64) -- the boundary of the reference's task HOOK
   * `action_to_rpc_actions` (soccer_2d_env.py:317-325): a user-defined task env builds pb2.PlayerAction objects in Python and the
   * host mirror turns them into this array.  Accepted by s2d_step in every task mode (16-byte aligned); not by s2d_rollout.
   * (The env-step counter of the episode statistics counts every env of the launch, frozen ones included.) */
  S2D_ACT_COMMAND = 5   /* the command word is read as the nearest of -1 / 0 / 1 / 2 (S2D_CMD_*); any other number, NaN included, is
                         * S2D_CMD_NONE */
};

/* ---- configuration -------------------------------------------------------------------- */
/* Physics parameters.  Field names follow ServerParam / PlayerType of idl/service.proto:
 * 1435-1732.  The reference never holds their VALUES (rcssserver sends them at run time,
 * server.py:105-118); the defaults of s2d_default_config() are rcssserver's stock values
 * (EXT / parity-unpinned).  Doubles here; each engine rounds to its
 * own arithmetic type (the HIP engine computes in float).                                 */
typedef struct S2DServerParams {
  double pitch_half_length;  /* 52.5  reach_ball_env.py:100,142 */
  double pitch_half_width;   /* 34.0  reach_ball_env.py:101,142 */
  double player_size, player_decay, player_rand, player_speed_max, player_accel_max;
  double inertia_moment;
  double stamina_max, stamina_inc_max, stamina_capacity, extra_stamina;
  double recover_init, recover_dec_thr, recover_min, recover_dec;
  double effort_init, effort_dec_thr, effort_min, effort_dec, effort_inc_thr, effort_inc;
  double dash_power_rate, max_dash_power, min_dash_power;
  double max_dash_angle, min_dash_angle, dash_angle_step, side_dash_rate, back_dash_rate;
  double max_moment, min_moment;
  double ball_size, ball_decay, ball_rand, ball_speed_max, ball_accel_max;
  double collision_vel_rate; /* -0.1: velocity factor applied to collided objects          */
} S2DServerParams;

/* Task parameters = ReachBallEnv kwargs, reach_ball_env.py:26-36, same names/defaults.   */
typedef struct S2DReachBallParams {
  int32_t change_ball_position;  /* True  */
  int32_t change_ball_velocity;  /* False */
  double ball_position_x, ball_position_y, ball_speed, ball_direction; /* 0 */
  double min_distance_to_ball;   /* 5.0 */
  int32_t max_steps;             /* 200 */
  int32_t use_continuous_action; /* True */
  int32_t action_space_size;     /* 16  */
  int32_t use_turning;           /* False */
  double reset_ball_decay;       /* 0.96: literal used by get_ball_velocity, reach_ball_env.py:207 */
} S2DReachBallParams;

typedef struct S2DConfig {
  uint32_t abi_version;   /* S2D_ABI_VERSION */
  uint32_t struct_bytes;  /* sizeof(S2DConfig), checked by s2d_create */
  S2DServerParams sp;
  S2DReachBallParams task;
  uint64_t seed;          /* Philox key; resets use stream 0, random policy stream 1, ... */
  int64_t env_id_offset;  /* global id of local env 0 (multi-GPU sharding: results are
                             invariant to how the global env range is cut into shards)   */
  int32_t auto_reset;     /* 1: a done env is reset inside the same step (SB3 VecEnv
                             convention); 0: caller resets (reference single-env flow)   */
  int32_t noise;          /* 0: player_rand/ball_rand ignored (parity mode); 1: Philox noise */
  int32_t reserved[4];
} S2DConfig;

/* ---- device buffers --------------------------------------------------------------------
 * All arrays are struct-of-arrays over the N local envs (one contiguous array per field,
 * 256-byte aligned), except obs / terminal_obs which are row-major [N][10] as PyTorch
 * consumers expect.  Valid for the life of the handle; contents change at every
 * s2d_step / s2d_reset / s2d_rollout.                                                     */
#define S2D_OBS_DIM 10
#define S2D_STATS_STRIPES 64
/* rows of S2DBuffers.stats of an engine of n envs: one per group of 64 envs (its wave owns it), at least S2D_STATS_STRIPES */
#define S2D_STATS_ROWS(n) ((((n) + 63) / 64) > S2D_STATS_STRIPES ? (((n) + 63) / 64) : S2D_STATS_STRIPES)
typedef struct S2DBuffers {
  int64_t n_envs;
  /* state, row S of SURVEY.md 8(a): 15 float + 2 int32 words per env (+ policy_step, which only
   * launches that draw in-engine policy randomness read or write) */
  float *player_x, *player_y, *player_vx, *player_vy, *player_body; /* body in degrees [-180,180] */
  float *stamina, *effort, *recovery, *stamina_capacity;
  float *ball_x, *ball_y, *ball_vx, *ball_vy;
  float *prev_dist, *prev_angle;  /* carry of check_trainer_observation, reach_ball_env.py:158-159 */
  int32_t *step_number;           /* reach_ball_env.py:55, 172 */
  int32_t *cycle;                 /* WorldModel.cycle, idl/service.proto:326 */
  int32_t *policy_step;           /* steps that consumed an in-engine policy / select draw (S2D_ACT_RANDOM,
                                   * or any step of a use_turning env): the Philox counter of those draws */
  int32_t *episode;               /* resets so far = index of the current episode (0 before the first reset): the
                                   * Philox counter of the reset sampler, so that episode j of env g is a function of
                                   * (g, j) alone and can be prepared ahead of the simulation */
  /* per-step outputs */
  float *obs;            /* [N][10]  reach_ball_env.py:98-107 */
  float *reward;         /* [N] */
  uint8_t *done;         /* [N] 0/1 */
  uint8_t *result;       /* [N] S2D_RESULT_* */
  float *terminal_obs;   /* [N][10] observation of the finished episode (valid where done) */
  float *action_dir;     /* [N] decoded relative direction in degrees of the last command  */
  uint8_t *action_cmd;   /* [N] S2D_CMD_* of the last command                              */
  /* episode statistics, one row per group of 64 envs (plain load / store by the wave that owns the group, no atomics):
   * stats[S2D_STATS_ROWS(n_envs)][8]; the value of counter k is the sum over rows of [r][k].
   * k: 0 = env-steps, 1 = Goal, 2 = Out, 3 = Timeout, 4..7 reserved                        */
  unsigned long long *stats;
} S2DBuffers;

/* Caller-owned rollout buffers for s2d_rollout, time-major (any pointer may be NULL to
 * skip that output).  T = n_steps, N = local envs.                                        */
typedef struct S2DRollout {
  float *obs;        /* [T][N][10] observation returned by step t (post auto-reset)        */
  void *action;      /* [T][N] int32 (discrete) | float[T][N][1] | float[T][N][4]          */
  float *reward;     /* [T][N] */
  uint8_t *done;     /* [T][N] */
  uint8_t *result;   /* [T][N] */
} S2DRollout;

/* Derived protobuf-mirroring fields that are not plain state words (row T1).  Each array
 * is [N]; NULL pointers are skipped.                                                       */
typedef struct S2DWorldModel {
  float *ball_dist_from_self;    /* Ball.dist_from_self   idl/service.proto:84 */
  float *ball_angle_from_self;   /* Ball.angle_from_self  idl/service.proto:85 */
  float *ball_relative_x, *ball_relative_y; /* Ball.relative_position idl/service.proto:70 */
  float *ball_pos_dist, *ball_pos_angle;    /* RpcVector2D.dist/.angle of Ball.position :25-26 */
  float *ball_vel_dist, *ball_vel_angle;    /* ... of Ball.velocity */
  float *self_pos_dist, *self_pos_angle;    /* ... of Self.position */
  float *self_vel_dist, *self_vel_angle;    /* ... of Self.velocity */
  float *self_dist_from_ball;    /* Self.dist_from_ball   idl/service.proto:204 */
  float *self_angle_from_ball;   /* Self.angle_from_ball  idl/service.proto:205 */
} S2DWorldModel;

typedef struct S2DEngine *S2DHandle;

/* ---- entry points ---------------------------------------------------------------------- */
const char *s2d_version(void);
/* last error text of the calling thread ("" if none) */
const char *s2d_last_error(void);
/* rcssserver stock ServerParam/PlayerType(0) values + ReachBallEnv kwargs defaults */
void s2d_default_config(S2DConfig *cfg);
/* 0 if cfg is acceptable, else S2D_EINVAL with s2d_last_error() set */
int s2d_validate_config(const S2DConfig *cfg);
/* bytes of device memory an engine of n_envs needs (0 on bad input) */
size_t s2d_arena_bytes(const S2DConfig *cfg, int64_t n_envs);
/* arena_dev == NULL: the engine hipMallocs (and owns) its arena; otherwise the caller owns
 * `arena_dev` (>= s2d_arena_bytes, 256-byte aligned) and keeps it alive until s2d_destroy.
 * The arena is zero-filled and the state initialised on `stream`. */
int s2d_create(const S2DConfig *cfg, int64_t n_envs, int device, void *arena_dev,
               size_t arena_bytes, void *stream, S2DHandle *out);
void s2d_destroy(S2DHandle h);
int s2d_buffers(S2DHandle h, S2DBuffers *out);
/* byte offset of every S2DBuffers pointer from the arena base, same field order
 * (n_envs slot = arena size); lets a host language build zero-copy views of an arena
 * it allocated itself. */
int s2d_buffer_offsets(S2DHandle h, int64_t *offsets, int n_offsets);
/* reset envs where mask_dev[i] != 0 (NULL = all): reach_ball_env.py:170-218 sampler, then
 * ONE simulator cycle (soccer_2d_env.py:187-197), obs + carry refreshed. */
int s2d_reset(S2DHandle h, const uint8_t *mask_dev, void *stream);
/* one cycle for every env: decode action, dash/turn, stamina, integrate, collide, decay,
 * obs, reward/done/result, auto-reset under cfg.auto_reset. */
int s2d_step(S2DHandle h, const void *actions_dev, int action_kind, void *stream);
/* k cycles (1 <= k <= 64) of the per-step API in ONE launch, for callers that hold their actions for k steps ahead (action repeat,
 * open-loop chunks): actions_dev is [k][N] in `action_kind` layout (or NULL with S2D_ACT_RANDOM); `out` (may be NULL, and any of its
 * arrays may be NULL) receives the per-step record [k][N]; the arena's per-step outputs hold the last step.  Same results as k calls
 * of s2d_step (soccer_2d_env.py:226-269 each); no prologue, the state makes one round trip. */
int s2d_step_k(S2DHandle h, int k, const void *actions_dev, int action_kind, const S2DRollout *out, void *stream);
/* n_steps cycles fused in ONE launch (state stays in registers).  actions_dev is
 * [T][N] in `action_kind` layout, or NULL with S2D_ACT_RANDOM. */
int s2d_rollout(S2DHandle h, int n_steps, const void *actions_dev, int action_kind,
                const S2DRollout *out, void *stream);
/* fill derived protobuf-mirroring fields from the current state */
int s2d_world_model(S2DHandle h, const S2DWorldModel *out, void *stream);
/* zero the statistics counters */
int s2d_stats_reset(S2DHandle h, void *stream);
/* name of the most recently launched kernel variant (for profiling reports) */
const char *s2d_kernel_name(S2DHandle h);
/* debug guard (the reference's desync recovery, soccer_2d_env.py:154-159, 257-260, has no counterpart in a
 * lockstep engine; what remains to watch is a state word leaving its domain).  counts_dev[8] (device memory) receives the
 * number of envs with: [0] a non-finite state word, [1] |body| or |prev_angle| > 180, [2] stamina outside [0, stamina_max] or a
 * negative capacity, [3] effort / recovery outside their ServerParam ranges, [4] a negative step / episode counter or
 * distance carry, [5] a non-finite observation; [6..7] reserved.  All zero for every state the engine produces. */
int s2d_validate_state(S2DHandle h, uint32_t *counts_dev, void *stream);
/* new Philox key for all later draws (gym's env.seed(); the reference's `random` / `np.random` are unseeded).  Takes
 * effect at the next launch on `stream`; stream-ordered like every other call (abi 3: the episodes s2d_step keeps prepared
 * are dropped by a hipMemsetAsync on `stream`, behind whatever is already queued there); callers normally follow it with
 * s2d_reset on the same stream. */
int s2d_set_seed(S2DHandle h, uint64_t seed, void *stream);
/* diagnostic: evaluate one primitive of the fp32 math spec / Philox on the device so that
 * tests can compare it bit for bit with the CPU oracle.  op: 0 sincos_deg (in[n] -> out[n][2]),
 * 1 atan2_deg (in[n][2]=y,x -> out[n]), 2 exp, 3 norm_deg, 4 philox4x32-10 (in = uint32[n][6]
 * ctr+key -> out = uint32[n][4]), 5 hypot (in[n][2] -> out[n]).
 * The three hooks of ReachBallEnv, evaluated stand-alone so that the device code can be checked against the
 * golden vectors produced by the reference's own Python (tests/golden):
 * 6 state_to_observation (reach_ball_env.py:87-111): in[n][9] = bx,by,bvx,bvy,px,py,body,1/half_length,1/half_width
 *   -> out[n][10];
 * 7 action_to_rpc_actions (:53-85): in[n][8] = a0,a1,a2,a3,u,mode(0 discrete,1 continuous,2 turning),360/n,0
 *   -> out[n][3] = S2D_CMD_*, power, relative direction;
 * 8 check_trainer_observation (:113-161): in[n][12] = bx,by,px,py,body,step_number,prev_dist,prev_angle,
 *   min_distance_to_ball,max_steps,half_length,half_width -> out[n][5] = done,reward,S2D_RESULT_*,dist,angle. */
/* 9 reset_sample_coop against reset_sample (in = uint32[n][4], n a multiple of 256; out[n][14]);
 * 10 the movement-noise draw of one commanded cycle (one Philox word per object and cycle -- magnitude uniform
 *   k / 65536 from its high half, direction a WHOLE degree from its low half): in = uint32[n][4] = global env id lo, hi, policy step k,
 *   seed (low word) -> out[n][6] = player magnitude uniform, sin, cos; ball magnitude uniform, sin, cos. */
int s2d_debug_eval(int op, const void *in_dev, void *out_dev, int64_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* S2D_H_ */
