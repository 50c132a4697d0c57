/*
 * s2d_gtc.h -- C ABI of the GoToCenter surrogate task: the reference's
 * pure-Python kinematic stand-in for reach_ball, GoToCenterEnv (python_sample_soccer_env.py:46-255),
 * as a second built-in batched task of libs2d_hip.so.  Same conventions as s2d.h.
 *
 *   s2d_gtc_reset   GoToCenterEnv.reset   python_sample_soccer_env.py:115-134
 *   s2d_gtc_step    GoToCenterEnv.step    python_sample_soccer_env.py:136-233
 *   obs             GoToCenterEnv._get_obs :235-255  [angle_diff/180, body/180, x/52.5, y/34]
 */
#ifndef S2D_GTC_H_
#define S2D_GTC_H_
#include "s2d.h"
#ifdef __cplusplus
extern "C" {
#endif

#define S2D_GTC_OBS_DIM 4

typedef struct S2DGtcConfig {
  uint32_t abi_version, struct_bytes;
  double x_min, x_max, y_min, y_max;      /* -52.5 52.5 -34 34      :93-94  */
  double min_distance_to_center;          /* 5.0                    :98     */
  int32_t max_steps;                      /* 200                    :97     */
  int32_t continuous;                     /* 0: Discrete(16) dash_r = (a/16 - .5)*2  :163-166; 1: Box(-1,1,(1,)) clipped :159-162 */
  uint64_t seed;
  int64_t env_id_offset;
  int32_t auto_reset;
  /* the script's --turn / --useturn / --actor_out_size switches (:70-79, :142-158; argparse defaults
   * True / True / 4, :356-359).  turn && continuous: Box(-1,1,(actor_out_size,)), actions[0] = dash angle;
   * with use_turn also actions[1] = turn angle, [2] = dash logit, [3] = turn logit, softmax([turn, dash]),
   * turn chosen iff U(0,1) < p[0] (:149-152); a turn changes the body, a dash does not. */
  int32_t turn, use_turn, actor_out_size;   /* 0, 0, 1 by default (the class's own defaults, :66) */
} S2DGtcConfig;

typedef struct S2DGtcBuffers {
  int64_t n_envs;
  float *x, *y, *body, *prev_distance, *prev_angle_diff;   /* :98-109 */
  int32_t *step_count, *episode;
  float *obs;            /* [N][4] */
  float *reward;         /* [N] */
  uint8_t *done;         /* [N] terminated or truncated */
  uint8_t *result;       /* [N] S2D_RESULT_* ('' / 'Goal' / 'Out' / 'Timeout', :198-217) */
  float *terminal_obs;   /* [N][4] */
  unsigned long long *stats;   /* [S2D_STATS_STRIPES][8]: env-steps, Goal, Out, Timeout */
} S2DGtcBuffers;

typedef struct S2DGtcRollout { float *obs; void *action; float *reward; uint8_t *done; uint8_t *result; } S2DGtcRollout;
typedef struct S2DGtcEngine *S2DGtcHandle;

void s2d_gtc_default_config(S2DGtcConfig *cfg);
size_t s2d_gtc_arena_bytes(const S2DGtcConfig *cfg, int64_t n_envs);
int s2d_gtc_create(const S2DGtcConfig *cfg, int64_t n_envs, int device, void *arena_dev, size_t arena_bytes,
                   void *stream, S2DGtcHandle *out);
void s2d_gtc_destroy(S2DGtcHandle h);
int s2d_gtc_buffer_offsets(S2DGtcHandle h, int64_t *offsets, int n_offsets);
int s2d_gtc_reset(S2DGtcHandle h, const uint8_t *mask_dev, void *stream);
/* actions: int32[N] (discrete), float[N] (continuous) or float[N][actor_out_size] (turn && continuous);
 * NULL = uniform random policy */
int s2d_gtc_step(S2DGtcHandle h, const void *actions_dev, void *stream);
/* same, with the turn / dash selection uniforms of :151 supplied by the caller (float[N] in [0,1)) instead of
 * the engine's Philox SELECT stream -- for callers that own the RNG, and for the reference-fixture tests */
int s2d_gtc_step_u(S2DGtcHandle h, const void *actions_dev, const float *select_u_dev, void *stream);
int s2d_gtc_rollout(S2DGtcHandle h, int n_steps, const S2DGtcRollout *out, void *stream);   /* random policy */

#ifdef __cplusplus
}
#endif
#endif
