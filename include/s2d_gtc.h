/*
 * s2d_gtc.h -- C ABI of the GoToCenter surrogate task: the reference's
 * pure-Python kinematic stand-in for reach_ball, GoToCenterEnv (python_sample_soccer_env.py:46-255),
 * as a second built-in batched task of libs2d_hip.so.  Same conventions as s2d.h.
 *
 *   s2d_gtc_reset   GoToCenterEnv.reset   python_sample_soccer_env.py:115-134
 *   s2d_gtc_step    GoToCenterEnv.step    python_sample_soccer_env.py:136-234
 *   obs             GoToCenterEnv._get_obs :236-255  [angle_diff/180, body/180, x/52.5, y/34]
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
  double x_min, x_max, y_min, y_max;      /* -52.5 52.5 -34 34      :91-92  */
  double min_distance_to_center;          /* 5.0                    :96     */
  int32_t max_steps;                      /* 200                    :95     */
  int32_t continuous;                     /* 0: Discrete(16) dash_r = (a/16 - .5)*2  :171-174; 1: Box(-1,1,(1,)) clipped :167-170 */
  uint64_t seed;
  int64_t env_id_offset;
  int32_t auto_reset;
  int32_t reserved[3];
} S2DGtcConfig;

typedef struct S2DGtcBuffers {
  int64_t n_envs;
  float *x, *y, *body, *prev_distance, *prev_angle_diff;   /* :98-109 */
  int32_t *step_count, *episode;
  float *obs;            /* [N][4] */
  float *reward;         /* [N] */
  uint8_t *done;         /* [N] terminated or truncated */
  uint8_t *result;       /* [N] S2D_RESULT_* ('' / 'Goal' / 'Out' / 'Timeout', :199-214) */
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
/* actions: int32[N] (discrete) or float[N] (continuous); NULL = uniform random policy */
int s2d_gtc_step(S2DGtcHandle h, const void *actions_dev, void *stream);
int s2d_gtc_rollout(S2DGtcHandle h, int n_steps, const S2DGtcRollout *out, void *stream);   /* random policy */

#ifdef __cplusplus
}
#endif
#endif
