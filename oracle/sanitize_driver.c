/*
 * sanitize_driver.c -- runs the three CPU oracles under AddressSanitizer + UndefinedBehaviorSanitizer
 * (`make -C oracle sanitize`; GPU ASan is not available on this pool, so the sanitizers cover the CPU restatement the
 * kernels are checked against -- SURVEY.md section 5 "race detection / sanitizers").  TEST INFRASTRUCTURE ONLY.
 * The configurations arrive as raw struct images written by tests/test_oracle_sanitizers.py.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/s2d.h"
#include "../include/s2d_gtc.h"
#include "../include/s2d_match.h"

typedef struct S2DOEngine S2DOEngine;
typedef struct S2DMOEngine S2DMOEngine;
typedef struct GEngine GEngine;
S2DOEngine *s2do_create(const S2DConfig *cfg, int64_t n);
void s2do_destroy(S2DOEngine *h);
void s2do_reset(S2DOEngine *h, const uint8_t *mask);
void s2do_step(S2DOEngine *h, const void *actions, int kind);
void s2do_rollout(S2DOEngine *h, int n_steps, const void *actions, int kind, float *obs, void *action, float *reward,
                  uint8_t *done, uint8_t *result);
const unsigned long long *s2do_stats(const S2DOEngine *h);
S2DMOEngine *s2dmo_create(const S2DMatchConfig *cfg, int64_t n);
void s2dmo_destroy(S2DMOEngine *h);
void s2dmo_reset(S2DMOEngine *h, const uint8_t *mask);
void s2dmo_step(S2DMOEngine *h, const float *actions);
const unsigned long long *s2dmo_stats(const S2DMOEngine *h);
GEngine *s2dgo_create(const S2DGtcConfig *c, int64_t n);
void s2dgo_destroy(GEngine *h);
void s2dgo_reset(GEngine *h, const uint8_t *mask);
void s2dgo_step(GEngine *h, const double *actions);
const unsigned long long *s2dgo_stats(const GEngine *h);

static int load(const char *path, void *dst, size_t n) {
  FILE *f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); return -1; }
  size_t got = fread(dst, 1, n, f);
  fclose(f);
  if (got != n) { fprintf(stderr, "%s: %zu bytes, expected %zu\n", path, got, n); return -1; }
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s reach.cfg match.cfg gtc.cfg\n", argv[0]); return 2; }
  S2DConfig rc; S2DMatchConfig mc; S2DGtcConfig gc;
  if (load(argv[1], &rc, sizeof rc) || load(argv[2], &mc, sizeof mc) || load(argv[3], &gc, sizeof gc)) return 2;
  const int64_t n = 257;                                   /* ragged on purpose */
  const int T = 300;
  {                                                        /* reach_ball: rollout with records, then per-step, masked reset */
    S2DOEngine *h = s2do_create(&rc, n);
    float *obs = malloc((size_t)T * n * 10 * sizeof(float)), *rew = malloc((size_t)T * n * sizeof(float));
    int32_t *act = malloc((size_t)T * n * 4 * sizeof(float));
    uint8_t *done = malloc((size_t)T * n), *res = malloc((size_t)T * n), *mask = calloc((size_t)n, 1);
    s2do_reset(h, NULL);
    s2do_rollout(h, T, NULL, S2D_ACT_RANDOM, obs, act, rew, done, res);
    for (int64_t i = 0; i < n; i += 3) mask[i] = 1;
    s2do_reset(h, mask);
    for (int t = 0; t < 50; ++t) s2do_step(h, NULL, S2D_ACT_RANDOM);
    printf("reach_ball: %llu env-steps, %llu goals %llu outs %llu timeouts\n", s2do_stats(h)[0], s2do_stats(h)[1],
           s2do_stats(h)[2], s2do_stats(h)[3]);
    free(obs); free(rew); free(act); free(done); free(res); free(mask);
    s2do_destroy(h);
  }
  {                                                        /* 11v11 */
    S2DMOEngine *h = s2dmo_create(&mc, 33);
    s2dmo_reset(h, NULL);
    for (int t = 0; t < 400; ++t) s2dmo_step(h, NULL);
    printf("match: %llu match-steps, %llu kicks\n", s2dmo_stats(h)[0], s2dmo_stats(h)[4]);
    s2dmo_destroy(h);
  }
  {                                                        /* GoToCenter, the script's default turn / use_turn mode */
    GEngine *h = s2dgo_create(&gc, n);
    s2dgo_reset(h, NULL);
    for (int t = 0; t < 400; ++t) s2dgo_step(h, NULL);
    printf("gtc: %llu env-steps\n", s2dgo_stats(h)[0]);
    s2dgo_destroy(h);
  }
  printf("sanitize ok\n");
  return 0;
}
