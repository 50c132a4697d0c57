/*
 * s2d_oracle_common.h -- shared part of the CPU oracle (TEST INFRASTRUCTURE ONLY, see
 * s2d_oracle.c): REAL selection, Philox4x32-10, and the elementary functions (fp64 build =
 * libm as the reference's Python; fp32 build = the deterministic spec of DESIGN.md section 4).
 */
#ifndef S2D_ORACLE_COMMON_H_
#define S2D_ORACLE_COMMON_H_
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/s2d.h"

#ifdef S2DO_F64
typedef double REAL;
#define R(x) x
#else
typedef float REAL;
#define R(x) x##f
#endif

#define API __attribute__((visibility("default")))

/* x / c for a constant c.  F64 (reference semantics): a true division.  F32 spec: one
 * multiplication by the correctly rounded reciprocal `ic`. */
#ifdef S2DO_F64
#define DIVC(x, c, ic) ((x) / (c))
#else
#define DIVC(x, c, ic) ((x) * (ic))
#endif

/* ======================================================================================
 * Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11).
 * Counter layout of this project:
 *   ctr = { gid_lo, gid_hi, cycle, (stream << 16) | block },  key = { seed_lo, seed_hi }
 * ==================================================================================== */
enum { ST_RESET = 0, ST_POLICY = 1, ST_SELECT = 2, ST_NOISE = 3, ST_NOISE_RESET = 5 };   /* 4 = tackle (match) */

static void s2do_philox4x32_10_impl(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static void draw(uint64_t seed, uint64_t gid, uint32_t cycle, uint32_t stream, uint32_t block,
                 uint32_t w[4]) {
  uint32_t ctr[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), cycle, (stream << 16) | block};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  s2do_philox4x32_10_impl(ctr, key, w);
}
/* integer in [0, span): multiply-high (bias <= span / 2^32) */
static int32_t rnd_below(uint32_t w, uint32_t span) { return (int32_t)(((uint64_t)w * span) >> 32); }
/* uniform in [0,1) with 24 bits, exact in float and double */
static REAL rnd_u01(uint32_t w) { return (REAL)(w >> 8) * R(5.9604644775390625e-8); }

/* ======================================================================================
 * Elementary functions.  F64: libm, as pyrusgeom / numpy do.  F32: DESIGN.md section 4.
 * ==================================================================================== */
#ifdef S2DO_F64
static void sincos_deg(REAL deg, REAL *s, REAL *c) {
  const double DEG2RAD = 3.14159265358979323846 / 180.0;
  *s = sin(deg * DEG2RAD);
  *c = cos(deg * DEG2RAD);
}
/* pyrusgeom AngleDeg.atan2_deg: 0 for the zero vector */
static REAL atan2_deg(REAL y, REAL x) {
  const double RAD2DEG = 180.0 / 3.14159265358979323846;
  if (x == 0.0 && y == 0.0) return 0.0;
  return atan2(y, x) * RAD2DEG;
}
static REAL sq2(REAL x, REAL y) { return x * x + y * y; }
static REAL hypot2(REAL x, REAL y) { return sqrt(x * x + y * y); }
static REAL exp_r(REAL x) { return exp(x); }
#else
static void sincos_deg(REAL deg, REAL *s, REAL *c) {
  float q = rintf(deg * 0.011111111111111112f);
  float r = fmaf(-q, 90.0f, deg);            /* exact: r in [-45, 45] */
  float x = r * 0.017453292519943295f;
  float z = x * x;
  float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = fmaf(z, ps, -1.6666654611e-1f);
  ps = fmaf(x * z, ps, x);
  float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = fmaf(z, pc, 4.166664568298827e-2f);
  pc = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
  switch (((int)q) & 3) {
    case 0: *s = ps; *c = pc; break;
    case 1: *s = pc; *c = -ps; break;
    case 2: *s = -ps; *c = -pc; break;
    default: *s = -pc; *c = ps; break;
  }
}
static REAL atan2_deg(REAL y, REAL x) {
  float ax = fabsf(x), ay = fabsf(y);
  float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  if (mx == 0.0f) return 0.0f;
  /* one division: atan(mn/mx) = 45deg + atan((mn-mx)/(mn+mx)) above tan(pi/8) */
  int hi = mn > 0.41421356237f * mx;
  float num = hi ? mn - mx : mn;
  float den = hi ? mn + mx : mx;
  float base = hi ? 45.0f : 0.0f;
  float t = num / den;
  float z = t * t;
  float p = fmaf(z, 8.05374449538e-2f, -1.38776856032e-1f);
  p = fmaf(z, p, 1.99777106478e-1f);
  p = fmaf(z, p, -3.33329491539e-1f);
  float a = fmaf(p * z, t, t);
  a = fmaf(a, 57.29577951308232f, base);
  if (ay > ax) a = 90.0f - a;
  if (x < 0.0f) a = 180.0f - a;
  if (y < 0.0f) a = -a;
  return a;
}
static REAL sq2(REAL x, REAL y) { return fmaf(x, x, y * y); }
static REAL hypot2(REAL x, REAL y) { return sqrtf(fmaf(x, x, y * y)); }
static REAL exp_r(REAL x) {
  float k = rintf(x * 1.44269504088896341f);
  float r = fmaf(-k, 0.693359375f, x);
  r = fmaf(-k, -2.12194440e-4f, r);
  float z = r * r;
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  float y = fmaf(p, z, r) + 1.0f;
  return ldexpf(y, (int)k);
}
#endif

/* pyrusgeom AngleDeg.__init__/normal(): fmod by 360 when |d| > 360, then one +-360. */
static REAL norm_deg(REAL d) {
  if (d < R(-360.0) || R(360.0) < d) d = R(fmod)(d, R(360.0));
  if (d < R(-180.0)) d += R(360.0);
  if (d > R(180.0)) d -= R(360.0);
  return d;
}


#endif /* S2D_ORACLE_COMMON_H_ */
