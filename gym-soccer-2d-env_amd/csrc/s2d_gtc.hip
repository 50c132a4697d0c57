// s2d_gtc.hip -- GoToCenter surrogate task (include/s2d_gtc.h): the reference's kinematic
// stand-in for reach_ball (python_sample_soccer_env.py:46-255) as a batched HIP kernel.
// One thread per env, 7 state words, obs row = one float4 (16 B per lane, coalesced).
// Same deterministic fp32 math spec as the other tasks; the tests hold an independent CPU
// restatement which this file matches bit for bit.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <string>

#include "s2d_device.h"
#include "../../include/s2d_gtc.h"

#define S2D_API extern "C" __attribute__((visibility("default")))
extern "C" void s2d_internal_set_error(const char* msg);
static int gfail(int code, const std::string& m) { s2d_internal_set_error(m.c_str()); return code; }

struct GParams { float x_min, x_max, y_min, y_max, min_dist; int max_steps, continuous, auto_reset; uint32_t seed_lo, seed_hi, gid_lo, gid_hi;
                 int turn, use_turn, adim; };   // adim = floats per action row (actor_out_size in the turn mode, else 1)
struct GEnv { float x, y, body, prev_distance, prev_angle_diff; int step_count, episode; };
enum { GF_X, GF_Y, GF_BODY, GF_PREV_D, GF_PREV_A, GF_STEP, GF_EPISODE, GF_PLANES };
struct GPtrs { float* S; int64_t stride; float* obs; float* reward; uint8_t* done; uint8_t* result; float* terminal_obs; unsigned long long* stats; };

S2D_DEV float g_wrap(float a) {                        // wrap_angle_deg :18-25 -> [-180, 180)
  float t = a + 180.0f;
  return (t - 360.0f * floorf(t * 0.002777777777777778f)) - 180.0f;
}
S2D_DEV float g_angle_to_center(float x, float y) { return g_wrap(atan2_deg(0.0f - y, 0.0f - x)); }   // :27-37
S2D_DEV float g_diff_abs(float a, float b) { return fabsf(g_wrap(a - b)); }                            // :39-44
S2D_DEV float4 g_obs(const GEnv& e) {                  // _get_obs :235-255
  float diff = g_wrap(g_angle_to_center(e.x, e.y) - e.body);
  return make_float4(diff * 0.005555555555555556f, e.body * 0.005555555555555556f, e.x * 0.01904761904761905f,
                     e.y * 0.029411764705882353f);
}
S2D_DEV void g_reset(const GParams& p, GEnv& e, uint32_t gl, uint32_t gh) {   // reset :115-134
  U4 w = philox4x32_10(gl, gh, (uint32_t)e.episode, (S2D_ST_RESET << 16) | 0u, p.seed_lo, p.seed_hi);
  e.x = p.x_min + rnd_u01(w.x) * (p.x_max - p.x_min);
  e.y = p.y_min + rnd_u01(w.y) * (p.y_max - p.y_min);
  e.body = -180.0f + rnd_u01(w.z) * 360.0f;
  e.step_count = 0; e.episode += 1;
  e.prev_distance = hypot2(e.x, e.y);
  e.prev_angle_diff = g_diff_abs(e.body, g_angle_to_center(e.x, e.y));
}
S2D_DEV float g_clip1(float v) { return v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v); }
// a = the action row (a.a0 only outside the turn mode), u = the selection uniform of :151
S2D_DEV void g_step(const GParams& p, GEnv& e, const Action4& a, float u, float& reward, int& done, int& result) {   // step :136-233
  float dash_r, turn_r = 0.0f;
  bool dash_selected = true, turn_selected = false;
  if (p.turn && p.continuous) {                         // :142-158
    dash_r = g_clip1(a.a0);                             // :143-144
    if (p.use_turn) {
      turn_r = g_clip1(a.a1);                           // :146
      const float dash_p = g_clip1(a.a2), turn_p = g_clip1(a.a3);   // :147-148
      const float et = exp_spec(turn_p), ed = exp_spec(dash_p);     // :149-150  softmax([turn_p, dash_p])
      const float p0 = et / (et + ed);
      turn_selected = u < p0;                           // :151  (p[0] is the TURN probability here, unlike reach_ball)
      dash_selected = !turn_selected;                   // :152
    }
  } else if (p.continuous) dash_r = g_clip1(a.a0);      // :159-162
  else dash_r = ((float)(int)a.a0 * 0.0625f - 0.5f) * 2.0f;   // :163-166
  if (dash_selected) {
    float dir = g_wrap(e.body + dash_r * 180.0f);       // :169
    float sn, cs;
    sincos_deg(dir, sn, cs);                            // :173-175
    e.x += cs; e.y += sn;                               // :178-179
  }
  if (turn_selected) e.body = g_wrap(e.body + turn_r * 180.0f);   // :181-183
  float d = hypot2(e.x, e.y);                           // :186
  float adiff = g_diff_abs(e.body, g_angle_to_center(e.x, e.y));   // :187-188
  float r = (e.prev_distance - d) + (e.prev_angle_diff - adiff) * 0.005555555555555556f;   // :191-194
  e.step_count += 1;                                    // :196
  int dn = 0, res = S2D_RESULT_NONE;
  if (e.x < p.x_min || e.x > p.x_max || e.y < p.y_min || e.y > p.y_max) { dn = 1; r -= 10.0f; res = S2D_RESULT_OUT; }   // :203-207
  else if (d < p.min_dist) { dn = 1; r += 10.0f; res = S2D_RESULT_GOAL; }                // :209-212
  else if (e.step_count >= p.max_steps) { dn = 1; r -= 5.0f; res = S2D_RESULT_TIMEOUT; }   // :214-217
  e.prev_distance = d; e.prev_angle_diff = adiff;       // :223-224
  reward = r; done = dn; result = res;
}
S2D_DEV void g_load(const GPtrs& q, int64_t i, GEnv& e) {
  e.x = q.S[GF_X * q.stride + i]; e.y = q.S[GF_Y * q.stride + i]; e.body = q.S[GF_BODY * q.stride + i];
  e.prev_distance = q.S[GF_PREV_D * q.stride + i]; e.prev_angle_diff = q.S[GF_PREV_A * q.stride + i];
  e.step_count = __float_as_int(q.S[GF_STEP * q.stride + i]); e.episode = __float_as_int(q.S[GF_EPISODE * q.stride + i]);
}
S2D_DEV void g_store(const GPtrs& q, int64_t i, const GEnv& e) {
  q.S[GF_X * q.stride + i] = e.x; q.S[GF_Y * q.stride + i] = e.y; q.S[GF_BODY * q.stride + i] = e.body;
  q.S[GF_PREV_D * q.stride + i] = e.prev_distance; q.S[GF_PREV_A * q.stride + i] = e.prev_angle_diff;
  q.S[GF_STEP * q.stride + i] = __int_as_float(e.step_count); q.S[GF_EPISODE * q.stride + i] = __int_as_float(e.episode);
}

__global__ __launch_bounds__(256) void s2d_gtc_reset_kernel(GParams p, GPtrs q, int64_t n, const uint8_t* __restrict__ mask) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n || (mask && !mask[i])) return;
  GEnv e; g_load(q, i, e);
  uint64_t gid = (((uint64_t)p.gid_hi << 32) | p.gid_lo) + (uint64_t)i;
  g_reset(p, e, (uint32_t)gid, (uint32_t)(gid >> 32));
  g_store(q, i, e);
  reinterpret_cast<float4*>(q.obs)[i] = g_obs(e);
  q.reward[i] = 0.0f; q.done[i] = 0; q.result[i] = 0;
}

struct GRoll { float* obs; void* action; float* reward; uint8_t* done; uint8_t* result; };

__global__ __launch_bounds__(256) void s2d_gtc_rollout_kernel(GParams p, GPtrs q, int64_t n, int n_steps,
                                                              const void* __restrict__ actions,
                                                              const float* __restrict__ select_u, GRoll ro) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool active = i < n;
  GEnv e{0, 0, 0, 0, 0, 0, 0};
  uint32_t gl = 0, gh = 0;
  if (active) {
    g_load(q, i, e);
    uint64_t gid = (((uint64_t)p.gid_hi << 32) | p.gid_lo) + (uint64_t)i;
    gl = (uint32_t)gid; gh = (uint32_t)(gid >> 32);
  }
  float reward = 0.0f; int done = 0, res = 0; float4 ob = make_float4(0, 0, 0, 0);
  unsigned int c1 = 0, c2 = 0, c3 = 0;
  for (int t = 0; t < n_steps; ++t) {
    if (active) {
      Action4 a{0.0f, 0.0f, 0.0f, 0.0f};
      float u = 0.0f;
      const int64_t row = (int64_t)t * n + i;
      if (actions) {
        if (!p.continuous) a.a0 = (float)static_cast<const int32_t*>(actions)[i];
        else {
          const float* ar = static_cast<const float*>(actions) + i * p.adim;
          a.a0 = ar[0];
          if (p.adim > 1) a.a1 = ar[1];
          if (p.adim > 2) a.a2 = ar[2];
          if (p.adim > 3) a.a3 = ar[3];
        }
      } else {
        U4 w = philox4x32_10(gl, gh, (uint32_t)e.episode, (S2D_ST_POLICY << 16) | (uint32_t)e.step_count, p.seed_lo, p.seed_hi);
        if (p.continuous) {
          a.a0 = rnd_u01(w.x) * 2.0f - 1.0f;
          if (p.adim > 1) a.a1 = rnd_u01(w.y) * 2.0f - 1.0f;
          if (p.adim > 2) a.a2 = rnd_u01(w.z) * 2.0f - 1.0f;
          if (p.adim > 3) a.a3 = rnd_u01(w.w) * 2.0f - 1.0f;
        } else a.a0 = (float)rnd_below(w.x, 16);
      }
      if (p.turn && p.continuous && p.use_turn) {        // the uniform of :151: caller's, or Philox SELECT stream
        if (select_u) u = select_u[i];
        else u = rnd_u01(philox4x32_10(gl, gh, (uint32_t)e.episode, (S2D_ST_SELECT << 16) | (uint32_t)e.step_count, p.seed_lo, p.seed_hi).x);
      }
      if (ro.action) {
        if (!p.continuous) static_cast<int32_t*>(ro.action)[row] = (int32_t)a.a0;
        else {
          float* ar = static_cast<float*>(ro.action) + row * p.adim;
          ar[0] = a.a0;
          if (p.adim > 1) ar[1] = a.a1;
          if (p.adim > 2) ar[2] = a.a2;
          if (p.adim > 3) ar[3] = a.a3;
        }
      }
      g_step(p, e, a, u, reward, done, res);
      ob = g_obs(e);
      c1 += res == 1; c2 += res == 2; c3 += res == 3;
      if (done && p.auto_reset) {
        reinterpret_cast<float4*>(q.terminal_obs)[i] = ob;
        g_reset(p, e, gl, gh);
        ob = g_obs(e);
      }
      if (ro.obs) reinterpret_cast<float4*>(ro.obs)[row] = ob;
      if (ro.reward) ro.reward[row] = reward;
      if (ro.done) ro.done[row] = (uint8_t)done;
      if (ro.result) ro.result[row] = (uint8_t)res;
    }
  }
  if (active) {
    g_store(q, i, e);
    reinterpret_cast<float4*>(q.obs)[i] = ob; q.reward[i] = reward; q.done[i] = (uint8_t)done; q.result[i] = (uint8_t)res;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { c1 += __shfl_down(c1, off); c2 += __shfl_down(c2, off); c3 += __shfl_down(c3, off); }
  unsigned long long* st = q.stats + (blockIdx.x % S2D_STATS_STRIPES) * 8;
  if ((threadIdx.x & 63) == 0) { if (c1) atomicAdd(&st[1], (unsigned long long)c1); if (c2) atomicAdd(&st[2], (unsigned long long)c2); if (c3) atomicAdd(&st[3], (unsigned long long)c3); }
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&st[0], (unsigned long long)n * (unsigned long long)n_steps);
}

// ---------------------------------------------------------------- host
struct S2DGtcEngine { S2DGtcConfig cfg; GParams gp; int64_t n, stride; int device; char* arena; size_t bytes; bool owns; GPtrs q; };
static size_t g_align(size_t v, size_t a) { return (v + a - 1) / a * a; }
struct GLayout { size_t S, obs, reward, done, result, term, stats, total; int64_t stride; };
static GLayout g_layout(int64_t n) {
  GLayout L; L.stride = (int64_t)g_align((size_t)n, 256); size_t s = (size_t)L.stride, off = 0;
  L.S = off; off += g_align((size_t)GF_PLANES * s * 4, 256);
  L.obs = off; off += g_align(s * 16, 256); L.reward = off; off += g_align(s * 4, 256);
  L.done = off; off += g_align(s, 256); L.result = off; off += g_align(s, 256);
  L.term = off; off += g_align(s * 16, 256); L.stats = off; off += (size_t)S2D_STATS_STRIPES * 64;
  L.total = off; return L;
}
S2D_API void s2d_gtc_default_config(S2DGtcConfig* c) {
  if (!c) return;
  std::memset(c, 0, sizeof *c);
  c->abi_version = S2D_ABI_VERSION; c->struct_bytes = (uint32_t)sizeof *c;
  c->x_min = -52.5; c->x_max = 52.5; c->y_min = -34.0; c->y_max = 34.0;     // python_sample_soccer_env.py:93-94
  c->min_distance_to_center = 5.0; c->max_steps = 200; c->continuous = 0;   // :97-98
  c->turn = 0; c->use_turn = 0; c->actor_out_size = 1;                       // the class's own defaults, :66
  c->seed = 0x5EEDull; c->auto_reset = 1;
}
S2D_API size_t s2d_gtc_arena_bytes(const S2DGtcConfig* cfg, int64_t n) { return (!cfg || n <= 0) ? 0 : g_layout(n).total; }
S2D_API int s2d_gtc_reset(S2DGtcHandle h, const uint8_t* mask, void* stream) {
  if (!h) return gfail(S2D_EINVAL, "NULL handle");
  hipLaunchKernelGGL(s2d_gtc_reset_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), h->gp, h->q, h->n, mask);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? S2D_OK : gfail(S2D_EHIP, hipGetErrorString(e));
}
S2D_API int s2d_gtc_create(const S2DGtcConfig* cfg, int64_t n, int device, void* arena, size_t arena_bytes, void* stream, S2DGtcHandle* out) {
  if (!out) return gfail(S2D_EINVAL, "out handle is NULL");
  *out = nullptr;
  if (!cfg || cfg->abi_version != S2D_ABI_VERSION || cfg->struct_bytes != sizeof(S2DGtcConfig)) return gfail(S2D_EINVAL, "bad S2DGtcConfig header");
  if (!(cfg->x_max > cfg->x_min) || !(cfg->y_max > cfg->y_min) || cfg->max_steps < 1) return gfail(S2D_EINVAL, "bad field bounds / max_steps");
  if (n <= 0) return gfail(S2D_EINVAL, "n_envs must be positive");
  const bool turn_mode = cfg->turn && cfg->continuous;
  if (turn_mode && (cfg->actor_out_size < 1 || cfg->actor_out_size > 4)) return gfail(S2D_EINVAL, "actor_out_size must be in [1, 4]");
  if (turn_mode && cfg->use_turn && cfg->actor_out_size < 4)       // the reference indexes actions[3] (:148)
    return gfail(S2D_EINVAL, "use_turn needs actor_out_size = 4");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return gfail(S2D_ENODEV, "no HIP device visible");
  if (device < 0 || device >= ndev) return gfail(S2D_EINVAL, "device index out of range");
  if (hipSetDevice(device) != hipSuccess) return gfail(S2D_EHIP, "hipSetDevice failed");
  GLayout L = g_layout(n);
  S2DGtcEngine* h = new (std::nothrow) S2DGtcEngine();
  if (!h) return gfail(S2D_ENOMEM, "host allocation failed");
  h->cfg = *cfg; h->n = n; h->stride = L.stride; h->device = device;
  if (arena) { if (arena_bytes < L.total || (reinterpret_cast<uintptr_t>(arena) & 255u)) { delete h; return gfail(S2D_ENOMEM, "arena too small or unaligned"); } h->arena = static_cast<char*>(arena); h->owns = false; }
  else { void* pm = nullptr; if (hipMalloc(&pm, L.total) != hipSuccess) { delete h; return gfail(S2D_ENOMEM, "hipMalloc failed"); } h->arena = static_cast<char*>(pm); h->owns = true; }
  h->bytes = L.total;
  h->gp = GParams{(float)cfg->x_min, (float)cfg->x_max, (float)cfg->y_min, (float)cfg->y_max, (float)cfg->min_distance_to_center,
                  cfg->max_steps, cfg->continuous, cfg->auto_reset, (uint32_t)cfg->seed, (uint32_t)(cfg->seed >> 32),
                  (uint32_t)(uint64_t)cfg->env_id_offset, (uint32_t)((uint64_t)cfg->env_id_offset >> 32),
                  cfg->turn, cfg->use_turn, turn_mode ? cfg->actor_out_size : 1};
  h->q = GPtrs{reinterpret_cast<float*>(h->arena + L.S), L.stride, reinterpret_cast<float*>(h->arena + L.obs), reinterpret_cast<float*>(h->arena + L.reward),
               reinterpret_cast<uint8_t*>(h->arena + L.done), reinterpret_cast<uint8_t*>(h->arena + L.result), reinterpret_cast<float*>(h->arena + L.term),
               reinterpret_cast<unsigned long long*>(h->arena + L.stats)};
  if (hipMemsetAsync(h->arena, 0, L.total, static_cast<hipStream_t>(stream)) != hipSuccess) { if (h->owns) (void)hipFree(h->arena); delete h; return gfail(S2D_EHIP, "hipMemsetAsync failed"); }
  *out = h;
  return S2D_OK;
}
S2D_API void s2d_gtc_destroy(S2DGtcHandle h) { if (!h) return; if (h->owns && h->arena) (void)hipFree(h->arena); delete h; }
S2D_API int s2d_gtc_buffer_offsets(S2DGtcHandle h, int64_t* off, int n_off) {
  if (!h || !off || n_off < 14) return gfail(S2D_EINVAL, "offsets array too small (need 14)");
  const char* b = h->arena; const float* S = h->q.S; const int64_t s = h->stride;
  const void* ptrs[] = {S + GF_X * s, S + GF_Y * s, S + GF_BODY * s, S + GF_PREV_D * s, S + GF_PREV_A * s, S + GF_STEP * s, S + GF_EPISODE * s,
                        h->q.obs, h->q.reward, h->q.done, h->q.result, h->q.terminal_obs, h->q.stats};
  off[0] = (int64_t)h->bytes;
  for (int k = 0; k < 13; ++k) off[k + 1] = (int64_t)(static_cast<const char*>(ptrs[k]) - b);
  return S2D_OK;
}
static int g_launch(S2DGtcHandle h, int n_steps, const void* actions, const float* select_u, const S2DGtcRollout* out, void* stream) {
  GRoll ro{nullptr, nullptr, nullptr, nullptr, nullptr};
  if (out) ro = GRoll{out->obs, out->action, out->reward, out->done, out->result};
  hipLaunchKernelGGL(s2d_gtc_rollout_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), h->gp, h->q, h->n,
                     n_steps, actions, select_u, ro);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? S2D_OK : gfail(S2D_EHIP, hipGetErrorString(e));
}
S2D_API int s2d_gtc_step(S2DGtcHandle h, const void* actions, void* stream) { return h ? g_launch(h, 1, actions, nullptr, nullptr, stream) : gfail(S2D_EINVAL, "NULL handle"); }
S2D_API int s2d_gtc_step_u(S2DGtcHandle h, const void* actions, const float* select_u, void* stream) {
  return h ? g_launch(h, 1, actions, select_u, nullptr, stream) : gfail(S2D_EINVAL, "NULL handle");
}
S2D_API int s2d_gtc_rollout(S2DGtcHandle h, int n_steps, const S2DGtcRollout* out, void* stream) {
  if (!h) return gfail(S2D_EINVAL, "NULL handle");
  if (n_steps <= 0) return n_steps == 0 ? S2D_OK : gfail(S2D_EINVAL, "n_steps must be >= 0");
  return g_launch(h, n_steps, nullptr, nullptr, out, stream);
}
