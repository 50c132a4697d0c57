// s2d_engine.hip -- kernels and C ABI (include/s2d.h) of the MI355X-native reach_ball engine.
//
// Layout in HBM: one arena; the 17 state words of an env (+ its policy_step
// counter) live in 18 struct-of-arrays planes of `stride` (= N rounded up to 256) 4-byte words, so a wave's 64
// lanes read/write 256 contiguous bytes per plane.  Observations are emitted row-major
// [N][10] (what a PyTorch policy consumes): each wave transposes its 64x10 block through a
// wave-private LDS tile and stores 2560 contiguous bytes with 16-byte-per-lane stores.
// One thread = one env; there is no cross-env communication, so the only cross-lane work is
// the LDS transposition and the ballot/popcount reduction of the episode counters.
// No MFMA: the path has no dense contraction (arithmetic intensity ~0.5 flop/byte).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>

#include "s2d_device.h"

#include "s2d_kernels.h"

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
__global__ void s2d_tables_kernel(S2DHot p, const S2DRare* __restrict__ rp, S2DTables* __restrict__ t) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && rp->tab_len > 0) tables_build(p, rp->recover_init, rp->tab_power, rp->tab_len, *t);
}
__global__ __launch_bounds__(kBlock) void s2d_init_kernel(S2DHot p, const S2DRare* __restrict__ rp,
                                                          float* __restrict__ S, int64_t stride, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  S[F_STAMINA * stride + i] = p.stamina_max;             // state after a trainer (recover)
  S[F_RECOVERY * stride + i] = rp->recover_init;
  S[F_EFFORT * stride + i] = p.effort_init;
  S[F_CAPACITY * stride + i] = p.stamina_capacity;
}

template <bool NOISE>
__global__ __launch_bounds__(kBlock) void s2d_reach_reset_kernel(S2DHot p, const S2DRare* __restrict__ rp,
                                                                 float* __restrict__ S, int64_t stride, int64_t n,
                                                                 const uint8_t* __restrict__ mask, StepOut o) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][kObsTile];
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t wave_first = i - lane;
  const bool in_range = i < n;
  const bool active = in_range && (mask == nullptr || mask[i] != 0);
  const bool any = __ballot(active) != 0ull;
  if (!any) return;                                      // wave-uniform
  ObsOut ob;
  if (in_range) {
    Env e;
    env_load(e, S, stride, i);
    if (active) {
      uint64_t gid = (((uint64_t)p.gid_hi << 32) | p.gid_lo) + (uint64_t)i;
      float d2 = env_reset<NOISE>(p, rp, e, (uint32_t)gid, (uint32_t)(gid >> 32));
      int d, r; float w;
      observe_and_check(p, e, d2, ob, d, w, r);
      env_store(e, S, stride, i);
      o.reward[i] = 0.0f; o.done[i] = 0; o.result[i] = 0;
      if (p.auto_reset) {                                // the per-step API's prepared episodes (see StepOut::prep)
        prep_store<NOISE>(p, rp, o.prep, stride, i, (uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)e.episode + 1u);
        prep_store<NOISE>(p, rp, o.prep, stride, i, (uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)e.episode + 2u);
      }
    } else {                                             // keep the row this env already has
#pragma unroll
      for (int k = 0; k < S2D_OBS_DIM; ++k) ob.o[k] = o.obs[i * S2D_OBS_DIM + k];
    }
  }
  int64_t rows = n - wave_first; if (rows > kWave) rows = kWave;
  store_obs_tile(lds[wv], ob, lane, in_range, o.obs + wave_first * S2D_OBS_DIM, (int)rows * S2D_OBS_DIM);
}

// The refill workgroups of s2d_step / s2d_step_k (see StepOut::prep): episode e + 2 into slot e & 1 where it is missing -- never the
// slot that holds episode e + 1, the only one a main wave of the same launch takes a reset from.
template <bool NOISE>
S2D_DEV void refill_prepared_slots(const S2DHot& p, const S2DRare* __restrict__ rp, float* __restrict__ S, int64_t stride, int64_t n,
                                   const StepOut& o, int lane, uint32_t* scratch) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  bool need = false;
  uint32_t e2 = 0u;
  if (i < n) {
    const uint32_t* const tags = prep_tags(o.prep, stride);
    const uint32_t t0 = tags[i], t1 = tags[stride + i];    // both tags: no load whose address waits for another load
    e2 = reinterpret_cast<const uint32_t*>(S + F_EPISODE * stride)[i] + 2u;
    need = (((e2 & 1u) ? t1 : t0) & kPrepTagMask) != (e2 & kPrepTagMask);
  }
  if (__ballot(need) == 0ull) return;                      // wave-uniform
  const uint64_t gid = (((uint64_t)p.gid_hi << 32) | p.gid_lo) + (uint64_t)i;
  prep_store_coop<NOISE>(p, rp, o.prep, stride, i, (uint32_t)gid, (uint32_t)(gid >> 32), e2, need, lane, scratch);   // the wave draws together
}

template <int MODE, bool NOISE>
__global__ __launch_bounds__(kBlock) void s2d_reach_step_kernel(S2DHot p, const S2DRare* __restrict__ rp,
                                                                float* __restrict__ S, int64_t stride, int64_t n,
                                                                const void* __restrict__ actions, int kind,
                                                                StepOut o, int refill_blocks) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][kObsTile];
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  if ((int)blockIdx.x < refill_blocks) {                   // the FIRST blocks of the grid, so that they start first
    refill_prepared_slots<NOISE>(p, rp, S, stride, n, o, lane, reinterpret_cast<uint32_t*>(&lds[wv][0]));
    return;
  }
  const int main_block = (int)blockIdx.x - refill_blocks;
  const int64_t i = (int64_t)main_block * kBlock + threadIdx.x;
  const int64_t wave_first = i - lane;
  if (wave_first >= n) return;                           // wave-uniform
  const bool active = i < n;
  const bool use_k = uses_policy_step<MODE, NOISE>(kind);
  uint32_t* const kplane = reinterpret_cast<uint32_t*>(S + F_POLICY * stride);
  unsigned long long* const srow = stats_row(o.stats, wave_first);
  const unsigned long long sold = stats_load(srow, lane);
  ObsOut ob;
  int res = 0;
  if (active) {
    Env e;
    env_load(e, S, stride, i);
    uint64_t gid = (((uint64_t)p.gid_hi << 32) | p.gid_lo) + (uint64_t)i;
    uint32_t gl = (uint32_t)gid, gh = (uint32_t)(gid >> 32);
    uint32_t k = 0;
    if (use_k) k = kplane[i];
    // the prepared next episode travels with the state (loaded always, used when the episode ends in this step)
    // (BOTH slots: which one holds episode + 1 depends on the episode word, and a load whose address waits for another load
    // would put a second memory latency on every step)
    float pw0[PS_WORDS], pw1[PS_WORDS];
    uint32_t ptag0 = 0u, ptag1 = 0u;
    if (p.auto_reset) {
      const float* src = o.prep + i;
#pragma unroll
      for (int w = 0; w < PS_WORDS; ++w) { pw0[w] = src[w * stride]; pw1[w] = src[(PS_WORDS + w) * stride]; }
      ptag0 = prep_tags(o.prep, stride)[i]; ptag1 = prep_tags(o.prep, stride)[stride + i];
    }
    U4 quad{0, 0, 0, 0}, squad{0, 0, 0, 0};
    float reward, dir; int done, cmd;
    CmdPrep c = decide<MODE>(p, actions, kind, i, gl, gh, k, true, quad, squad, nullptr, cmd, dir);
    if (kind == S2D_ACT_COMMAND && cmd < 0) {              // S2D_CMD_FREEZE: not part of this cycle -- keep state, outputs and the row
#pragma unroll
      for (int w = 0; w < S2D_OBS_DIM; ++w) ob.o[w] = o.obs[i * S2D_OBS_DIM + w];
    } else {
      // A1: one Soccer2DEnv.step (soccer_2d_env.py:226-269), as step_env(), with the reset served from the prefetched slot
      e.step_number += 1;                                  // reach_ball_env.py:55
      NoiseIn nz{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      if (NOISE) nz = noise_prepare(p, gl, gh, k, S2D_ST_NOISE, cmd == S2D_CMD_TURN);
      float d2 = sim_cycle<NOISE, true>(p, rp, e, cmd, c, nz);   // trainer forces PlayOn each cycle (:242)
      observe_and_check(p, e, d2, ob, done, reward, res);
      if (done && p.auto_reset) {                          // SB3 VecEnv convention
        float* const terminal_row = o.terminal_obs + i * S2D_OBS_DIM;
#pragma unroll
        for (int w = 0; w < S2D_OBS_DIM; ++w) terminal_row[w] = ob.o[w];
        const bool odd = (((uint32_t)e.episode + 1u) & 1u) != 0u;
        const uint32_t ptag = odd ? ptag1 : ptag0;
        float pw[PS_WORDS];
#pragma unroll
        for (int w = 0; w < PS_WORDS; ++w) pw[w] = odd ? pw1[w] : pw0[w];
        if ((ptag & kPrepTagMask) == (((uint32_t)e.episode + 1u) & kPrepTagMask)) {   // prepared: a copy + the words that follow from it
          const NextEpisode q = prep_episode(p, rp, pw, ptag);
          episode_begin(e, q);
          const FirstObs f = first_obs(p, q);
#pragma unroll
          for (int w = 0; w < S2D_OBS_DIM; ++w) ob.o[w] = f.o[w];
          e.prev_dist = f.dist; e.prev_angle = f.rel;      // reach_ball_env.py:166: carry seeded
        } else {                                           // slot not (yet) valid: draw it here
          d2 = env_reset<NOISE>(p, rp, e, gl, gh);
          int dn2, r2; float w2;
          observe_and_check(p, e, d2, ob, dn2, w2, r2);    // reach_ball_env.py:166: carry seeded, outputs dropped
        }
      }
      env_store(e, S, stride, i);
      if (use_k) kplane[i] = k + 1u;
      o.reward[i] = reward;
      o.done[i] = (uint8_t)done;
      o.result[i] = (uint8_t)res;
      o.action_dir[i] = dir;
      o.action_cmd[i] = (uint8_t)cmd;
    }
  }
  int64_t rows = n - wave_first; if (rows > kWave) rows = kWave;
  store_obs_tile(lds[wv], ob, lane, active, o.obs + wave_first * S2D_OBS_DIM, (int)rows * S2D_OBS_DIM);
  stats_store(srow, lane, sold, wave_first == 0 ? (unsigned long long)n : 0ull, wave_count(active && res == S2D_RESULT_GOAL),
              wave_count(active && res == S2D_RESULT_OUT), wave_count(active && res == S2D_RESULT_TIMEOUT));
}


// s2d_step_k: K cycles of the per-step API in ONE launch (SURVEY 8b's s2d_step_k; a learner that supplies its actions for K steps --
// action repeat, open-loop chunks -- pays the launch and the state round trip once).  The per-step kernel's shape: one wave per 64
// envs, no prologue, resets served from the persistent prepared slots (refill workgroups in front of the grid) -- at most ONE per env
// and launch from a slot (the slot of episode e + 2 may be rewritten by this launch's refill workgroups while it is read), later
// ones are drawn inline.  Per-step outputs go to the caller's record [K][N] (any array may be NULL), the last step's also to the arena.
template <int MODE, bool NOISE>
__global__ __launch_bounds__(kBlock) void s2d_reach_step_k_kernel(S2DHot p, const S2DRare* __restrict__ rp,
                                                                  float* __restrict__ S, int64_t stride, int64_t n, int n_steps,
                                                                  const void* __restrict__ actions, int kind, RolloutOut ro,
                                                                  StepOut o, int refill_blocks) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][kObsTile];
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  if ((int)blockIdx.x < refill_blocks) {
    refill_prepared_slots<NOISE>(p, rp, S, stride, n, o, lane, reinterpret_cast<uint32_t*>(&lds[wv][0]));
    return;
  }
  const int64_t i = (int64_t)((int)blockIdx.x - refill_blocks) * kBlock + threadIdx.x;
  const int64_t wave_first = i - lane;
  if (wave_first >= n) return;                           // wave-uniform
  const bool active = i < n;
  int64_t rows = n - wave_first; if (rows > kWave) rows = kWave;
  const int valid = (int)rows * S2D_OBS_DIM;
  const bool use_k = uses_policy_step<MODE, NOISE>(kind);
  uint32_t* const kplane = reinterpret_cast<uint32_t*>(S + F_POLICY * stride);
  unsigned long long* const srow = stats_row(o.stats, wave_first);
  const unsigned long long sold = stats_load(srow, lane);
  Env e{};
  uint32_t gl = 0, gh = 0, k0 = 0;
  float pw0[PS_WORDS], pw1[PS_WORDS];
  uint32_t ptag0 = 0u, ptag1 = 0u;
#pragma unroll
  for (int w = 0; w < PS_WORDS; ++w) { pw0[w] = 0.0f; pw1[w] = 0.0f; }
  if (active) {
    env_load(e, S, stride, i);
    const uint64_t gid = (((uint64_t)p.gid_hi << 32) | p.gid_lo) + (uint64_t)i;
    gl = (uint32_t)gid; gh = (uint32_t)(gid >> 32);
    if (use_k) k0 = kplane[i];
    if (p.auto_reset) {                                    // both slots travel with the state (see s2d_reach_step_kernel)
      const float* src = o.prep + i;
#pragma unroll
      for (int w = 0; w < PS_WORDS; ++w) { pw0[w] = src[w * stride]; pw1[w] = src[(PS_WORDS + w) * stride]; }
      ptag0 = prep_tags(o.prep, stride)[i]; ptag1 = prep_tags(o.prep, stride)[stride + i];
    }
  }
  ObsOut ob;
  U4 quad{0, 0, 0, 0}, squad{0, 0, 0, 0};
  float reward = 0.0f, dir = 0.0f; int done = 0, cmd = 0, res = 0;
  bool slot_used = false;
  unsigned int c1 = 0, c2 = 0, c3 = 0;                     // wave-uniform episode counters
  int64_t row = 0;
  for (int t = 0; t < n_steps; ++t, row += n) {
    res = 0;
    if (active) {
      const uint32_t k = k0 + (uint32_t)t;
      const CmdPrep c = decide<MODE>(p, actions, kind, row + i, gl, gh, k, t == 0 || (k & 3u) == 0u, quad, squad, ro.action, cmd, dir);
      e.step_number += 1;                                  // reach_ball_env.py:55
      NoiseIn nz{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
      if (NOISE) nz = noise_prepare(p, gl, gh, k, S2D_ST_NOISE, cmd == S2D_CMD_TURN);
      float d2 = sim_cycle<NOISE, true>(p, rp, e, cmd, c, nz);   // trainer forces PlayOn each cycle (soccer_2d_env.py:242)
      observe_and_check(p, e, d2, ob, done, reward, res);
      if (done && p.auto_reset) {                          // SB3 VecEnv convention
        float* const terminal_row = o.terminal_obs + i * S2D_OBS_DIM;
#pragma unroll
        for (int w = 0; w < S2D_OBS_DIM; ++w) terminal_row[w] = ob.o[w];
        const bool odd = (((uint32_t)e.episode + 1u) & 1u) != 0u;
        const uint32_t ptag = odd ? ptag1 : ptag0;
        // only the FIRST reset of a launch may come from a slot: it wants episode e + 1, whose slot this launch's refill workgroups
        // leave alone; a later one wants e + 2, the slot they may be writing while it was loaded above
        const bool first_reset = !slot_used;
        slot_used = true;
        if (first_reset && (ptag & kPrepTagMask) == (((uint32_t)e.episode + 1u) & kPrepTagMask)) {   // prepared: a copy
          float pw[PS_WORDS];
#pragma unroll
          for (int w = 0; w < PS_WORDS; ++w) pw[w] = odd ? pw1[w] : pw0[w];
          const NextEpisode q = prep_episode(p, rp, pw, ptag);
          episode_begin(e, q);
          const FirstObs f = first_obs(p, q);
#pragma unroll
          for (int w = 0; w < S2D_OBS_DIM; ++w) ob.o[w] = f.o[w];
          e.prev_dist = f.dist; e.prev_angle = f.rel;      // reach_ball_env.py:166: carry seeded
        } else {                                           // no valid slot (left): draw it here
          d2 = env_reset<NOISE>(p, rp, e, gl, gh);
          int dn2, r2; float w2;
          observe_and_check(p, e, d2, ob, dn2, w2, r2);    // reach_ball_env.py:166: carry seeded, outputs dropped
        }
      }
      if (ro.reward) ro.reward[row + i] = reward;
      if (ro.done) ro.done[row + i] = (uint8_t)done;
      if (ro.result) ro.result[row + i] = (uint8_t)res;
    }
    c1 += wave_count(active && res == S2D_RESULT_GOAL); c2 += wave_count(active && res == S2D_RESULT_OUT);
    c3 += wave_count(active && res == S2D_RESULT_TIMEOUT);
    if (ro.obs) store_obs_tile(lds[wv], ob, lane, active, ro.obs + (row + wave_first) * S2D_OBS_DIM, valid);
  }
  if (active) {
    env_store(e, S, stride, i);
    if (use_k) kplane[i] = k0 + (uint32_t)n_steps;
    o.reward[i] = reward; o.done[i] = (uint8_t)done; o.result[i] = (uint8_t)res;
    o.action_dir[i] = dir; o.action_cmd[i] = (uint8_t)cmd;
  }
  store_obs_tile(lds[wv], ob, lane, active, o.obs + wave_first * S2D_OBS_DIM, valid);
  stats_store(srow, lane, sold, wave_first == 0 ? (unsigned long long)n * (unsigned long long)n_steps : 0ull, c1, c2, c3);
}

template <int MODE, bool NOISE>
__global__ __launch_bounds__(kBlock) void s2d_reach_rollout_kernel(S2DHot p_sgpr, const S2DRare* __restrict__ rp,
                                                                   float* __restrict__ S, int64_t stride, int64_t n,
                                                                   int n_steps, const void* __restrict__ actions,
                                                                   int kind, RolloutOut ro, StepOut o) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][kObsTile];
  __shared__ PrepTile prep[kWavesPerBlock];
  __shared__ float4 act_lut[kWavesPerBlock][kWave];        // decoded commands of a small discrete action space (per wave)
  __shared__ float ep_lds[S2D_TAB_MAX];                    // dash-only fast path (shared by the block's waves): effort * power
  __shared__ float2 sc_lut[361];                           //   by step number, (sin, cos) of the whole degrees -180 .. 180
  const S2DHot p = hot_in_vgprs(p_sgpr);
  const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t wave_first = i - lane;
  const S2DTables* const tb = tables_of(rp);
  const int tab_len = MODE != S2D_MODE_TURN4 ? rp->tab_len : 0;
  if (tab_len > 0) {                                       // every wave of the block helps, before any of them may leave
    for (int k = threadIdx.x; k < tab_len; k += kBlock) ep_lds[k] = tb->ep[k];
    for (int k = threadIdx.x; k <= 360; k += kBlock) {
      float sn, cs;
      sincos_deg((float)(k - 180), sn, cs);
      sc_lut[k] = make_float2(sn, cs);
    }
    __syncthreads();
  }
  if (wave_first >= n) return;
  const bool active = i < n;
  int64_t rows = n - wave_first; if (rows > kWave) rows = kWave;
  const int valid = (int)rows * S2D_OBS_DIM;
  const bool use_k = uses_policy_step<MODE, NOISE>(kind);
  uint32_t* const kplane = reinterpret_cast<uint32_t*>(S + F_POLICY * stride);
  Env e;
  uint32_t gl = 0, gh = 0, k0 = 0;
  if (active) {
    env_load(e, S, stride, i);
    if (use_k) k0 = kplane[i];
    uint64_t gid = (((uint64_t)p.gid_hi << 32) | p.gid_lo) + (uint64_t)i;
    gl = (uint32_t)gid; gh = (uint32_t)(gid >> 32);
    // Consume every loaded word once BEFORE the loop: the s_waitcnt for the prologue loads
    // is then placed here and not inside the loop body, where (vmcnt being in-order) it would
    // also wait for the previous iteration's stores to be acknowledged.
    asm volatile("" ::"v"(e.px), "v"(e.py), "v"(e.vx), "v"(e.vy), "v"(e.body), "v"(e.stamina), "v"(e.effort),
                 "v"(e.recovery), "v"(e.capacity), "v"(e.bx), "v"(e.by), "v"(e.bvx), "v"(e.bvy), "v"(e.prev_dist),
                 "v"(e.prev_angle), "v"(e.step_number), "v"(e.cycle), "v"(k0));
  }
  // this wave's envs all sit on the stamina table and have whole-degree body angles? (see the pipeline kernel's simulate wave)
  bool fast = false;
  if (tab_len > 0) {
    bool ok = true;
    if (active) {
      const int sn = e.step_number;
      ok = sn >= 0 && sn < tab_len;
      const int q = ok ? sn : 0;
      ok = ok && e.stamina == tb->stamina[q] && e.effort == tb->effort[q] && e.recovery == tb->recovery[q] &&
           e.capacity == tb->capacity[q] && e.body == rintf(e.body) && fabsf(e.body) <= 180.0f;
    }
    fast = __ballot(active && !ok) == 0ull;
  }
  ObsOut ob;
  float reward = 0.0f, dir = 0.0f; int done = 0, res = 0, cmd = 0;
  unsigned int cnt1 = 0, cnt2 = 0, cnt3 = 0;
  float* const term_row = o.terminal_obs + i * S2D_OBS_DIM;
  U4 quad{0, 0, 0, 0}, squad{0, 0, 0, 0};
  bool have_prep = false;
  uint32_t* const coop_scratch = reinterpret_cast<uint32_t*>(&lds[wv][0]);   // the observation tile is idle between cycles
  if (p.auto_reset) {                                      // full wave, drawn together (reset_sample_coop)
    prep_fill_coop<NOISE>(p, rp, prep[wv], lane, active ? reset_key(e) : 0u, gl, gh, active, coop_scratch);
    have_prep = active;
  }
  int n_missing = 0;                                       // wave-uniform: lanes whose prepared sample is used up
  int64_t row = 0;
  // in-engine policy of a small discrete action space: one table entry per action (see the pipeline kernel's P-wave).
  // Measured at 1 M envs: +6 % with noise on, -1 % with noise off -- so only the noise build uses it here.
  const bool lut = NOISE && MODE == S2D_MODE_DISCRETE && kind == S2D_ACT_RANDOM && p.n_actions <= kWave;
  if (lut && lane < p.n_actions) {
    int c0; float pw, d0;
    action_map<MODE>(p, Action4{(float)lane, 0.0f, 0.0f, 0.0f}, 0.0f, c0, pw, d0);
    const CmdPrep c = cmd_prepare(p, c0, pw, d0);
    act_lut[wv][lane] = make_float4(c.power, c.dir, c.dir_rate, d0);
  }
  for (int t = 0; t < n_steps; ++t, row += n) {
    res = 0;
    if (n_missing >= kRefillMin) {                         // batched refill (wave-uniform counter: no ballot per cycle)
      // (sequential draw: with ~8-20 lanes to serve, the cooperative loop's three rounds cost what their ~4 tries cost, and its
      // registers cost the 1 M-env launch 3 %)
      if (active && !have_prep) { prep_fill<NOISE>(p, rp, prep[wv], lane, e, gl, gh); have_prep = true; }
      n_missing = 0;
    }
    if (active) {
      const uint32_t k = k0 + (uint32_t)t;
      CmdPrep c;
      if (lut) {
        if (t == 0 || (k & 3u) == 0u) quad = policy_quad(p, gl, gh, k, S2D_ST_POLICY);
        const int a = (int)rnd_below(quad_word(quad, k), (uint32_t)p.n_actions);
        if (ro.action) static_cast<int32_t*>(ro.action)[row + i] = a;
        const float4 e4 = act_lut[wv][a];
        c = CmdPrep{e4.x, e4.y, e4.z}; dir = e4.w; cmd = S2D_CMD_DASH;
      } else {
        c = decide<MODE>(p, actions, kind, row + i, gl, gh, k, t == 0 || (k & 3u) == 0u, quad, squad, ro.action, cmd, dir);
      }
      if (fast) step_env<NOISE, true>(p, rp, e, gl, gh, k, cmd, c, ob, reward, done, res, term_row, &prep[wv], lane, have_prep, ep_lds, sc_lut);
      else step_env<NOISE, false>(p, rp, e, gl, gh, k, cmd, c, ob, reward, done, res, term_row, &prep[wv], lane, have_prep);
      if (ro.reward) ro.reward[row + i] = reward;
      if (ro.done) ro.done[row + i] = (uint8_t)done;
      if (ro.result) ro.result[row + i] = (uint8_t)res;
      cnt1 += res == S2D_RESULT_GOAL; cnt2 += res == S2D_RESULT_OUT; cnt3 += res == S2D_RESULT_TIMEOUT;
    }
    // samples consumed by this cycle's resets -- counted by EVERY lane of the wave (also those past the last env): wave-uniform
    if (p.auto_reset) n_missing += __popcll(__ballot(active && done != 0));
    if (ro.obs) store_obs_tile(lds[wv], ob, lane, active, ro.obs + (row + wave_first) * S2D_OBS_DIM, valid);
  }
  if (active) {
    if (fast) {                                            // the stamina words the fast path did not carry
      const int q = e.step_number;
      e.stamina = tb->stamina[q]; e.effort = tb->effort[q]; e.recovery = tb->recovery[q]; e.capacity = tb->capacity[q];
    }
    env_store(e, S, stride, i);
    if (use_k) kplane[i] = k0 + (uint32_t)n_steps;
    o.reward[i] = reward; o.done[i] = (uint8_t)done; o.result[i] = (uint8_t)res;
    o.action_dir[i] = dir; o.action_cmd[i] = (uint8_t)cmd;
  }
  store_obs_tile(lds[wv], ob, lane, active, o.obs + wave_first * S2D_OBS_DIM, valid);
  // wave-level reduction of the per-lane episode counters (butterfly: every lane ends with the sums), then the wave's row
  if (!active) { cnt1 = cnt2 = cnt3 = 0; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    cnt1 += __shfl_xor(cnt1, off); cnt2 += __shfl_xor(cnt2, off); cnt3 += __shfl_xor(cnt3, off);
  }
  unsigned long long* const srow = stats_row(o.stats, wave_first);
  stats_store(srow, lane, stats_load(srow, lane), wave_first == 0 ? (unsigned long long)n * (unsigned long long)n_steps : 0ull, cnt1, cnt2, cnt3);
}

// ------------------------------------------------------------------------------------------
// Wave-specialised rollout: a software pipeline of four waves through LDS.
//
// At N = 65 536 the unified kernel leaves ONE wave per SIMD: its ~560 instructions per cycle are a
// single dependent stream, and one wave issues at most one instruction per ~5 clocks
// (profiles/r01/instr_rate_gfx950.txt).  Here every group of 64 envs is a workgroup of four waves,
// each one stage of the cycle:
//   P-wave (policy):   action of step t (caller's, or Philox policy) -> decoded command, the
//                      command-only half of the dash (clamps, direction rate), action record
//   S-wave (simulate): command -> dash/turn -> integrate -> stamina -> done test -> reset (a copy)
//   A-wave (agent):    player half of the observation (angle to the ball), distance, reward,
//                      labels, reward / done / result stores, episode counters, reward carry
//   B-wave (ball):     ball half of the observation (speed, direction), and the transposed
//                      observation block of the PREVIOUS step streamed out of LDS
// In iteration s the P-wave works on step s, the S-wave on step s-1, the A- and B-waves on step s-2
// (the B-wave also stores the observations of step s-3); the hand-offs are double-buffered in LDS and
// ONE s_barrier per iteration separates them.  There is no feedback edge: policy draws are keyed by
// policy_step (not by the cycle, which resets advance), the S-wave evaluates the done conditions
// itself, the A-wave owns the reward carry.
//
// Resets.  What a reset leaves behind (trainer moves + recover + the command-less cycle + the first
// observation and reward carry) is a function of (env id, episode index) alone, and drawing it -- two or more
// Philox blocks, a rejection loop with sine / cosine, one simulator cycle, two atan2 -- costs ~1 us of a
// single wave.  Round 1 did that on the simulating wave whenever an episode ended (in ~30 % of the
// iterations), i.e. on the critical path of the whole group.  Now the next kSlots episodes of every env
// are prepared BEFORE the loop by the three waves that would otherwise idle while the pipeline fills
// (policy, agent, ball: one slot each), into LDS; a reset in the loop is a copy of 13 words by the
// simulating wave and of the prepared first observation by the observing waves.  Only an env that ends
// more than kSlots episodes within one launch prepares inline (and publishes through the slot it used
// longest ago, which every reader has left at least two barriers earlier).
// The arithmetic is the same functions in the same order as in the unified kernel: results are bit-identical.
// Issue priority of the four role waves of a group (s_setprio): a SIMD holds one wave of each role (of four different groups)
// and the arbiter should prefer them by their slack.  Noise off: simulate / agent / ball / policy = 1/2/3/0 (the ball wave, which
// streams the observations, on top); noise on: 3/2/2/1.  The sweeps behind these (in us and in clocks): profiles/r03/ws_prio_sweeps.txt.
// Overridable for experiments.
// noise on (simulate / agent / ball / policy) 3/2/2/1: tuned in round 3, when the simulating wave was the long one (1 482 busy clocks);
// re-measured in round 4, when the policy wave is (1 233 against 879): still the fastest of six assignments
// (profiles/r04/ab_noise_priorities.txt)
#ifndef S2D_NPRIO_S
#define S2D_NPRIO_S 3
#endif
#ifndef S2D_NPRIO_A
#define S2D_NPRIO_A 2
#endif
#ifndef S2D_NPRIO_B
#define S2D_NPRIO_B 2
#endif
#ifndef S2D_NPRIO_P
#define S2D_NPRIO_P 1
#endif
#ifndef S2D_PRIO_S
#define S2D_PRIO_S 1
#endif
#ifndef S2D_PRIO_A
#define S2D_PRIO_A 2
#endif
#ifndef S2D_PRIO_B
#define S2D_PRIO_B 3
#endif

// experiment build (-DS2D_STAMPS): per role wave, the busy clocks (barrier release -> arrival at the next barrier), the clocks of its
// whole loop, the 100 MHz real-time stamps of its begin and end, and its HW_ID / XCC_ID words (where it ran), written by lane 0 into terminal_obs row wave_first + 2 * role
#ifdef S2D_STAMPS
#define WS_STAMP_DECL uint64_t st_busy = 0, st_t0 = __builtin_amdgcn_s_memtime(); const uint64_t st_begin = st_t0, st_rt0 = __builtin_amdgcn_s_memrealtime()
#define WS_BARRIER() do { st_busy += __builtin_amdgcn_s_memtime() - st_t0; __syncthreads(); st_t0 = __builtin_amdgcn_s_memtime(); } while (0)
#define WS_STAMP_STORE() do { if (lane == 0) { float* q_ = o.terminal_obs + (wave_first + 2 * role) * S2D_OBS_DIM; \
    q_[0] = (float)st_busy; q_[1] = (float)(__builtin_amdgcn_s_memtime() - st_begin); q_[2] = (float)(st_rt0 & 0xffffff); \
    q_[3] = (float)(__builtin_amdgcn_s_memrealtime() & 0xffffff); \
    q_[4] = __int_as_float((int)__builtin_amdgcn_s_getreg((31 << 11) | 4)); q_[5] = __int_as_float((int)__builtin_amdgcn_s_getreg((31 << 11) | 20)); } } while (0)
#else
#define WS_STAMP_DECL do {} while (0)
#define WS_BARRIER() __syncthreads()
#define WS_STAMP_STORE() do {} while (0)
#endif

// (Dealing work by slack -- the ball's direction computed by the policy wave, done / result derived by the ball wave, the observation
// block's tail streamed by the policy wave, a fifth wave for all stores -- was built and measured in round 4: the four waves of a SIMD
// share one issue port that this mix keeps busy; profiles/r04/ws_stamps_balance.txt, ws_store_assignment.txt, ws_store_wave.txt.)
// REC: what the kernel knows about the record at compile time.  0: nothing (every array may be absent, `nt` is a run-time flag);
// 1 / 2: all five arrays are there and the stores are plain / non-temporal.  The presence tests and the nt selection are
// wave-uniform branches, eleven of them per cycle in the two waves that store -- and those are the long waves when noise is off:
// with them compiled out the slowest workgroup of a 256-cycle launch counts 296 k clocks instead of 321 k
// (profiles/r03/ws_static_record.txt).  Instantiated for the discrete-action, noise-off configuration (the DQN script's).
template <int MODE, bool NOISE, int REC = 0>
__global__ __launch_bounds__(kWsBlock, 4) void s2d_reach_rollout_ws_kernel(S2DHot p_sgpr, const S2DRare* __restrict__ rp,
                                                                        float* __restrict__ S, int64_t stride,
                                                                        int64_t n, int n_steps,
                                                                        const void* __restrict__ actions, int kind,
                                                                        RolloutOut ro, StepOut o) {
  constexpr int kActWords = NOISE ? (int)WA_WORDS : (int)WA_NPM;
  // decoded command (+ prepared noise) of step t in slot t mod 3: the policy wave runs TWO steps ahead of the simulating wave, which
  // fetches the command of its NEXT step at the start of an iteration and, at the iteration's end -- when the body angle after a
  // possible reset is known --, issues the two table reads that depend on it (dash fast path), so that an iteration starts on
  // registers: ~200 clocks of LDS round trips off the longest chain of the group (profiles/r04/ws_stamps_prefetch.txt)
  __shared__ float act[3][kActWords][kWave];
  __shared__ float snap[2][WS_WORDS][kWave];               // post-cycle snapshot of step t, double-buffered
  __shared__ float slots[kSlots][SL_WORDS][kWave];         // prepared episodes first_ep + k of every lane (see above)
  __shared__ __attribute__((aligned(16))) float tile[2][kObsTile];   // observation rows of step t, double-buffered
  __shared__ float4 act_lut[kWave];                        // decoded commands of a small discrete action space
  __shared__ float ep_lds[S2D_TAB_MAX];                    // dash-only fast path: effort * power by step number
  __shared__ float2 sc_lut[361];                           //   and (sin, cos) of the whole degrees -180 .. 180
  const int lane = threadIdx.x & (kWave - 1);
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);   // 0 policy, 1 simulate, 2 agent, 3 ball
  const bool nt = REC == 2 || (REC == 0 && ro.nt != 0);     // the small arrays (action, reward, done, result)
  // (One policy for all five arrays.  In the store pattern alone plain observation stores next to nt done / result stores are 3 %
  // faster than everything nt -- profiles/r04/store_pattern5.txt -- but in this kernel they are 2-4 % slower: profiles/r04/ab_obs_plain.txt.)
  const int64_t wave_first = (int64_t)blockIdx.x * kWave;
  const int64_t i = wave_first + lane;
  const bool active = i < n;
  int64_t rows = n - wave_first; rows = rows > kWave ? kWave : rows;
  const int valid = (int)rows * S2D_OBS_DIM;
  const int n_iter = n_steps + 3;

  // ---- before the loop: the three waves that idle while the pipeline fills prepare one future episode each
  if (role != 1 && p_sgpr.auto_reset) {                    // wave-uniform: the wave draws together (reset_sample_coop)
    const uint64_t gid = (((uint64_t)p_sgpr.gid_hi << 32) | p_sgpr.gid_lo) + (uint64_t)i;
    uint32_t ep0 = 0u;
    if (active) ep0 = reinterpret_cast<const uint32_t*>(S + F_EPISODE * stride)[i];
    const int k = role == 0 ? 0 : role - 1;
    slot_fill_coop<NOISE>(p_sgpr, rp, slots[k], lane, (uint32_t)gid, (uint32_t)(gid >> 32), ep0 + 1u + (uint32_t)k, active,
                          reinterpret_cast<uint32_t*>(&tile[0][0]) + role * kWave);   // the observation tiles are idle before the loop
  }

  if (role == 0) {
    if constexpr (NOISE) __builtin_amdgcn_s_setprio(S2D_NPRIO_P);   // three Philox blocks per four cycles + the noise table reads
    // ------------------------------------------------------------------ P-wave
    const S2DHot& p = p_sgpr;
    const bool use_k = uses_policy_step<MODE, NOISE>(kind);
    uint32_t* const kplane = reinterpret_cast<uint32_t*>(S + F_POLICY * stride);
    uint32_t gl = 0, gh = 0, k0 = 0;
    if (active) {
      if (use_k) k0 = kplane[i];
      uint64_t gid = (((uint64_t)p.gid_hi << 32) | p.gid_lo) + (uint64_t)i;
      gl = (uint32_t)gid; gh = (uint32_t)(gid >> 32);
    }
    U4 quad{0, 0, 0, 0}, squad{0, 0, 0, 0}, nblk{0, 0, 0, 0};
    float dir = 0.0f; int cmd = 0;
    int64_t row = 0;
    // The in-engine policy of a small discrete action space picks one of n_actions decoded commands: the action map and
    // the command-only half of the dash are evaluated once per launch into a table (one LDS read per step instead of
    // ~27 VALU instructions; the entries are the values decide<> computes, so nothing changes bit-wise).
    const bool lut = MODE == S2D_MODE_DISCRETE && kind == S2D_ACT_RANDOM && p.n_actions <= kWave;
    if (lut && lane < p.n_actions) {
      int c0; float pw, d0;
      action_map<MODE>(p, Action4{(float)lane, 0.0f, 0.0f, 0.0f}, 0.0f, c0, pw, d0);
      const CmdPrep c = cmd_prepare(p, c0, pw, d0);
      act_lut[lane] = make_float4(c.power, c.dir, c.dir_rate, d0);
    }
    __syncthreads();                                       // prepared episodes published
    WS_STAMP_DECL;
    int wslot = 0;                                         // ring slot of the step computed next (step mod 3)
    auto policy_step = [&](int t) {                        // command of step t -> act[t mod 3]
      if (active) {
        const uint32_t k = k0 + (uint32_t)t;
        CmdPrep c;
        if (lut) {
          if (t == 0 || (k & 3u) == 0u) quad = policy_quad(p, gl, gh, k, S2D_ST_POLICY);
          const int a = (int)rnd_below(quad_word(quad, k), (uint32_t)p.n_actions);
          if (REC != 0 || ro.action) rec_store(static_cast<int32_t*>(ro.action) + row + i, (int32_t)a, nt);
          const float4 e4 = act_lut[a];
          c = CmdPrep{e4.x, e4.y, e4.z}; dir = e4.w; cmd = S2D_CMD_DASH;
        } else {
          c = decide<MODE>(p, actions, kind, row + i, gl, gh, k, t == 0 || (k & 3u) == 0u, quad, squad, ro.action, cmd, dir);
        }
        const int b = wslot;
        if (MODE == S2D_MODE_TURN4) act[b][WA_CMD][lane] = __int_as_float(cmd);
        act[b][WA_POWER][lane] = c.power;
        act[b][WA_DIR][lane] = c.dir; act[b][WA_RATE][lane] = c.dir_rate;
        if constexpr (NOISE) {                             // the state-independent half of this cycle's noise; the sine / cosine of
          // the whole-degree directions are entries of the table the simulating wave built before the first barrier
          const NoiseWords nw = noise_words(p, gl, gh, k, S2D_ST_NOISE, cmd == S2D_CMD_TURN, nblk, t == 0 || (k & 1u) == 0u);
          const float2 ps = sc_lut[noise_dir_index(nw.wp)], bs = sc_lut[noise_dir_index(nw.wb)];
          act[b][WA_NPM][lane] = noise_mag(nw.wp); act[b][WA_NPS][lane] = ps.x; act[b][WA_NPC][lane] = ps.y;
          act[b][WA_NBM][lane] = noise_mag(nw.wb); act[b][WA_NBS][lane] = bs.x; act[b][WA_NBC][lane] = bs.y;
          if (MODE == S2D_MODE_TURN4) act[b][WA_NTU][lane] = nw.tu;
        }
        row += n;
      }
      wslot = wslot == 2 ? 0 : wslot + 1;
    };
    for (int s = 0; s < n_iter; ++s) {                     // iteration s: step s + 1 (iteration 0: steps 0 and 1)
      if (s == 0 && n_steps > 0) policy_step(0);
      if (s + 1 < n_steps) policy_step(s + 1);
      WS_BARRIER();
    }
    WS_STAMP_STORE();
    if (active) {
      if (use_k) kplane[i] = k0 + (uint32_t)n_steps;
      o.action_dir[i] = dir; o.action_cmd[i] = (uint8_t)cmd;
    }
  } else if (role == 1) {
    // ------------------------------------------------------------------ S-wave
    if constexpr (NOISE) __builtin_amdgcn_s_setprio(S2D_NPRIO_S);
    else __builtin_amdgcn_s_setprio(S2D_PRIO_S);
    const S2DHot p = hot_in_vgprs(p_sgpr);
    Env e;
    uint32_t gl = 0, gh = 0;
    int nth = 0, j = 0;                                    // episodes this lane began in this launch; slot of the next one (nth mod kSlots)
    if (active) {
      env_load(e, S, stride, i);
      uint64_t gid = (((uint64_t)p.gid_hi << 32) | p.gid_lo) + (uint64_t)i;
      gl = (uint32_t)gid; gh = (uint32_t)(gid >> 32);
      asm volatile("" ::"v"(e.px), "v"(e.py), "v"(e.vx), "v"(e.vy), "v"(e.body), "v"(e.stamina), "v"(e.effort),
                   "v"(e.recovery), "v"(e.capacity), "v"(e.bx), "v"(e.by), "v"(e.bvx), "v"(e.bvy),
                   "v"(e.step_number), "v"(e.cycle), "v"(e.episode));
    }
    // Dash-only fast path (s2d_device.h, S2DTables): taken by a group whose envs all sit on the stamina table and have
    // whole-degree body angles -- true for every state the engine itself produces in the dash-only modes; states loaded
    // from elsewhere (or the turning mode) run the generic loop with the same results.
    const S2DTables* const tb = tables_of(rp);
    const int tab_len = MODE != S2D_MODE_TURN4 ? rp->tab_len : 0;
    bool fast = false;
    if (tab_len > 0) {
      bool ok = true;
      if (active) {
        const int sn = e.step_number;
        ok = sn >= 0 && sn < tab_len;
        const int q = ok ? sn : 0;
        ok = ok && e.stamina == tb->stamina[q] && e.effort == tb->effort[q] && e.recovery == tb->recovery[q] &&
             e.capacity == tb->capacity[q] && e.body == rintf(e.body) && fabsf(e.body) <= 180.0f;
      }
      fast = __ballot(active && !ok) == 0ull;
      if (fast)
        for (int k = lane; k < tab_len; k += kWave) ep_lds[k] = tb->ep[k];
    }
    if (fast || NOISE) {                                   // (sin, cos) of the whole degrees: dash directions and noise directions
      for (int k = lane; k <= 360; k += kWave) {
        float sn, cs;
        sincos_deg((float)(k - 180), sn, cs);
        sc_lut[k] = make_float2(sn, cs);
      }
    }
    const ResetStamina rst = reset_stamina(p, rp);         // what every reset leaves in the stamina words (not kept in the slots)
    __syncthreads();                                       // prepared episodes (and this wave's tables) published
    WS_STAMP_DECL;
    auto loop = [&](auto fast_tag) {
      constexpr bool FAST = decltype(fast_tag)::value;
      struct ActRegs { float cmd, power, dir, rate, npm, nps, npc, nbm, nbs, nbc, ntu; };
      ActRegs cur{}, nxt{};                                // command of the step simulated in this iteration / in the next one
      float ep_cur = 0.0f; float2 sc_cur = make_float2(0.0f, 0.0f);   // dash fast path: the two table entries of `cur`
      int rslot = 1;                                       // ring slot of the step fetched next (wave-uniform; step 0 is fetched by the filling iteration)
      auto fetch = [&](int slot) {
        ActRegs r{};
        if (MODE == S2D_MODE_TURN4) r.cmd = act[slot][WA_CMD][lane];
        if (!FAST) r.power = act[slot][WA_POWER][lane];
        r.dir = act[slot][WA_DIR][lane]; r.rate = act[slot][WA_RATE][lane];
        if constexpr (NOISE) {
          r.npm = act[slot][WA_NPM][lane]; r.nps = act[slot][WA_NPS][lane]; r.npc = act[slot][WA_NPC][lane];
          r.nbm = act[slot][WA_NBM][lane]; r.nbs = act[slot][WA_NBS][lane]; r.nbc = act[slot][WA_NBC][lane];
          if (MODE == S2D_MODE_TURN4) r.ntu = act[slot][WA_NTU][lane];
        }
        return r;
      };
      auto lookups = [&]() {                               // what the dash of `cur` needs of the CURRENT state (after a possible reset)
        if constexpr (FAST) {
          ep_cur = ep_lds[e.step_number];                  // effort * power of the dash at this step number
          sc_cur = sc_lut[(int)norm_deg(e.body + cur.dir) + 180];
        }
      };
      // stretches: s = 1 fills (fetches its own command: the one exposed round trip), 2 <= s < n_steps steady, s = n_steps has no next step
      auto simulate_iteration = [&](int s, auto steady_tag) {
        constexpr bool STEADY = decltype(steady_tag)::value;
        if ((STEADY || (s >= 1 && s <= n_steps)) && active) {   // step s - 1
          if (!STEADY && s == 1) { cur = fetch(0); lookups(); }
          const bool more = STEADY || s < n_steps;
          if (more) nxt = fetch(rslot);                    // step s: lands while step s - 1 is computed
          const int b = (s - 1) & 1;
          int cmd = S2D_CMD_DASH;                          // only the turning mode has another command
          if (MODE == S2D_MODE_TURN4) cmd = __float_as_int(cur.cmd);
          NoiseIn nz{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
          if constexpr (NOISE) {
            nz = NoiseIn{cur.npm, cur.nps, cur.npc, cur.nbm, cur.nbs, cur.nbc, 0.0f};
            if (MODE == S2D_MODE_TURN4) nz.tu = cur.ntu;
          }
          float d2;
          if constexpr (FAST) {
            e.step_number += 1;                            // reach_ball_env.py:55
            d2 = sim_cycle_dash_fast<NOISE>(p, rp, e, ep_cur, cur.rate, sc_cur.x, sc_cur.y, nz);
          } else {
            const CmdPrep c{cur.power, cur.dir, cur.rate};
            e.step_number += 1;                            // reach_ball_env.py:55
            d2 = sim_cycle<NOISE, true>(p, rp, e, cmd, c, nz);
          }
          int flags = judge_sq(p, e.px, e.py, d2, e.step_number);
          const bool took = flags && p.auto_reset;
          snap[b][WS_PX][lane] = e.px; snap[b][WS_PY][lane] = e.py; snap[b][WS_BODY][lane] = e.body;
          snap[b][WS_BX][lane] = e.bx; snap[b][WS_BY][lane] = e.by;
          snap[b][WS_BVX][lane] = e.bvx; snap[b][WS_BVY][lane] = e.bvy;
          snap[b][WS_FLAGS][lane] = __int_as_float(flags | (j << 8));   // bits 8..: the slot holding the next episode
          if (took) {                                      // rare: the prepared episode is a copy
            if (nth >= kSlots)                             // more than kSlots episodes ended in this launch: prepare inline
              slot_fill<NOISE>(p, rp, slots[j], lane, gl, gh, (uint32_t)e.episode + 1u);
            episode_begin(e, slot_take<kWave>(slots[j], lane, rst));
            nth += 1; j = (j + 1 == kSlots) ? 0 : j + 1;
          }
          if (more) { cur = nxt; lookups(); }              // in flight across the barrier
        }
        if (STEADY || (s >= 1 && s < n_steps)) rslot = rslot == 2 ? 0 : rslot + 1;   // (wave-uniform, also for lanes without an env)
        WS_BARRIER();
      };
      int s = 0;
      for (; s < 2 && s < n_iter; ++s) simulate_iteration(s, std::false_type{});
      for (; s < n_steps; ++s) simulate_iteration(s, std::true_type{});
      for (; s < n_iter; ++s) simulate_iteration(s, std::false_type{});
    };
    if (fast) {
      loop(std::true_type{});
      if (active) {                                        // the stamina words the fast loop did not carry
        const int q = e.step_number;
        e.stamina = tb->stamina[q]; e.effort = tb->effort[q]; e.recovery = tb->recovery[q]; e.capacity = tb->capacity[q];
      }
    } else {
      loop(std::false_type{});
    }
    WS_STAMP_STORE();
    if (active) {                                          // prev_dist / prev_angle belong to the A-wave
      S[F_PX * stride + i] = e.px; S[F_PY * stride + i] = e.py;
      S[F_VX * stride + i] = e.vx; S[F_VY * stride + i] = e.vy;
      S[F_BODY * stride + i] = e.body;
      S[F_STAMINA * stride + i] = e.stamina; S[F_EFFORT * stride + i] = e.effort;
      S[F_RECOVERY * stride + i] = e.recovery; S[F_CAPACITY * stride + i] = e.capacity;
      S[F_BX * stride + i] = e.bx; S[F_BY * stride + i] = e.by;
      S[F_BVX * stride + i] = e.bvx; S[F_BVY * stride + i] = e.bvy;
      S[F_STEP * stride + i] = __int_as_float(e.step_number);
      S[F_CYCLE * stride + i] = __int_as_float(e.cycle);
      S[F_EPISODE * stride + i] = __int_as_float(e.episode);
    }
  } else if (role == 2) {
    __builtin_amdgcn_s_setprio(NOISE ? S2D_NPRIO_A : S2D_PRIO_A);
    // ------------------------------------------------------------------ A-wave (player half, reward, labels)
    const S2DHot p = hot_in_vgprs(p_sgpr);                 // no kernarg re-loads (s_load + s_waitcnt) inside the loop
    const bool auto_reset = p_sgpr.auto_reset != 0;
    float prev_dist = 0.0f, prev_angle = 0.0f;
    if (active) {
      prev_dist = S[F_PREV_DIST * stride + i]; prev_angle = S[F_PREV_ANGLE * stride + i];
      asm volatile("" ::"v"(prev_dist), "v"(prev_angle));
    }
    float oa[S2D_OBS_DIM];                                 // only oa[0..3] are produced here
    float reward = 0.0f; int res = 0, done = 0;
    unsigned int cnt1 = 0, cnt2 = 0, cnt3 = 0;
    unsigned long long* const srow = stats_row(o.stats, wave_first);
    const unsigned long long sold = stats_load(srow, lane);  // this group's row of the episode counters (stored after the loop)
    float* const term_row = o.terminal_obs + i * S2D_OBS_DIM;
    int64_t row = 0;
    __syncthreads();                                       // prepared episodes published
    WS_STAMP_DECL;
    auto agent_iteration = [&](int s, auto steady_tag) {   // (three stretches: see the ball wave)
      constexpr bool STEADY = decltype(steady_tag)::value;
      if (STEADY || (s >= 2 && s < n_steps + 2)) {         // step s - 2
        const int b = s & 1;
        res = 0;
        if (active) {
          float px = snap[b][WS_PX][lane], py = snap[b][WS_PY][lane], body = snap[b][WS_BODY][lane];
          float bx = snap[b][WS_BX][lane], by = snap[b][WS_BY][lane];
          const int fw = __float_as_int(snap[b][WS_FLAGS][lane]);
          const int flags = fw & 0xff;
          float dist = hypot2(bx - px, by - py);
          float rel = observe_player(p, px, py, body, bx, by, oa);
          reward = reward_of(prev_dist, prev_angle, dist, rel, flags, res);
          prev_dist = dist; prev_angle = rel;
          done = flags ? 1 : 0;
          if (flags && auto_reset) {                       // rare: terminal row, then the new episode's first obs
            const float (*sl)[kWave] = slots[fw >> 8];
#pragma unroll
            for (int k = 0; k < 4; ++k) term_row[k] = oa[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) oa[k] = sl[SL_FIRST + k][lane];
            prev_dist = sl[SL_DIST][lane]; prev_angle = sl[SL_REL][lane];   // reach_ball_env.py:166 carry seeded
          }
          if (REC != 0 || ro.reward) rec_store(ro.reward + row + i, reward, nt);
          if (REC != 0 || ro.done) rec_store(ro.done + row + i, (uint8_t)done, nt);
          if (REC != 0 || ro.result) rec_store(ro.result + row + i, (uint8_t)res, nt);
          cnt1 += res == S2D_RESULT_GOAL; cnt2 += res == S2D_RESULT_OUT; cnt3 += res == S2D_RESULT_TIMEOUT;
          if (REC != 0 || ro.obs) {                                    // this wave's four words of the row
            float* t = &tile[b][lane * S2D_OBS_DIM];
            t[0] = oa[0]; t[1] = oa[1]; t[2] = oa[2]; t[3] = oa[3];
          }
        }
        row += n;
      }
      WS_BARRIER();
    };
    {
      int s = 0;
      for (; s < 3 && s < n_iter; ++s) agent_iteration(s, std::false_type{});
      for (; s < n_steps + 2; ++s) agent_iteration(s, std::true_type{});
      for (; s < n_iter; ++s) agent_iteration(s, std::false_type{});
    }
    WS_STAMP_STORE();
    if (active) {
      S[F_PREV_DIST * stride + i] = prev_dist; S[F_PREV_ANGLE * stride + i] = prev_angle;
      o.reward[i] = reward; o.done[i] = (uint8_t)done; o.result[i] = (uint8_t)res;
#pragma unroll
      for (int k = 0; k < 4; ++k) o.obs[i * S2D_OBS_DIM + k] = oa[k];      // last observation, player half
    }
    if (!active) { cnt1 = cnt2 = cnt3 = 0; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      cnt1 += __shfl_xor(cnt1, off); cnt2 += __shfl_xor(cnt2, off); cnt3 += __shfl_xor(cnt3, off);
    }
    stats_store(srow, lane, sold, wave_first == 0 ? (unsigned long long)n * (unsigned long long)n_steps : 0ull, cnt1, cnt2, cnt3);
  } else {
    // ------------------------------------------------------------------ B-wave (ball half, observation stream)
    if constexpr (NOISE) __builtin_amdgcn_s_setprio(S2D_NPRIO_B);
    else __builtin_amdgcn_s_setprio(S2D_PRIO_B);
    const S2DHot p = hot_in_vgprs(p_sgpr);                 // no kernarg re-loads inside the loop
    const bool auto_reset = p_sgpr.auto_reset != 0;
    float ob6[S2D_OBS_DIM];                                // only ob6[4..9] are produced here
    float* const term_row = o.terminal_obs + i * S2D_OBS_DIM;
    // every row of this group's observation stream is a whole tile at a 16-byte-aligned address when the first one is and the row
    // stride (n x 40 bytes) keeps it so
    const bool obs_all_vec = valid == kObsTile && ((n * S2D_OBS_DIM * 4) & 15) == 0 &&
                             (reinterpret_cast<uintptr_t>(ro.obs + wave_first * S2D_OBS_DIM) & 15u) == 0;
    __syncthreads();                                       // prepared episodes published
    WS_STAMP_DECL;
    // The loop in three stretches: filling (s < 3), steady (3 <= s < n_steps + 2: every stage of the pipeline has work, the range tests
    // are compiled out -- wave-uniform branches the long waves pay for in every cycle) and draining.  One barrier per iteration in all.
    auto ball_iteration = [&](int s, auto steady_tag) {
      constexpr bool STEADY = decltype(steady_tag)::value;
      if ((STEADY || s >= 3) && (REC != 0 || ro.obs))      // observation block of step s - 3, completed in iteration s - 1
        tile_flush(tile[(s - 1) & 1], lane, ro.obs + ((int64_t)(s - 3) * n + wave_first) * S2D_OBS_DIM, valid, nt, obs_all_vec);
      if ((STEADY || (s >= 2 && s < n_steps + 2)) && active) {   // step s - 2
        const int b = s & 1;
        float bx = snap[b][WS_BX][lane], by = snap[b][WS_BY][lane];
        float bvx = snap[b][WS_BVX][lane], bvy = snap[b][WS_BVY][lane];
        const int fw = __float_as_int(snap[b][WS_FLAGS][lane]);
        observe_ball(p, bx, by, bvx, bvy, ob6);
        if ((fw & 0xff) && auto_reset) {                   // rare: terminal row, then the new episode's first obs
          const float (*sl)[kWave] = slots[fw >> 8];
#pragma unroll
          for (int k = 4; k < S2D_OBS_DIM; ++k) term_row[k] = ob6[k];
#pragma unroll
          for (int k = 4; k < S2D_OBS_DIM; ++k) ob6[k] = sl[SL_FIRST + k][lane];
        }
        if (REC != 0 || ro.obs) {
          float* t = &tile[b][lane * S2D_OBS_DIM];
#pragma unroll
          for (int k = 4; k < S2D_OBS_DIM; ++k) t[k] = ob6[k];
        }
      }
      WS_BARRIER();
    };
    {
      int s = 0;
      for (; s < 3 && s < n_iter; ++s) ball_iteration(s, std::false_type{});
      for (; s < n_steps + 2; ++s) ball_iteration(s, std::true_type{});
      for (; s < n_iter; ++s) ball_iteration(s, std::false_type{});
    }
    WS_STAMP_STORE();
    if (active) {
#pragma unroll
      for (int k = 4; k < S2D_OBS_DIM; ++k) o.obs[i * S2D_OBS_DIM + k] = ob6[k];   // last observation, ball half
    }
  }
}

// derived protobuf-mirroring fields (row T1; idl/service.proto:22-27, 68-86, 181-223)
__global__ __launch_bounds__(kBlock) void s2d_world_model_kernel(const float* __restrict__ S, int64_t stride,
                                                                 int64_t n, S2DWorldModel w) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  float px = S[F_PX * stride + i], py = S[F_PY * stride + i], vx = S[F_VX * stride + i], vy = S[F_VY * stride + i];
  float bx = S[F_BX * stride + i], by = S[F_BY * stride + i], bvx = S[F_BVX * stride + i], bvy = S[F_BVY * stride + i];
  float dx = bx - px, dy = by - py;
  float dist = hypot2(dx, dy), ang = atan2_deg(dy, dx);
  if (w.ball_dist_from_self) w.ball_dist_from_self[i] = dist;
  if (w.ball_angle_from_self) w.ball_angle_from_self[i] = ang;
  if (w.ball_relative_x) w.ball_relative_x[i] = dx;
  if (w.ball_relative_y) w.ball_relative_y[i] = dy;
  if (w.ball_pos_dist) w.ball_pos_dist[i] = hypot2(bx, by);
  if (w.ball_pos_angle) w.ball_pos_angle[i] = atan2_deg(by, bx);
  if (w.ball_vel_dist) w.ball_vel_dist[i] = hypot2(bvx, bvy);
  if (w.ball_vel_angle) w.ball_vel_angle[i] = atan2_deg(bvy, bvx);
  if (w.self_pos_dist) w.self_pos_dist[i] = hypot2(px, py);
  if (w.self_pos_angle) w.self_pos_angle[i] = atan2_deg(py, px);
  if (w.self_vel_dist) w.self_vel_dist[i] = hypot2(vx, vy);
  if (w.self_vel_angle) w.self_vel_angle[i] = atan2_deg(vy, vx);
  if (w.self_dist_from_ball) w.self_dist_from_ball[i] = dist;
  if (w.self_angle_from_ball) w.self_angle_from_ball[i] = atan2_deg(-dy, -dx);
}

// diagnostic ("validate state" guard): count the envs whose state words left their domain
enum { SV_NONFINITE, SV_BODY, SV_STAMINA, SV_EFFORT_RECOVERY, SV_COUNTERS, SV_OBS, SV_WORDS = 8 };
__global__ __launch_bounds__(kBlock) void s2d_validate_kernel(S2DHot p, const float* __restrict__ S, int64_t stride, int64_t n,
                                                              const float* __restrict__ obs, unsigned int* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  Env e;
  env_load(e, S, stride, i);
  const float w[15] = {e.px, e.py, e.vx, e.vy, e.body, e.stamina, e.effort, e.recovery, e.capacity, e.bx, e.by, e.bvx, e.bvy,
                       e.prev_dist, e.prev_angle};
  bool finite = true;
#pragma unroll
  for (int k = 0; k < 15; ++k) finite = finite && isfinite(w[k]);
  bool obs_ok = true;
#pragma unroll
  for (int k = 0; k < S2D_OBS_DIM; ++k) obs_ok = obs_ok && isfinite(obs[i * S2D_OBS_DIM + k]);
  if (!finite) atomicAdd(&out[SV_NONFINITE], 1u);
  if (!(fabsf(e.body) <= 180.0f) || !(fabsf(e.prev_angle) <= 180.0f)) atomicAdd(&out[SV_BODY], 1u);
  if (!(e.stamina >= 0.0f && e.stamina <= p.stamina_max) || !(e.capacity >= 0.0f || p.stamina_capacity < 0.0f)) atomicAdd(&out[SV_STAMINA], 1u);
  if (!(e.effort >= p.effort_min && e.effort <= p.effort_init) || !(e.recovery >= p.recover_min && e.recovery <= p.recover_init))
    atomicAdd(&out[SV_EFFORT_RECOVERY], 1u);
  if (e.step_number < 0 || e.episode < 0 || !(e.prev_dist >= 0.0f)) atomicAdd(&out[SV_COUNTERS], 1u);
  if (!obs_ok) atomicAdd(&out[SV_OBS], 1u);
}

// diagnostic: evaluate the math spec / Philox on the device (tests compare with the oracle)
__global__ void s2d_debug_eval_kernel(int op, const float* __restrict__ in, float* __restrict__ out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (op == 9) {                                         // reset_sample_coop against reset_sample (whole waves: n is a multiple of 64)
    __shared__ uint32_t scratch[256];
    const uint32_t* u = reinterpret_cast<const uint32_t*>(in) + 4 * i;   // key, need, seed, travel factor (float bits)
    S2DHot p{}; p.seed_lo = u[2]; p.seed_hi = 0x5EEDu; p.half_l = 52.5f; p.half_w = 34.0f;
    S2DRare r{}; r.change_ball_position = 1; r.change_ball_velocity = 1; r.travel_factor = __uint_as_float(u[3]);
    const int lane = threadIdx.x & 63;
    const ResetSample a = reset_sample_coop(p, r, (uint32_t)i, 7u, u[0], u[1] != 0u, lane, scratch + (threadIdx.x & ~63u));
    const ResetSample b = reset_sample(p, r, (uint32_t)i, 7u, u[0]);
    float* q = out + 14 * i;
    q[0] = a.px; q[1] = a.py; q[2] = a.body; q[3] = a.bx; q[4] = a.by; q[5] = a.bvx; q[6] = a.bvy;
    q[7] = b.px; q[8] = b.py; q[9] = b.body; q[10] = b.bx; q[11] = b.by; q[12] = b.bvx; q[13] = b.bvy;
    return;
  }
  if (i >= n) return;
  switch (op) {
    case 10: {                                           // movement noise of one commanded cycle (s2d_device.h: noise_prepare)
      const uint32_t* u = reinterpret_cast<const uint32_t*>(in) + 4 * i;   // gid_lo, gid_hi, counter k, seed_lo (seed_hi = 0)
      S2DHot p{}; p.seed_lo = u[3]; p.seed_hi = 0u;
      const NoiseIn nz = noise_prepare(p, u[0], u[1], u[2], S2D_ST_NOISE, false);
      float* q = out + 6 * i;
      q[0] = nz.pm; q[1] = nz.ps; q[2] = nz.pc; q[3] = nz.bm; q[4] = nz.bs; q[5] = nz.bc;
      break;
    }
    case 0: { float s, c; sincos_deg(in[i], s, c); out[2 * i] = s; out[2 * i + 1] = c; break; }
    case 1: out[i] = atan2_deg(in[2 * i], in[2 * i + 1]); break;
    case 2: out[i] = exp_spec(in[i]); break;
    case 3: out[i] = norm_deg_any(in[i]); break;
    case 4: {
      const uint32_t* u = reinterpret_cast<const uint32_t*>(in) + 6 * i;
      U4 r = philox4x32_10(u[0], u[1], u[2], u[3], u[4], u[5]);
      uint32_t* q = reinterpret_cast<uint32_t*>(out) + 4 * i;
      q[0] = r.x; q[1] = r.y; q[2] = r.z; q[3] = r.w;
      break;
    }
    case 5: out[i] = hypot2(in[2 * i], in[2 * i + 1]); break;
    case 6: {                                            // A3 state_to_observation
      const float* r = in + 9 * i;
      S2DHot p{}; p.inv_half_l = r[7]; p.inv_half_w = r[8];
      ObsOut ob;
      observe(p, r[4], r[5], r[6], r[0], r[1], r[2], r[3], ob);
      for (int k = 0; k < S2D_OBS_DIM; ++k) out[S2D_OBS_DIM * i + k] = ob.o[k];
      break;
    }
    case 7: {                                            // A2 action_to_rpc_actions
      const float* r = in + 8 * i;
      S2DHot p{}; p.act_scale = r[6];
      const Action4 a{r[0], r[1], r[2], r[3]};
      int cmd = 0; float power = 0.0f, dir = 0.0f;
      const int mode = (int)r[5];
      if (mode == S2D_MODE_DISCRETE) action_map<S2D_MODE_DISCRETE>(p, a, r[4], cmd, power, dir);
      else if (mode == S2D_MODE_CONT1) action_map<S2D_MODE_CONT1>(p, a, r[4], cmd, power, dir);
      else action_map<S2D_MODE_TURN4>(p, a, r[4], cmd, power, dir);
      out[3 * i] = (float)cmd; out[3 * i + 1] = power; out[3 * i + 2] = dir;
      break;
    }
    case 8: {                                            // A4 check_trainer_observation
      const float* r = in + 12 * i;
      S2DHot p{}; p.min_distance_to_ball = r[8]; p.max_steps = (int)r[9]; p.half_l = r[10]; p.half_w = r[11];
      Env e{}; e.bx = r[0]; e.by = r[1]; e.px = r[2]; e.py = r[3]; e.body = r[4]; e.step_number = (int)r[5];
      e.prev_dist = r[6]; e.prev_angle = r[7];
      ObsOut ob; int done, result; float reward;
      observe_and_check(p, e, sq2(e.bx - e.px, e.by - e.py), ob, done, reward, result);
      float* q = out + 5 * i;
      q[0] = (float)done; q[1] = reward; q[2] = (float)result; q[3] = e.prev_dist; q[4] = e.prev_angle;
      break;
    }
    default: break;
  }
}

// ------------------------------------------------------------------------------------------
// host side: engine object + C ABI
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
// shared with s2d_match.hip (same library, hidden symbol)
extern "C" void s2d_internal_set_error(const char* msg) { g_err = msg ? msg : ""; }
#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return fail(S2D_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));               \
  } while (0)

// s2d_rollout2.hip (same library, hidden symbol): launches the two-envs-per-lane pipeline if the batch and the record allow it
extern "C" int s2d_internal_rollout2(int mode, int noise, const S2DHot* hot, const S2DRare* rare_dev, float* S, int64_t stride,
                                     int64_t n, int n_steps, const void* actions_dev, int kind, const RolloutOut* ro,
                                     const StepOut* o, void* stream, char* name);

struct S2DEngine {
  S2DConfig cfg;
  S2DHot hot;
  S2DRare rare;
  const S2DRare* rare_dev;
  int mode;      // S2D_MODE_*
  bool noise;
  int rollout_ws;  // -1 auto (by batch size), 0 unified kernel, 1 wave-specialised kernel
  int rollout_e;   // envs per lane of the wave-specialised rollout: 2 (s2d_rollout2.hip, where the batch and the record allow it) or 1
  int rollout_nt;  // -1 by record size, 0 / 1: plain / non-temporal record stores (experiments)
  char kernel_name[96];   // full instantiation of the last rollout launch
  int64_t n, stride;
  int device;
  char* arena;
  size_t arena_bytes;
  bool owns_arena;
  S2DBuffers buf;
  StepOut out;
  const char* last_kernel;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static int64_t stride_for(int64_t n) { return (int64_t)align_up((size_t)n, 256); }

static_assert(sizeof(S2DRare) <= 256, "S2DTables sit 256 bytes behind S2DRare");
struct ArenaLayout {
  size_t state, obs, reward, done, result, terminal_obs, action_dir, action_cmd, stats, prep, rare, tables, total;
};
static ArenaLayout layout_for(int64_t n) {
  ArenaLayout L;
  size_t s = (size_t)stride_for(n), off = 0;
  L.state = off; off += align_up((size_t)F_COUNT * s * 4, 256);
  L.obs = off; off += align_up(s * S2D_OBS_DIM * 4, 256);
  L.reward = off; off += align_up(s * 4, 256);
  L.done = off; off += align_up(s, 256);
  L.result = off; off += align_up(s, 256);
  L.terminal_obs = off; off += align_up(s * S2D_OBS_DIM * 4, 256);
  L.action_dir = off; off += align_up(s * 4, 256);
  L.action_cmd = off; off += align_up(s, 256);
  L.stats = off; off += align_up((size_t)S2D_STATS_ROWS(n) * 8 * sizeof(unsigned long long), 256);
  L.prep = off; off += align_up((size_t)(2 * PS_WORDS + 2) * s * 4, 256);
  L.rare = off; off += align_up(sizeof(S2DRare), 256);
  L.tables = off; off += align_up(sizeof(S2DTables), 256);   // directly behind S2DRare: kernels find them at rp + 256 bytes
  L.total = off;
  return L;
}

#define S2D_STR2(x) #x
#define S2D_STR(x) S2D_STR2(x)
S2D_API const char* s2d_version(void) { return "s2d-hip 0.4 (gfx950, abi " S2D_STR(S2D_ABI_VERSION) ")"; }
S2D_API const char* s2d_last_error(void) { return g_err.c_str(); }

S2D_API void s2d_default_config(S2DConfig* c) {
  if (!c) return;
  std::memset(c, 0, sizeof *c);
  c->abi_version = S2D_ABI_VERSION;
  c->struct_bytes = (uint32_t)sizeof(S2DConfig);
  S2DServerParams& s = c->sp;  // rcssserver stock values (EXT)
  s.pitch_half_length = 52.5; s.pitch_half_width = 34.0;
  s.player_size = 0.3; s.player_decay = 0.4; s.player_rand = 0.1; s.player_speed_max = 1.05;
  s.player_accel_max = 1.0; s.inertia_moment = 5.0;
  s.stamina_max = 8000.0; s.stamina_inc_max = 45.0; s.stamina_capacity = 130600.0; s.extra_stamina = 50.0;
  s.recover_init = 1.0; s.recover_dec_thr = 0.3; s.recover_min = 0.5; s.recover_dec = 0.002;
  s.effort_init = 1.0; s.effort_dec_thr = 0.3; s.effort_min = 0.6; s.effort_dec = 0.005;
  s.effort_inc_thr = 0.6; s.effort_inc = 0.01;
  s.dash_power_rate = 0.006; s.max_dash_power = 100.0; s.min_dash_power = 0.0;
  s.max_dash_angle = 180.0; s.min_dash_angle = -180.0; s.dash_angle_step = 1.0;
  s.side_dash_rate = 0.4; s.back_dash_rate = 0.6;
  s.max_moment = 180.0; s.min_moment = -180.0;
  s.ball_size = 0.085; s.ball_decay = 0.94; s.ball_rand = 0.05; s.ball_speed_max = 3.0; s.ball_accel_max = 2.7;
  s.collision_vel_rate = -0.1;
  S2DReachBallParams& t = c->task;  // reach_ball_env.py:26-36
  t.change_ball_position = 1; t.change_ball_velocity = 0;
  t.ball_position_x = 0; t.ball_position_y = 0; t.ball_speed = 0; t.ball_direction = 0;
  t.min_distance_to_ball = 5.0; t.max_steps = 200;
  t.use_continuous_action = 1; t.action_space_size = 16; t.use_turning = 0;
  t.reset_ball_decay = 0.96;   // reach_ball_env.py:207
  c->seed = 0x5EEDull; c->env_id_offset = 0; c->auto_reset = 1;
  // The reference starts rcssserver with synch_mode / auto_mode / fullstate_l / coach only (soccer_2d_env.py:363-368), so the
  // stock player_rand = 0.1 / ball_rand = 0.05 stay ON: noisy dynamics are the drop-in default.  noise = 0 is the explicit
  // opt-in for deterministic runs (parity tests, the headline bench of SURVEY 8d).
  c->noise = 1;
}

S2D_API int s2d_validate_config(const S2DConfig* c) {
  if (!c) return fail(S2D_EINVAL, "config is NULL");
  if (c->abi_version != S2D_ABI_VERSION) return fail(S2D_EINVAL, "config.abi_version mismatch");
  if (c->struct_bytes != sizeof(S2DConfig)) return fail(S2D_EINVAL, "config.struct_bytes != sizeof(S2DConfig)");
  const S2DServerParams& s = c->sp;
  const S2DReachBallParams& t = c->task;
  if (!(s.pitch_half_length > 0) || !(s.pitch_half_width > 0)) return fail(S2D_EINVAL, "pitch extents must be > 0");
  if (!(s.player_decay >= 0 && s.player_decay <= 1) || !(s.ball_decay >= 0 && s.ball_decay <= 1))
    return fail(S2D_EINVAL, "decay must be in [0,1]");
  if (!(s.stamina_max > 0)) return fail(S2D_EINVAL, "stamina_max must be > 0");
  if (!(s.dash_angle_step >= 0)) return fail(S2D_EINVAL, "dash_angle_step must be >= 0");
  if (!t.use_continuous_action && (t.action_space_size < 1 || t.action_space_size > (1 << 20)))
    return fail(S2D_EINVAL, "action_space_size must be in [1, 2^20]");
  if (t.max_steps < 0) return fail(S2D_EINVAL, "max_steps must be >= 0");
  if (!(t.reset_ball_decay > 0 && t.reset_ball_decay < 1)) return fail(S2D_EINVAL, "reset_ball_decay must be in (0,1)");
  if (c->env_id_offset < 0) return fail(S2D_EINVAL, "env_id_offset must be >= 0");
  return S2D_OK;
}

static void dev_params_from_config(const S2DConfig& c, S2DHot& h, S2DRare& r) {
  const S2DServerParams& s = c.sp;
  const S2DReachBallParams& t = c.task;
  std::memset(&h, 0, sizeof h);
  std::memset(&r, 0, sizeof r);
  const float accel_max = (float)s.player_accel_max, pspeed_max = (float)s.player_speed_max;
  const float bspeed_max = (float)s.ball_speed_max;
  const float rsum = (float)s.player_size + (float)s.ball_size;
  h.inv_half_l = (float)(1.0 / s.pitch_half_length); h.inv_half_w = (float)(1.0 / s.pitch_half_width);
  h.half_l = (float)s.pitch_half_length; h.half_w = (float)s.pitch_half_width;
  h.player_decay = (float)s.player_decay; h.ball_decay = (float)s.ball_decay;
  h.player_accel_max2 = accel_max * accel_max; h.player_speed_max2 = pspeed_max * pspeed_max;
  h.ball_speed_max2 = bspeed_max * bspeed_max; h.rsum2 = rsum * rsum;
  h.stamina_max = (float)s.stamina_max; h.stamina_inc_max = (float)s.stamina_inc_max;
  h.extra_stamina = (float)s.extra_stamina; h.stamina_capacity = (float)s.stamina_capacity;
  h.recover_init = (float)s.recover_init;
  h.recover_dec_thr_value = (float)(s.recover_dec_thr * s.stamina_max);
  h.recover_min = (float)s.recover_min; h.recover_dec = (float)s.recover_dec;
  h.effort_init = (float)s.effort_init;
  h.effort_dec_thr_value = (float)(s.effort_dec_thr * s.stamina_max);
  h.effort_min = (float)s.effort_min; h.effort_dec = (float)s.effort_dec;
  h.effort_inc_thr_value = (float)(s.effort_inc_thr * s.stamina_max);
  h.effort_inc = (float)s.effort_inc;
  h.dash_power_rate = (float)s.dash_power_rate; h.max_dash_power = (float)s.max_dash_power;
  h.min_dash_power = (float)s.min_dash_power; h.max_dash_angle = (float)s.max_dash_angle;
  h.min_dash_angle = (float)s.min_dash_angle; h.dash_angle_step = (float)s.dash_angle_step;
  h.inv_dash_angle_step = s.dash_angle_step > 0 ? (float)(1.0 / s.dash_angle_step) : 0.0f;
  h.side_dash_rate = (float)s.side_dash_rate; h.back_dash_rate = (float)s.back_dash_rate;
  h.min_distance_to_ball = (float)t.min_distance_to_ball;
  {  // smallest float T with sqrtf(T) >= min_distance (correctly rounded sqrt is monotone)
    const float m = h.min_distance_to_ball;
    float T = m > 0.0f ? m * m : 0.0f;
    if (m > 0.0f) {
      while (T > 0.0f && std::sqrt(std::nextafter(T, 0.0f)) >= m) T = std::nextafter(T, 0.0f);
      while (std::sqrt(T) < m) T = std::nextafter(T, INFINITY);
    }
    h.min_dist2_thr = T;
  }
  h.act_scale = (float)(360.0 / (double)(t.action_space_size > 0 ? t.action_space_size : 1));
  h.max_steps = t.max_steps; h.n_actions = t.action_space_size; h.auto_reset = c.auto_reset;
  h.seed_lo = (uint32_t)c.seed; h.seed_hi = (uint32_t)(c.seed >> 32);
  h.gid_lo = (uint32_t)(uint64_t)c.env_id_offset; h.gid_hi = (uint32_t)((uint64_t)c.env_id_offset >> 32);
  h.max_moment = (float)s.max_moment; h.min_moment = (float)s.min_moment;
  h.inertia_moment = (float)s.inertia_moment; h.player_rand = (float)s.player_rand; h.ball_rand = (float)s.ball_rand;
  r.player_accel_max = accel_max; r.player_speed_max = pspeed_max; r.ball_speed_max = bspeed_max;
  r.rsum = rsum; r.collision_vel_rate = (float)s.collision_vel_rate;
  r.recover_init = (float)s.recover_init;
  r.ball_position_x = (float)t.ball_position_x; r.ball_position_y = (float)t.ball_position_y;
  r.ball_speed = (float)t.ball_speed; r.ball_direction = (float)t.ball_direction;
  r.travel_factor = (float)((1.0 - std::pow(t.reset_ball_decay, (double)t.max_steps)) / (1.0 - t.reset_ball_decay));
  r.change_ball_position = t.change_ball_position; r.change_ball_velocity = t.change_ball_velocity;
  // dash-only fast path (s2d_device.h, S2DTables): every command of the discrete / 1-D continuous modes is Dash(100, dir)
  {
    const float power = 100.0f < h.min_dash_power ? h.min_dash_power : (100.0f > h.max_dash_power ? h.max_dash_power : 100.0f);   // clampf
    const bool whole_step = h.dash_angle_step >= 1.0f && h.dash_angle_step == std::rint(h.dash_angle_step);
    const bool dash_only = !(t.use_continuous_action && t.use_turning);
    r.tab_power = power;
    r.tab_len = (dash_only && whole_step && power > 0.0f && c.auto_reset && t.max_steps + 1 <= S2D_TAB_MAX) ? t.max_steps + 1 : 0;
  }
}

S2D_API size_t s2d_arena_bytes(const S2DConfig* cfg, int64_t n_envs) {
  if (!cfg || n_envs <= 0) return 0;
  return layout_for(n_envs).total;
}

struct DeviceGuard {
  int prev = -1; bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess) { ok = (prev == dev) || (hipSetDevice(dev) == hipSuccess); }
  }
  ~DeviceGuard() { if (ok && prev >= 0) (void)hipSetDevice(prev); }
};

static int grid_for(int64_t n) { return (int)((n + kBlock - 1) / kBlock); }

S2D_API int s2d_create(const S2DConfig* cfg, int64_t n_envs, int device, void* arena_dev, size_t arena_bytes,
                       void* stream, S2DHandle* out) {
  if (!out) return fail(S2D_EINVAL, "out handle is NULL");
  *out = nullptr;
  int rc = s2d_validate_config(cfg);
  if (rc != S2D_OK) return rc;
  if (n_envs <= 0 || n_envs > (int64_t)1 << 31) return fail(S2D_EINVAL, "n_envs must be in [1, 2^31]");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(S2D_ENODEV, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(S2D_EINVAL, "device index out of range");
  DeviceGuard guard(device);
  if (!guard.ok) return fail(S2D_EHIP, "hipSetDevice failed");
  ArenaLayout L = layout_for(n_envs);
  S2DEngine* h = new (std::nothrow) S2DEngine();
  if (!h) return fail(S2D_ENOMEM, "host allocation failed");
  h->cfg = *cfg; h->n = n_envs; h->stride = stride_for(n_envs); h->device = device;
  h->last_kernel = "";
  dev_params_from_config(*cfg, h->hot, h->rare);
  h->mode = !cfg->task.use_continuous_action ? S2D_MODE_DISCRETE
                                             : (cfg->task.use_turning ? S2D_MODE_TURN4 : S2D_MODE_CONT1);
  h->noise = cfg->noise != 0;
  h->rollout_ws = -1;
  if (const char* v = std::getenv("S2D_ROLLOUT_WS")) h->rollout_ws = std::atoi(v) != 0 ? 1 : 0;
  // envs per lane of the wave-specialised rollout: 1 (default).  2 = s2d_rollout2.hip: bit-identical, whole-line record stores,
  // but 2 waves per SIMD at 65 536 envs leave the waves issue-bound: opt-in
  h->rollout_e = 1; h->rollout_nt = -1; h->kernel_name[0] = 0;
  if (const char* v = std::getenv("S2D_ROLLOUT_E")) h->rollout_e = std::atoi(v) == 2 ? 2 : 1;
  if (const char* v = std::getenv("S2D_ROLLOUT_NT")) h->rollout_nt = std::atoi(v) != 0 ? 1 : 0;
  if (arena_dev) {
    if (arena_bytes < L.total) { delete h; return fail(S2D_ENOMEM, "arena smaller than s2d_arena_bytes()"); }
    if (reinterpret_cast<uintptr_t>(arena_dev) & 255u) { delete h; return fail(S2D_EINVAL, "arena must be 256-byte aligned"); }
    h->arena = static_cast<char*>(arena_dev); h->owns_arena = false;
  } else {
    void* p = nullptr;
    if (hipMalloc(&p, L.total) != hipSuccess) { delete h; return fail(S2D_ENOMEM, "hipMalloc of the arena failed"); }
    h->arena = static_cast<char*>(p); h->owns_arena = true;
  }
  h->arena_bytes = L.total;
  S2DBuffers& b = h->buf;
  b.n_envs = n_envs;
  float* S = reinterpret_cast<float*>(h->arena + L.state);
  float** planes[15] = {&b.player_x, &b.player_y, &b.player_vx, &b.player_vy, &b.player_body, &b.stamina,
                        &b.effort, &b.recovery, &b.stamina_capacity, &b.ball_x, &b.ball_y, &b.ball_vx,
                        &b.ball_vy, &b.prev_dist, &b.prev_angle};
  for (int f = 0; f < 15; ++f) *planes[f] = S + (size_t)f * h->stride;
  b.step_number = reinterpret_cast<int32_t*>(S + (size_t)F_STEP * h->stride);
  b.cycle = reinterpret_cast<int32_t*>(S + (size_t)F_CYCLE * h->stride);
  b.policy_step = reinterpret_cast<int32_t*>(S + (size_t)F_POLICY * h->stride);
  b.episode = reinterpret_cast<int32_t*>(S + (size_t)F_EPISODE * h->stride);
  b.obs = reinterpret_cast<float*>(h->arena + L.obs);
  b.reward = reinterpret_cast<float*>(h->arena + L.reward);
  b.done = reinterpret_cast<uint8_t*>(h->arena + L.done);
  b.result = reinterpret_cast<uint8_t*>(h->arena + L.result);
  b.terminal_obs = reinterpret_cast<float*>(h->arena + L.terminal_obs);
  b.action_dir = reinterpret_cast<float*>(h->arena + L.action_dir);
  b.action_cmd = reinterpret_cast<uint8_t*>(h->arena + L.action_cmd);
  b.stats = reinterpret_cast<unsigned long long*>(h->arena + L.stats);
  h->out = StepOut{b.obs, b.reward, b.done, b.result, b.terminal_obs, b.action_dir, b.action_cmd, b.stats,
                   reinterpret_cast<float*>(h->arena + L.prep)};
  h->rare_dev = reinterpret_cast<const S2DRare*>(h->arena + L.rare);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(h->arena, 0, L.total, st);
  // h->rare lives as long as the handle, so the (possibly staged) copy may complete later
  if (e == hipSuccess) e = hipMemcpyAsync(h->arena + L.rare, &h->rare, sizeof(S2DRare), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(s2d_init_kernel, dim3(grid_for(n_envs)), dim3(kBlock), 0, st, h->hot, h->rare_dev, S,
                       h->stride, h->n);
    hipLaunchKernelGGL(s2d_tables_kernel, dim3(1), dim3(64), 0, st, h->hot, h->rare_dev,
                       reinterpret_cast<S2DTables*>(h->arena + L.tables));
    e = hipGetLastError();
  }
  if (e != hipSuccess) {
    std::string m = std::string("arena initialisation: ") + hipGetErrorString(e);
    if (h->owns_arena) (void)hipFree(h->arena);
    delete h;
    return fail(S2D_EHIP, m);
  }
  *out = h;
  return S2D_OK;
}

S2D_API void s2d_destroy(S2DHandle h) {
  if (!h) return;
  if (h->owns_arena && h->arena) {
    DeviceGuard guard(h->device);
    (void)hipFree(h->arena);
  }
  delete h;
}

S2D_API int s2d_buffers(S2DHandle h, S2DBuffers* out) {
  if (!h || !out) return fail(S2D_EINVAL, "NULL argument");
  *out = h->buf;
  return S2D_OK;
}

S2D_API int s2d_buffer_offsets(S2DHandle h, int64_t* offsets, int n_offsets) {
  if (!h || !offsets) return fail(S2D_EINVAL, "NULL argument");
  const void* ptrs[] = {h->buf.player_x, h->buf.player_y, h->buf.player_vx, h->buf.player_vy, h->buf.player_body,
                        h->buf.stamina, h->buf.effort, h->buf.recovery, h->buf.stamina_capacity, h->buf.ball_x,
                        h->buf.ball_y, h->buf.ball_vx, h->buf.ball_vy, h->buf.prev_dist, h->buf.prev_angle,
                        h->buf.step_number, h->buf.cycle, h->buf.policy_step, h->buf.episode, h->buf.obs, h->buf.reward, h->buf.done, h->buf.result,
                        h->buf.terminal_obs, h->buf.action_dir, h->buf.action_cmd, h->buf.stats};
  const int count = 1 + (int)(sizeof ptrs / sizeof ptrs[0]);
  if (n_offsets < count) return fail(S2D_EINVAL, "offsets array too small (need 28)");
  offsets[0] = (int64_t)h->arena_bytes;
  for (int k = 1; k < count; ++k) offsets[k] = (int64_t)(static_cast<const char*>(ptrs[k - 1]) - h->arena);
  return S2D_OK;
}

static int check_action_kind(const S2DEngine* h, const void* actions, int kind) {
  const S2DReachBallParams& t = h->cfg.task;
  switch (kind) {
    case S2D_ACT_DISCRETE_I32:
    case S2D_ACT_DISCRETE_I64:
      if (t.use_continuous_action) return fail(S2D_EINVAL, "discrete actions given to a continuous-action env");
      break;
    case S2D_ACT_CONTINUOUS:
      if (!t.use_continuous_action || t.use_turning) return fail(S2D_EINVAL, "float[N][1] actions need use_continuous_action && !use_turning");
      break;
    case S2D_ACT_TURNING:
      if (!t.use_continuous_action || !t.use_turning) return fail(S2D_EINVAL, "float[N][4] actions need use_continuous_action && use_turning");
      if (reinterpret_cast<uintptr_t>(actions) & 15u) return fail(S2D_EINVAL, "float[N][4] actions must be 16-byte aligned");
      break;
    case S2D_ACT_RANDOM:
      return S2D_OK;
    case S2D_ACT_COMMAND:
      if (reinterpret_cast<uintptr_t>(actions) & 15u) return fail(S2D_EINVAL, "float[N][4] commands must be 16-byte aligned");
      break;
    default:
      return fail(S2D_EINVAL, "unknown action_kind");
  }
  if (!actions) return fail(S2D_EINVAL, "actions pointer is NULL");
  return S2D_OK;
}

S2D_API int s2d_reset(S2DHandle h, const uint8_t* mask_dev, void* stream) {
  if (!h) return fail(S2D_EINVAL, "NULL handle");
  DeviceGuard guard(h->device);
  auto k = h->noise ? s2d_reach_reset_kernel<true> : s2d_reach_reset_kernel<false>;
  hipLaunchKernelGGL(k, dim3(grid_for(h->n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream), h->hot,
                     h->rare_dev, reinterpret_cast<float*>(h->buf.player_x), h->stride, h->n, mask_dev, h->out);
  HIP_TRY(hipGetLastError());
  h->last_kernel = "s2d_reach_reset_kernel";
  return S2D_OK;
}

S2D_API int s2d_step(S2DHandle h, const void* actions_dev, int action_kind, void* stream) {
  if (!h) return fail(S2D_EINVAL, "NULL handle");
  int rc = check_action_kind(h, actions_dev, action_kind);
  if (rc != S2D_OK) return rc;
  DeviceGuard guard(h->device);
  using StepK = void (*)(S2DHot, const S2DRare*, float*, int64_t, int64_t, const void*, int, StepOut, int);
  static const StepK table[3][2] = {
      {s2d_reach_step_kernel<S2D_MODE_DISCRETE, false>, s2d_reach_step_kernel<S2D_MODE_DISCRETE, true>},
      {s2d_reach_step_kernel<S2D_MODE_CONT1, false>, s2d_reach_step_kernel<S2D_MODE_CONT1, true>},
      {s2d_reach_step_kernel<S2D_MODE_TURN4, false>, s2d_reach_step_kernel<S2D_MODE_TURN4, true>}};
  // main workgroups + (with auto-reset) as many refill workgroups: they keep the prepared episodes of StepOut::prep topped up
  const int main_blocks = grid_for(h->n), refill_blocks = h->cfg.auto_reset ? main_blocks : 0;
  hipLaunchKernelGGL(table[h->mode][h->noise ? 1 : 0], dim3(main_blocks + refill_blocks), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), h->hot, h->rare_dev,
                     reinterpret_cast<float*>(h->buf.player_x), h->stride, h->n, actions_dev, action_kind, h->out,
                     refill_blocks);
  HIP_TRY(hipGetLastError());
  h->last_kernel = "s2d_reach_step_kernel";
  return S2D_OK;
}

S2D_API int s2d_step_k(S2DHandle h, int k, const void* actions_dev, int action_kind, const S2DRollout* out, void* stream) {
  if (!h) return fail(S2D_EINVAL, "NULL handle");
  if (k < 1 || k > 64) return fail(S2D_EINVAL, "s2d_step_k: k must be in [1, 64] (longer launches: s2d_rollout)");
  if (action_kind == S2D_ACT_COMMAND) return fail(S2D_EINVAL, "S2D_ACT_COMMAND is a per-step action kind (s2d_step)");
  int rc = check_action_kind(h, actions_dev, action_kind);
  if (rc != S2D_OK) return rc;
  RolloutOut ro{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  if (out) {
    ro = RolloutOut{out->obs, out->action, out->reward, out->done, out->result, 0};
    if (h->mode == S2D_MODE_TURN4 && (reinterpret_cast<uintptr_t>(out->action) & 15u))
      return fail(S2D_EINVAL, "record action buffer float[K][N][4] must be 16-byte aligned");
    if (reinterpret_cast<uintptr_t>(out->obs) & 3u) return fail(S2D_EINVAL, "record obs buffer must be 4-byte aligned");
  }
  DeviceGuard guard(h->device);
  using StepK = void (*)(S2DHot, const S2DRare*, float*, int64_t, int64_t, int, const void*, int, RolloutOut, StepOut, int);
  static const StepK table[3][2] = {
      {s2d_reach_step_k_kernel<S2D_MODE_DISCRETE, false>, s2d_reach_step_k_kernel<S2D_MODE_DISCRETE, true>},
      {s2d_reach_step_k_kernel<S2D_MODE_CONT1, false>, s2d_reach_step_k_kernel<S2D_MODE_CONT1, true>},
      {s2d_reach_step_k_kernel<S2D_MODE_TURN4, false>, s2d_reach_step_k_kernel<S2D_MODE_TURN4, true>}};
  const int main_blocks = grid_for(h->n), refill_blocks = h->cfg.auto_reset ? main_blocks : 0;
  hipLaunchKernelGGL(table[h->mode][h->noise ? 1 : 0], dim3(main_blocks + refill_blocks), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), h->hot, h->rare_dev, reinterpret_cast<float*>(h->buf.player_x), h->stride, h->n,
                     k, actions_dev, action_kind, ro, h->out, refill_blocks);
  HIP_TRY(hipGetLastError());
  h->last_kernel = "s2d_reach_step_k_kernel";
  return S2D_OK;
}

S2D_API int s2d_rollout(S2DHandle h, int n_steps, const void* actions_dev, int action_kind, const S2DRollout* out,
                        void* stream) {
  if (!h) return fail(S2D_EINVAL, "NULL handle");
  if (n_steps < 0) return fail(S2D_EINVAL, "n_steps must be >= 0");
  if (action_kind == S2D_ACT_COMMAND) return fail(S2D_EINVAL, "S2D_ACT_COMMAND is a per-step action kind (s2d_step)");
  int rc = check_action_kind(h, actions_dev, action_kind);
  if (rc != S2D_OK) return rc;
  if (n_steps == 0) return S2D_OK;
  RolloutOut ro{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
  if (out) {
    ro = RolloutOut{out->obs, out->action, out->reward, out->done, out->result, 0};
    const int64_t per_step = (out->obs ? 4 * S2D_OBS_DIM : 0) + (out->action ? (h->mode == S2D_MODE_TURN4 ? 16 : 4) : 0) + (out->reward ? 4 : 0) + (out->done ? 1 : 0) +
                             (out->result ? 1 : 0);
    ro.nt = h->rollout_nt >= 0 ? h->rollout_nt : ((int64_t)n_steps * h->n * per_step > kInfinityCacheBytes);
    if (h->cfg.task.use_continuous_action && h->cfg.task.use_turning && (reinterpret_cast<uintptr_t>(out->action) & 15u))
      return fail(S2D_EINVAL, "rollout action buffer float[T][N][4] must be 16-byte aligned");
    if (reinterpret_cast<uintptr_t>(out->obs) & 3u) return fail(S2D_EINVAL, "rollout obs buffer must be 4-byte aligned");
  }
  DeviceGuard guard(h->device);
  using RollK = void (*)(S2DHot, const S2DRare*, float*, int64_t, int64_t, int, const void*, int, RolloutOut, StepOut);
  static const RollK table[3][2] = {
      {s2d_reach_rollout_kernel<S2D_MODE_DISCRETE, false>, s2d_reach_rollout_kernel<S2D_MODE_DISCRETE, true>},
      {s2d_reach_rollout_kernel<S2D_MODE_CONT1, false>, s2d_reach_rollout_kernel<S2D_MODE_CONT1, true>},
      {s2d_reach_rollout_kernel<S2D_MODE_TURN4, false>, s2d_reach_rollout_kernel<S2D_MODE_TURN4, true>}};
  static const RollK table_ws[3][2] = {
      {s2d_reach_rollout_ws_kernel<S2D_MODE_DISCRETE, false>, s2d_reach_rollout_ws_kernel<S2D_MODE_DISCRETE, true>},
      {s2d_reach_rollout_ws_kernel<S2D_MODE_CONT1, false>, s2d_reach_rollout_ws_kernel<S2D_MODE_CONT1, true>},
      {s2d_reach_rollout_ws_kernel<S2D_MODE_TURN4, false>, s2d_reach_rollout_ws_kernel<S2D_MODE_TURN4, true>}};
  // small batches: four waves per env group (policy | simulate | agent | ball)
  const bool ws = h->rollout_ws < 0 ? (h->n <= kWsMaxEnvs) : (h->rollout_ws != 0);
  static const char* const mode_names[3] = {"discrete", "continuous", "turning"};
  if (ws) {
    // two envs per lane (s2d_rollout2.hip) where the batch is a multiple of 128 envs and the record is complete and aligned
    if (h->rollout_e == 2 && s2d_internal_rollout2(h->mode, h->noise ? 1 : 0, &h->hot, h->rare_dev, reinterpret_cast<float*>(h->buf.player_x),
                                                   h->stride, h->n, n_steps, actions_dev, action_kind, &ro, &h->out, stream, h->kernel_name)) {
      HIP_TRY(hipGetLastError());
      h->last_kernel = h->kernel_name;
      return S2D_OK;
    }
    RollK kern_ws = table_ws[h->mode][h->noise ? 1 : 0];
    int rec = 0;
    if (h->mode == S2D_MODE_DISCRETE && !h->noise && ro.obs && ro.action && ro.reward && ro.done && ro.result) {
      kern_ws = ro.nt ? s2d_reach_rollout_ws_kernel<S2D_MODE_DISCRETE, false, 2> : s2d_reach_rollout_ws_kernel<S2D_MODE_DISCRETE, false, 1>;
      rec = ro.nt ? 2 : 1;
    }
    hipLaunchKernelGGL(kern_ws, dim3((unsigned)((h->n + kWave - 1) / kWave)),
                       dim3(kWsBlock), 0, static_cast<hipStream_t>(stream), h->hot, h->rare_dev,
                       reinterpret_cast<float*>(h->buf.player_x), h->stride, h->n, n_steps, actions_dev, action_kind,
                       ro, h->out);
    HIP_TRY(hipGetLastError());
    std::snprintf(h->kernel_name, sizeof h->kernel_name, "s2d_reach_rollout_ws_kernel<%s,noise=%d,rec=%d,nt=%d>", mode_names[h->mode],
                  h->noise ? 1 : 0, rec, ro.nt ? 1 : 0);
    h->last_kernel = h->kernel_name;
    return S2D_OK;
  }
  hipLaunchKernelGGL(table[h->mode][h->noise ? 1 : 0], dim3(grid_for(h->n)), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), h->hot, h->rare_dev,
                     reinterpret_cast<float*>(h->buf.player_x), h->stride, h->n, n_steps, actions_dev, action_kind,
                     ro, h->out);
  HIP_TRY(hipGetLastError());
  std::snprintf(h->kernel_name, sizeof h->kernel_name, "s2d_reach_rollout_kernel<%s,noise=%d>", mode_names[h->mode], h->noise ? 1 : 0);
  h->last_kernel = h->kernel_name;
  return S2D_OK;
}

S2D_API int s2d_world_model(S2DHandle h, const S2DWorldModel* out, void* stream) {
  if (!h || !out) return fail(S2D_EINVAL, "NULL argument");
  DeviceGuard guard(h->device);
  hipLaunchKernelGGL(s2d_world_model_kernel, dim3(grid_for(h->n)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float*>(h->buf.player_x), h->stride, h->n, *out);
  HIP_TRY(hipGetLastError());
  h->last_kernel = "s2d_world_model_kernel";
  return S2D_OK;
}

S2D_API int s2d_stats_reset(S2DHandle h, void* stream) {
  if (!h) return fail(S2D_EINVAL, "NULL handle");
  DeviceGuard guard(h->device);
  HIP_TRY(hipMemsetAsync(h->buf.stats, 0, (size_t)S2D_STATS_ROWS(h->n) * 8 * sizeof(unsigned long long), static_cast<hipStream_t>(stream)));
  return S2D_OK;
}

S2D_API const char* s2d_kernel_name(S2DHandle h) { return h ? h->last_kernel : ""; }

S2D_API int s2d_validate_state(S2DHandle h, uint32_t* counts_dev, void* stream) {
  if (!h || !counts_dev) return fail(S2D_EINVAL, "NULL argument");
  DeviceGuard guard(h->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  HIP_TRY(hipMemsetAsync(counts_dev, 0, SV_WORDS * sizeof(uint32_t), st));
  hipLaunchKernelGGL(s2d_validate_kernel, dim3(grid_for(h->n)), dim3(kBlock), 0, st, h->hot,
                     reinterpret_cast<const float*>(h->buf.player_x), h->stride, h->n, h->buf.obs, counts_dev);
  HIP_TRY(hipGetLastError());
  return S2D_OK;
}

S2D_API int s2d_set_seed(S2DHandle h, uint64_t seed, void* stream) {
  if (!h) return fail(S2D_EINVAL, "NULL handle");
  // The per-step API's prepared episodes were drawn with the old key: drop their tags, stream-ordered behind every launch
  // already queued on `stream` (the refill workgroups of an s2d_step still in flight there finish first) and ahead of the next
  // one, which draws inline and whose refill workgroups prepare the slots again with the new key.  The key itself travels in the
  // kernarg of later launches, so it needs no device write.  Capturable into a hipGraph like every other entry point.
  DeviceGuard guard(h->device);
  const ArenaLayout L = layout_for(h->n);
  HIP_TRY(hipMemsetAsync(h->arena + L.prep + (size_t)2 * PS_WORDS * (size_t)h->stride * 4, 0, 2 * (size_t)h->stride * 4,
                         static_cast<hipStream_t>(stream)));
  h->cfg.seed = seed;
  h->hot.seed_lo = (uint32_t)seed; h->hot.seed_hi = (uint32_t)(seed >> 32);
  return S2D_OK;
}

S2D_API int s2d_debug_eval(int op, const void* in_dev, void* out_dev, int64_t n, void* stream) {
  if (!in_dev || !out_dev || n <= 0 || op < 0 || op > 10 || (op == 9 && n % 256 != 0)) return fail(S2D_EINVAL, "bad s2d_debug_eval argument");
  hipLaunchKernelGGL(s2d_debug_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), op, static_cast<const float*>(in_dev),
                     static_cast<float*>(out_dev), n);
  HIP_TRY(hipGetLastError());
  return S2D_OK;
}
