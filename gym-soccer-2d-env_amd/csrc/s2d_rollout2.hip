// s2d_rollout2.hip -- the fused rollout pipeline with TWO envs per lane (round 4).
//
// Same software pipeline as s2d_reach_rollout_ws_kernel (s2d_engine.hip: policy | simulate | agent | ball waves on steps
// t, t-1, t-2, t-2, double-buffered LDS hand-offs, one s_barrier per iteration, prepared episodes in LDS slots), same functions of
// s2d_device.h in the same order -- results are bit-identical -- but a workgroup owns 128 envs and lane l of every role wave
// carries envs 2l and 2l + 1:
//   * the record leaves as whole lines: done / result are ONE 2-byte store per lane = one 128-byte line per wave-instruction
//     (the 64-env kernel wrote 64-byte halves of lines shared with the neighbouring workgroup, which is what held its write
//     stream at 4.3-4.5 TB/s: profiles/r03/store_pattern.txt, profiles/r04/store_pattern_e2.txt), reward / action are 8-byte
//     stores (512 contiguous bytes), the observation block of a step is 5 120 contiguous bytes = five 16-byte-per-lane stores;
//   * every wave carries two independent dependency chains, and the per-iteration overhead (barrier, loop, addresses, LDS
//     hand-off instructions -- all 8-byte LDS accesses now) is paid once per 128 envs.
// Launched for batches that are a multiple of 128 envs with a complete, suitably aligned record; everything else runs the
// 64-env kernels.  Reference semantics: Soccer2DEnv.step (soccer_2d_env.py:226-269) with the ReachBallEnv hooks
// (reach_ball_env.py:53-161), T steps fused.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "s2d_kernels.h"

static constexpr int kE = 2;                            // envs per lane
static constexpr int kGroup = kWave * kE;               // envs per workgroup
static constexpr int kTile2 = kGroup * S2D_OBS_DIM;     // floats of one step's observation block

typedef float v2f32_t __attribute__((ext_vector_type(2)));
typedef int v2i32_t __attribute__((ext_vector_type(2)));

// 8-byte LDS accesses of a lane's pair of envs (row = one word of 128 envs, col = 2 * lane)
S2D_DEV float2 ld2(const float* row, int col) { return *reinterpret_cast<const float2*>(row + col); }
S2D_DEV void st2(float* row, int col, float a, float b) { *reinterpret_cast<float2*>(row + col) = make_float2(a, b); }

// record stores of the pair (NT: the record does not fit the Infinity Cache, see rec_store16 in s2d_kernels.h)
template <bool NT> S2D_DEV void rec2_f32(float* p, float a, float b) {
  const v2f32_t v = {a, b};
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v2f32_t*>(p)); else *reinterpret_cast<v2f32_t*>(p) = v;
}
template <bool NT> S2D_DEV void rec2_i32(int32_t* p, int a, int b) {
  const v2i32_t v = {a, b};
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v2i32_t*>(p)); else *reinterpret_cast<v2i32_t*>(p) = v;
}
template <bool NT> S2D_DEV void rec2_u8(uint8_t* p, int a, int b) {
  const unsigned short v = (unsigned short)((a & 0xff) | ((b & 0xff) << 8));
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<unsigned short*>(p)); else *reinterpret_cast<unsigned short*>(p) = v;
}
template <bool NT> S2D_DEV void rec_f32x4(float4* p, const float4& v) {
  const v4f32_t w = {v.x, v.y, v.z, v.w};
  if (NT) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(w) : "memory");
  else *p = v;
}

// the caller's actions of the pair at row offset idx (even) -- layouts of include/s2d.h
template <int MODE>
S2D_DEV void load_action2(const void* __restrict__ actions, int kind, int64_t idx, Action4& a0, Action4& a1) {
  a0 = Action4{0.0f, 0.0f, 0.0f, 0.0f}; a1 = a0;
  if (MODE == S2D_MODE_DISCRETE) {
    if (kind == S2D_ACT_DISCRETE_I64) {
      const longlong2 v = *reinterpret_cast<const longlong2*>(static_cast<const long long*>(actions) + idx);
      a0.a0 = (float)v.x; a1.a0 = (float)v.y;
    } else {
      const int2 v = *reinterpret_cast<const int2*>(static_cast<const int32_t*>(actions) + idx);
      a0.a0 = (float)v.x; a1.a0 = (float)v.y;
    }
  } else if (MODE == S2D_MODE_CONT1) {
    const float2 v = *reinterpret_cast<const float2*>(static_cast<const float*>(actions) + idx);
    a0.a0 = v.x; a1.a0 = v.y;
  } else {
    const float4 v = static_cast<const float4*>(actions)[idx], w = static_cast<const float4*>(actions)[idx + 1];
    a0 = Action4{v.x, v.y, v.z, v.w}; a1 = Action4{w.x, w.y, w.z, w.w};
  }
}
// experiment build (-DS2D_STAMPS): busy clocks of every role wave (barrier release -> arrival at the next barrier) and the clocks of
// its whole loop, written by lane 0 into the terminal_obs rows of the group's first envs (role r: row first + 2 r, words 0 / 1)
#ifdef S2D_STAMPS
#define WS2_STAMP_DECL uint64_t st_busy = 0, st_t0 = __builtin_amdgcn_s_memtime(); const uint64_t st_begin = st_t0
#define WS2_BARRIER() do { st_busy += __builtin_amdgcn_s_memtime() - st_t0; __syncthreads(); st_t0 = __builtin_amdgcn_s_memtime(); } while (0)
#define WS2_STAMP_STORE() do { if (lane == 0) { float* q_ = o.terminal_obs + (first + 2 * role) * S2D_OBS_DIM; \
    q_[0] = (float)st_busy; q_[1] = (float)(__builtin_amdgcn_s_memtime() - st_begin); } } while (0)
#else
#define WS2_STAMP_DECL do {} while (0)
#define WS2_BARRIER() __syncthreads()
#define WS2_STAMP_STORE() do {} while (0)
#endif

#ifndef S2D_PRIO2_P
#define S2D_PRIO2_P 0
#endif
#ifndef S2D_PRIO2_S
#define S2D_PRIO2_S 1
#endif
#ifndef S2D_PRIO2_A
#define S2D_PRIO2_A 2
#endif
#ifndef S2D_PRIO2_B
#define S2D_PRIO2_B 3
#endif

template <int MODE, bool NOISE, bool NT>
__global__ __launch_bounds__(kWsBlock) void s2d_reach_rollout_ws2_kernel(S2DHot p_sgpr, const S2DRare* __restrict__ rp,
                                                                         float* __restrict__ S, int64_t stride, int64_t n,
                                                                         int n_steps, const void* __restrict__ actions, int kind,
                                                                         RolloutOut ro, StepOut o) {
  constexpr int kActWords = NOISE ? (int)WA_WORDS : (int)WA_NPM;
  __shared__ __attribute__((aligned(16))) float act[2][kActWords][kGroup];   // decoded command (+ prepared noise) of step t
  __shared__ __attribute__((aligned(16))) float snap[2][WS_WORDS][kGroup];   // post-cycle snapshot of step t
  __shared__ __attribute__((aligned(16))) float slots[kSlots][SL_WORDS][kGroup];   // prepared episodes first_ep + k of every env
  __shared__ __attribute__((aligned(16))) float tile[2][kTile2];             // observation rows of step t (row-major [128][10])
  __shared__ float4 act_lut[kWave];                        // decoded commands of a small discrete action space
  __shared__ float ep_lds[S2D_TAB_MAX];                    // dash-only fast path: effort * power by step number
  __shared__ float2 sc_lut[361];                           //   and (sin, cos) of the whole degrees -180 .. 180
  const int lane = threadIdx.x & (kWave - 1);
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);   // 0 policy, 1 simulate, 2 agent, 3 ball
  const int64_t first = (int64_t)blockIdx.x * kGroup;      // every group is complete (the host launches this kernel only then)
  const int col = kE * lane;
  const int64_t i0 = first + col;                          // this lane's envs: i0, i0 + 1
  const int n_iter = n_steps + 3;
  const uint64_t gid0 = (((uint64_t)p_sgpr.gid_hi << 32) | p_sgpr.gid_lo) + (uint64_t)i0;

  // ---- before the loop: the three waves that idle while the pipeline fills prepare one future episode of every env each
  if (role != 1 && p_sgpr.auto_reset) {                    // wave-uniform: the wave draws together (reset_sample_coop), env by env
    const uint2 ep0 = *reinterpret_cast<const uint2*>(S + F_EPISODE * stride + i0);
    const int k = role == 0 ? 0 : role - 1;
    uint32_t* const scratch = reinterpret_cast<uint32_t*>(&tile[0][0]) + role * kWave;   // the observation tiles are idle here
    const S2DRare r = *rp;
#pragma unroll
    for (int e = 0; e < kE; ++e) {
      const uint64_t gid = gid0 + (uint64_t)e;
      const NextEpisode q = episode_prepare_coop<NOISE>(p_sgpr, rp, r, (uint32_t)gid, (uint32_t)(gid >> 32),
                                                        (e ? ep0.y : ep0.x) + 1u + (uint32_t)k, true, lane, scratch);
      slot_put<kGroup>(slots[k], col + e, q, first_obs(p_sgpr, q));
    }
  }

  if (role == 0) {
    // ------------------------------------------------------------------ P-wave
    __builtin_amdgcn_s_setprio(NOISE ? 1 : S2D_PRIO2_P);
    const S2DHot& p = p_sgpr;
    const bool use_k = uses_policy_step<MODE, NOISE>(kind);
    uint32_t* const kplane = reinterpret_cast<uint32_t*>(S + F_POLICY * stride);
    uint32_t gl[kE], gh[kE], k0[kE] = {0u, 0u};
    if (use_k) { const uint2 kk = *reinterpret_cast<const uint2*>(kplane + i0); k0[0] = kk.x; k0[1] = kk.y; }
#pragma unroll
    for (int e = 0; e < kE; ++e) { const uint64_t g = gid0 + (uint64_t)e; gl[e] = (uint32_t)g; gh[e] = (uint32_t)(g >> 32); }
    U4 quad[kE] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, squad[kE] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, nblk[kE] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    float dir[kE] = {0.0f, 0.0f}; int cmd[kE] = {0, 0};
    int64_t row = 0;
    // in-engine policy of a small discrete action space: one table entry per action (see s2d_reach_rollout_ws_kernel)
    const bool lut = MODE == S2D_MODE_DISCRETE && kind == S2D_ACT_RANDOM && p.n_actions <= kWave;
    if (lut && lane < p.n_actions) {
      int c0; float pw, d0;
      action_map<MODE>(p, Action4{(float)lane, 0.0f, 0.0f, 0.0f}, 0.0f, c0, pw, d0);
      const CmdPrep c = cmd_prepare(p, c0, pw, d0);
      act_lut[lane] = make_float4(c.power, c.dir, c.dir_rate, d0);
    }
    __syncthreads();                                       // prepared episodes published
    WS2_STAMP_DECL;
    for (int s = 0; s < n_iter; ++s) {
      if (s < n_steps) {
        const int b = s & 1;
        CmdPrep c[kE];
        if (lut) {
          int a[kE];
#pragma unroll
          for (int e = 0; e < kE; ++e) {
            const uint32_t k = k0[e] + (uint32_t)s;
            if (s == 0 || (k & 3u) == 0u) quad[e] = policy_quad(p, gl[e], gh[e], k, S2D_ST_POLICY);
            a[e] = (int)rnd_below(quad_word(quad[e], k), (uint32_t)p.n_actions);
            const float4 e4 = act_lut[a[e]];
            c[e] = CmdPrep{e4.x, e4.y, e4.z}; dir[e] = e4.w; cmd[e] = S2D_CMD_DASH;
          }
          rec2_i32<NT>(static_cast<int32_t*>(ro.action) + row + i0, a[0], a[1]);
        } else {
          Action4 a[kE];
          if (kind == S2D_ACT_RANDOM) {
#pragma unroll
            for (int e = 0; e < kE; ++e) {
              const uint32_t k = k0[e] + (uint32_t)s;
              a[e] = random_action<MODE>(p, gl[e], gh[e], k, quad[e], s == 0 || (k & 3u) == 0u);
            }
          } else {
            load_action2<MODE>(actions, kind, row + i0, a[0], a[1]);
          }
          if (MODE == S2D_MODE_DISCRETE) rec2_i32<NT>(static_cast<int32_t*>(ro.action) + row + i0, (int32_t)a[0].a0, (int32_t)a[1].a0);
          else if (MODE == S2D_MODE_CONT1) rec2_f32<NT>(static_cast<float*>(ro.action) + row + i0, a[0].a0, a[1].a0);
          else {
            float4* const d = static_cast<float4*>(ro.action) + row + i0;
            rec_f32x4<NT>(d, make_float4(a[0].a0, a[0].a1, a[0].a2, a[0].a3));
            rec_f32x4<NT>(d + 1, make_float4(a[1].a0, a[1].a1, a[1].a2, a[1].a3));
          }
#pragma unroll
          for (int e = 0; e < kE; ++e) {
            const uint32_t k = k0[e] + (uint32_t)s;
            c[e] = decode_action<MODE>(p, a[e], gl[e], gh[e], k, s == 0 || (k & 3u) == 0u, squad[e], cmd[e], dir[e]);
          }
        }
        if (MODE == S2D_MODE_TURN4) st2(act[b][WA_CMD], col, __int_as_float(cmd[0]), __int_as_float(cmd[1]));
        st2(act[b][WA_POWER], col, c[0].power, c[1].power);
        st2(act[b][WA_DIR], col, c[0].dir, c[1].dir);
        st2(act[b][WA_RATE], col, c[0].dir_rate, c[1].dir_rate);
        if constexpr (NOISE) {                             // the state-independent half of this cycle's noise (table: simulate wave)
          float pm[kE], psn[kE], pcs[kE], bm[kE], bsn[kE], bcs[kE], tu[kE];
#pragma unroll
          for (int e = 0; e < kE; ++e) {
            const uint32_t k = k0[e] + (uint32_t)s;
            const NoiseWords nw = noise_words(p, gl[e], gh[e], k, S2D_ST_NOISE, cmd[e] == S2D_CMD_TURN, nblk[e], s == 0 || (k & 1u) == 0u);
            const float2 ps = sc_lut[noise_dir_index(nw.wp)], bs = sc_lut[noise_dir_index(nw.wb)];
            pm[e] = noise_mag(nw.wp); psn[e] = ps.x; pcs[e] = ps.y;
            bm[e] = noise_mag(nw.wb); bsn[e] = bs.x; bcs[e] = bs.y;
            tu[e] = nw.tu;
          }
          st2(act[b][WA_NPM], col, pm[0], pm[1]); st2(act[b][WA_NPS], col, psn[0], psn[1]); st2(act[b][WA_NPC], col, pcs[0], pcs[1]);
          st2(act[b][WA_NBM], col, bm[0], bm[1]); st2(act[b][WA_NBS], col, bsn[0], bsn[1]); st2(act[b][WA_NBC], col, bcs[0], bcs[1]);
          if (MODE == S2D_MODE_TURN4) st2(act[b][WA_NTU], col, tu[0], tu[1]);
        }
        row += n;
      }
      WS2_BARRIER();
    }
    WS2_STAMP_STORE();
    if (use_k) *reinterpret_cast<uint2*>(kplane + i0) = make_uint2(k0[0] + (uint32_t)n_steps, k0[1] + (uint32_t)n_steps);
    *reinterpret_cast<float2*>(o.action_dir + i0) = make_float2(dir[0], dir[1]);
    *reinterpret_cast<unsigned short*>(o.action_cmd + i0) = (unsigned short)((cmd[0] & 0xff) | ((cmd[1] & 0xff) << 8));
  } else if (role == 1) {
    // ------------------------------------------------------------------ S-wave
    __builtin_amdgcn_s_setprio(NOISE ? 3 : S2D_PRIO2_S);
    const S2DHot p = hot_in_vgprs(p_sgpr);
    Env e[kE];
    uint32_t gl[kE], gh[kE];
    int nth[kE] = {0, 0}, j[kE] = {0, 0};                  // episodes this env began in this launch; slot of the next one
    {
      float2 w[F_COUNT];
#pragma unroll
      for (int f = 0; f < F_COUNT; ++f) w[f] = (f == F_PREV_DIST || f == F_PREV_ANGLE || f == F_POLICY) ? make_float2(0.0f, 0.0f)
                                                  : *reinterpret_cast<const float2*>(S + (int64_t)f * stride + i0);
#pragma unroll
      for (int q = 0; q < kE; ++q) {
        auto pick = [&](int f) { return q ? w[f].y : w[f].x; };
        e[q].px = pick(F_PX); e[q].py = pick(F_PY); e[q].vx = pick(F_VX); e[q].vy = pick(F_VY); e[q].body = pick(F_BODY);
        e[q].stamina = pick(F_STAMINA); e[q].effort = pick(F_EFFORT); e[q].recovery = pick(F_RECOVERY); e[q].capacity = pick(F_CAPACITY);
        e[q].bx = pick(F_BX); e[q].by = pick(F_BY); e[q].bvx = pick(F_BVX); e[q].bvy = pick(F_BVY);
        e[q].prev_dist = 0.0f; e[q].prev_angle = 0.0f;     // the A-wave's
        e[q].step_number = __float_as_int(pick(F_STEP)); e[q].cycle = __float_as_int(pick(F_CYCLE));
        e[q].episode = __float_as_int(pick(F_EPISODE));
        const uint64_t g = gid0 + (uint64_t)q; gl[q] = (uint32_t)g; gh[q] = (uint32_t)(g >> 32);
        asm volatile("" ::"v"(e[q].px), "v"(e[q].py), "v"(e[q].vx), "v"(e[q].vy), "v"(e[q].body), "v"(e[q].stamina), "v"(e[q].effort),
                     "v"(e[q].recovery), "v"(e[q].capacity), "v"(e[q].bx), "v"(e[q].by), "v"(e[q].bvx), "v"(e[q].bvy),
                     "v"(e[q].step_number), "v"(e[q].cycle), "v"(e[q].episode));
      }
    }
    // dash-only fast path (s2d_device.h, S2DTables): taken by a group whose envs all sit on the stamina table and have whole-degree
    // body angles; anything else runs the generic loop with the same results
    const S2DTables* const tb = tables_of(rp);
    const int tab_len = MODE != S2D_MODE_TURN4 ? rp->tab_len : 0;
    bool fast = false;
    if (tab_len > 0) {
      bool ok = true;
#pragma unroll
      for (int q = 0; q < kE; ++q) {
        const int sn = e[q].step_number;
        bool okq = sn >= 0 && sn < tab_len;
        const int t = okq ? sn : 0;
        okq = okq && e[q].stamina == tb->stamina[t] && e[q].effort == tb->effort[t] && e[q].recovery == tb->recovery[t] &&
              e[q].capacity == tb->capacity[t] && e[q].body == rintf(e[q].body) && fabsf(e[q].body) <= 180.0f;
        ok = ok && okq;
      }
      fast = __ballot(!ok) == 0ull;
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
    WS2_STAMP_DECL;
    auto loop = [&](auto fast_tag) {
      constexpr bool FAST = decltype(fast_tag)::value;
      auto simulate_iteration = [&](int s, auto steady_tag) {   // (three stretches: fill / steady / drain)
        constexpr bool STEADY = decltype(steady_tag)::value;
        if (STEADY || (s >= 1 && s <= n_steps)) {          // step s - 1
          const int b = (s - 1) & 1;
          const float2 a_dir = ld2(act[b][WA_DIR], col), a_rate = ld2(act[b][WA_RATE], col);
          float2 a_pow = make_float2(0.0f, 0.0f), a_cmd = make_float2(0.0f, 0.0f);
          if (!FAST) a_pow = ld2(act[b][WA_POWER], col);
          if (MODE == S2D_MODE_TURN4) a_cmd = ld2(act[b][WA_CMD], col);
          float2 n_pm, n_ps, n_pc, n_bm, n_bs, n_bc, n_tu = make_float2(0.0f, 0.0f);
          if constexpr (NOISE) {
            n_pm = ld2(act[b][WA_NPM], col); n_ps = ld2(act[b][WA_NPS], col); n_pc = ld2(act[b][WA_NPC], col);
            n_bm = ld2(act[b][WA_NBM], col); n_bs = ld2(act[b][WA_NBS], col); n_bc = ld2(act[b][WA_NBC], col);
            if (MODE == S2D_MODE_TURN4) n_tu = ld2(act[b][WA_NTU], col);
          }
          int fl[kE];
#pragma unroll
          for (int q = 0; q < kE; ++q) {
            auto pk = [&](const float2& v) { return q ? v.y : v.x; };
            int cmd = S2D_CMD_DASH;                        // only the turning mode has another command
            if (MODE == S2D_MODE_TURN4) cmd = __float_as_int(pk(a_cmd));
            NoiseIn nz{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (NOISE) nz = NoiseIn{pk(n_pm), pk(n_ps), pk(n_pc), pk(n_bm), pk(n_bs), pk(n_bc), pk(n_tu)};
            float d2;
            if constexpr (FAST) {
              const float ep = ep_lds[e[q].step_number];   // effort * power of the dash at this step number
              const float2 sc = sc_lut[(int)norm_deg(e[q].body + pk(a_dir)) + 180];
              e[q].step_number += 1;                       // reach_ball_env.py:55
              d2 = sim_cycle_dash_fast<NOISE>(p, rp, e[q], ep, pk(a_rate), sc.x, sc.y, nz);
            } else {
              const CmdPrep c{pk(a_pow), pk(a_dir), pk(a_rate)};
              e[q].step_number += 1;                       // reach_ball_env.py:55
              d2 = sim_cycle<NOISE, true>(p, rp, e[q], cmd, c, nz);
            }
            fl[q] = judge_sq(p, e[q].px, e[q].py, d2, e[q].step_number);
          }
          st2(snap[b][WS_PX], col, e[0].px, e[1].px); st2(snap[b][WS_PY], col, e[0].py, e[1].py);
          st2(snap[b][WS_BODY], col, e[0].body, e[1].body);
          st2(snap[b][WS_BX], col, e[0].bx, e[1].bx); st2(snap[b][WS_BY], col, e[0].by, e[1].by);
          st2(snap[b][WS_BVX], col, e[0].bvx, e[1].bvx); st2(snap[b][WS_BVY], col, e[0].bvy, e[1].bvy);
          st2(snap[b][WS_FLAGS], col, __int_as_float(fl[0] | (j[0] << 8)), __int_as_float(fl[1] | (j[1] << 8)));   // bits 8..: slot of the next episode
#pragma unroll
          for (int q = 0; q < kE; ++q) {
            if (fl[q] && p.auto_reset) {                   // rare: the prepared episode is a copy
              if (nth[q] >= kSlots) {                      // more than kSlots episodes ended in this launch: prepare inline
                const S2DRare r = *rp;
                const NextEpisode ne = episode_prepare<NOISE>(p, rp, r, gl[q], gh[q], (uint32_t)e[q].episode + 1u);
                slot_put<kGroup>(slots[j[q]], col + q, ne, first_obs(p, ne));
              }
              episode_begin(e[q], slot_take<kGroup>(slots[j[q]], col + q, rst));
              nth[q] += 1; j[q] = (j[q] + 1 == kSlots) ? 0 : j[q] + 1;
            }
          }
        }
        WS2_BARRIER();
      };
      int s = 0;
      for (; s < 1 && s < n_iter; ++s) simulate_iteration(s, std::false_type{});
      for (; s <= n_steps; ++s) simulate_iteration(s, std::true_type{});
      for (; s < n_iter; ++s) simulate_iteration(s, std::false_type{});
    };
    if (fast) {
      loop(std::true_type{});
#pragma unroll
      for (int q = 0; q < kE; ++q) {                       // the stamina words the fast loop did not carry
        const int t = e[q].step_number;
        e[q].stamina = tb->stamina[t]; e[q].effort = tb->effort[t]; e[q].recovery = tb->recovery[t]; e[q].capacity = tb->capacity[t];
      }
    } else {
      loop(std::false_type{});
    }
    WS2_STAMP_STORE();
    {                                                      // prev_dist / prev_angle belong to the A-wave
      auto put = [&](int f, float a, float b) { *reinterpret_cast<float2*>(S + (int64_t)f * stride + i0) = make_float2(a, b); };
      put(F_PX, e[0].px, e[1].px); put(F_PY, e[0].py, e[1].py); put(F_VX, e[0].vx, e[1].vx); put(F_VY, e[0].vy, e[1].vy);
      put(F_BODY, e[0].body, e[1].body);
      put(F_STAMINA, e[0].stamina, e[1].stamina); put(F_EFFORT, e[0].effort, e[1].effort);
      put(F_RECOVERY, e[0].recovery, e[1].recovery); put(F_CAPACITY, e[0].capacity, e[1].capacity);
      put(F_BX, e[0].bx, e[1].bx); put(F_BY, e[0].by, e[1].by); put(F_BVX, e[0].bvx, e[1].bvx); put(F_BVY, e[0].bvy, e[1].bvy);
      put(F_STEP, __int_as_float(e[0].step_number), __int_as_float(e[1].step_number));
      put(F_CYCLE, __int_as_float(e[0].cycle), __int_as_float(e[1].cycle));
      put(F_EPISODE, __int_as_float(e[0].episode), __int_as_float(e[1].episode));
    }
  } else if (role == 2) {
    // ------------------------------------------------------------------ A-wave (player half, reward, labels)
    __builtin_amdgcn_s_setprio(NOISE ? 2 : S2D_PRIO2_A);
    const S2DHot p = hot_in_vgprs(p_sgpr);                 // no kernarg re-loads (s_load + s_waitcnt) inside the loop
    const bool auto_reset = p_sgpr.auto_reset != 0;
    float prev_dist[kE], prev_angle[kE];
    {
      const float2 pd = *reinterpret_cast<const float2*>(S + F_PREV_DIST * stride + i0);
      const float2 pa = *reinterpret_cast<const float2*>(S + F_PREV_ANGLE * stride + i0);
      prev_dist[0] = pd.x; prev_dist[1] = pd.y; prev_angle[0] = pa.x; prev_angle[1] = pa.y;
      asm volatile("" ::"v"(prev_dist[0]), "v"(prev_dist[1]), "v"(prev_angle[0]), "v"(prev_angle[1]));
    }
    float oa[kE][4] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};   // words 0..3 of the observation rows
    float reward[kE] = {0.0f, 0.0f}; int res[kE] = {0, 0}, done[kE] = {0, 0};
    unsigned int cnt1 = 0, cnt2 = 0, cnt3 = 0;
    unsigned long long* const srow = stats_row(o.stats, first);   // the group's first row of the episode counters (stored after the loop)
    const unsigned long long sold = stats_load(srow, lane);
    int64_t row = 0;
    __syncthreads();                                       // prepared episodes published
    WS2_STAMP_DECL;
    auto agent_iteration = [&](int s, auto steady_tag) {   // (three stretches: fill / steady / drain)
      constexpr bool STEADY = decltype(steady_tag)::value;
      if (STEADY || (s >= 2 && s < n_steps + 2)) {         // step s - 2
        const int b = s & 1;
        const float2 s_px = ld2(snap[b][WS_PX], col), s_py = ld2(snap[b][WS_PY], col), s_body = ld2(snap[b][WS_BODY], col);
        const float2 s_bx = ld2(snap[b][WS_BX], col), s_by = ld2(snap[b][WS_BY], col), s_fw = ld2(snap[b][WS_FLAGS], col);
#pragma unroll
        for (int q = 0; q < kE; ++q) {
          auto pk = [&](const float2& v) { return q ? v.y : v.x; };
          const float px = pk(s_px), py = pk(s_py), body = pk(s_body), bx = pk(s_bx), by = pk(s_by);
          const int fw = __float_as_int(pk(s_fw));
          const int flags = fw & 0xff;
          const float dist = hypot2(bx - px, by - py);
          float ob[S2D_OBS_DIM];
          const float rel = observe_player(p, px, py, body, bx, by, ob);
          reward[q] = reward_of(prev_dist[q], prev_angle[q], dist, rel, flags, res[q]);
          prev_dist[q] = dist; prev_angle[q] = rel;
          done[q] = flags ? 1 : 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) oa[q][k] = ob[k];
          if (flags && auto_reset) {                       // rare: terminal row, then the new episode's first obs
            const float (*sl)[kGroup] = slots[fw >> 8];
            float* const term_row = o.terminal_obs + (i0 + q) * S2D_OBS_DIM;
#pragma unroll
            for (int k = 0; k < 4; ++k) term_row[k] = oa[q][k];
#pragma unroll
            for (int k = 0; k < 4; ++k) oa[q][k] = sl[SL_FIRST + k][col + q];
            prev_dist[q] = sl[SL_DIST][col + q]; prev_angle[q] = sl[SL_REL][col + q];   // reach_ball_env.py:166 carry seeded
          }
          cnt1 += res[q] == S2D_RESULT_GOAL; cnt2 += res[q] == S2D_RESULT_OUT; cnt3 += res[q] == S2D_RESULT_TIMEOUT;
        }
        rec2_f32<NT>(ro.reward + row + i0, reward[0], reward[1]);
        rec2_u8<NT>(ro.done + row + i0, done[0], done[1]);
        rec2_u8<NT>(ro.result + row + i0, res[0], res[1]);
        float* const t = &tile[b][col * S2D_OBS_DIM];      // rows 2l, 2l + 1 = 20 floats at a 16-byte-aligned address
        *reinterpret_cast<float4*>(t) = make_float4(oa[0][0], oa[0][1], oa[0][2], oa[0][3]);
        *reinterpret_cast<float2*>(t + 10) = make_float2(oa[1][0], oa[1][1]);
        *reinterpret_cast<float2*>(t + 12) = make_float2(oa[1][2], oa[1][3]);
        row += n;
      }
      WS2_BARRIER();
    };
    {
      int s = 0;
      for (; s < 3 && s < n_iter; ++s) agent_iteration(s, std::false_type{});
      for (; s < n_steps + 2; ++s) agent_iteration(s, std::true_type{});
      for (; s < n_iter; ++s) agent_iteration(s, std::false_type{});
    }
    WS2_STAMP_STORE();
    *reinterpret_cast<float2*>(S + F_PREV_DIST * stride + i0) = make_float2(prev_dist[0], prev_dist[1]);
    *reinterpret_cast<float2*>(S + F_PREV_ANGLE * stride + i0) = make_float2(prev_angle[0], prev_angle[1]);
    *reinterpret_cast<float2*>(o.reward + i0) = make_float2(reward[0], reward[1]);
    *reinterpret_cast<unsigned short*>(o.done + i0) = (unsigned short)(done[0] | (done[1] << 8));
    *reinterpret_cast<unsigned short*>(o.result + i0) = (unsigned short)(res[0] | (res[1] << 8));
    {                                                      // last observation, player half (rows i0, i0 + 1 of [N][10])
      float* const d = o.obs + i0 * S2D_OBS_DIM;
      *reinterpret_cast<float4*>(d) = make_float4(oa[0][0], oa[0][1], oa[0][2], oa[0][3]);
      *reinterpret_cast<float2*>(d + 10) = make_float2(oa[1][0], oa[1][1]);
      *reinterpret_cast<float2*>(d + 12) = make_float2(oa[1][2], oa[1][3]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      cnt1 += __shfl_xor(cnt1, off); cnt2 += __shfl_xor(cnt2, off); cnt3 += __shfl_xor(cnt3, off);
    }
    stats_store(srow, lane, sold, first == 0 ? (unsigned long long)n * (unsigned long long)n_steps : 0ull, cnt1, cnt2, cnt3);
  } else {
    // ------------------------------------------------------------------ B-wave (ball half, observation stream)
    __builtin_amdgcn_s_setprio(NOISE ? 2 : S2D_PRIO2_B);
    const S2DHot p = hot_in_vgprs(p_sgpr);                 // no kernarg re-loads inside the loop
    const bool auto_reset = p_sgpr.auto_reset != 0;
    float ob6[kE][S2D_OBS_DIM];                            // only words 4..9 are produced here
#pragma unroll
    for (int q = 0; q < kE; ++q)
#pragma unroll
      for (int k = 0; k < S2D_OBS_DIM; ++k) ob6[q][k] = 0.0f;
    const float4* const obs_dst = reinterpret_cast<const float4*>(ro.obs + first * S2D_OBS_DIM) + lane;   // + step * (n * 10 / 4)
    const int64_t row4 = n * S2D_OBS_DIM / 4;              // float4 units per step (n is a multiple of 128)
    __syncthreads();                                       // prepared episodes published
    WS2_STAMP_DECL;
    auto ball_iteration = [&](int s, auto steady_tag) {
      constexpr bool STEADY = decltype(steady_tag)::value;
      if (STEADY || s >= 3) {                              // observation block of step s - 3, completed in iteration s - 1
        const float4* const t4 = reinterpret_cast<const float4*>(tile[(s - 1) & 1]) + lane;
        float4* const d4 = const_cast<float4*>(obs_dst) + (int64_t)(s - 3) * row4;
#pragma unroll
        for (int k = 0; k < (kTile2 / 4) / kWave; ++k) rec_f32x4<NT>(d4 + k * kWave, t4[k * kWave]);
      }
      if (STEADY || (s >= 2 && s < n_steps + 2)) {         // step s - 2
        const int b = s & 1;
        const float2 s_bx = ld2(snap[b][WS_BX], col), s_by = ld2(snap[b][WS_BY], col);
        const float2 s_bvx = ld2(snap[b][WS_BVX], col), s_bvy = ld2(snap[b][WS_BVY], col), s_fw = ld2(snap[b][WS_FLAGS], col);
#pragma unroll
        for (int q = 0; q < kE; ++q) {
          auto pk = [&](const float2& v) { return q ? v.y : v.x; };
          const int fw = __float_as_int(pk(s_fw));
          observe_ball(p, pk(s_bx), pk(s_by), pk(s_bvx), pk(s_bvy), ob6[q]);
          if ((fw & 0xff) && auto_reset) {                 // rare: terminal row, then the new episode's first obs
            const float (*sl)[kGroup] = slots[fw >> 8];
            float* const term_row = o.terminal_obs + (i0 + q) * S2D_OBS_DIM;
#pragma unroll
            for (int k = 4; k < S2D_OBS_DIM; ++k) term_row[k] = ob6[q][k];
#pragma unroll
            for (int k = 4; k < S2D_OBS_DIM; ++k) ob6[q][k] = sl[SL_FIRST + k][col + q];
          }
        }
        float* const t = &tile[b][col * S2D_OBS_DIM];
        *reinterpret_cast<float4*>(t + 4) = make_float4(ob6[0][4], ob6[0][5], ob6[0][6], ob6[0][7]);
        *reinterpret_cast<float2*>(t + 8) = make_float2(ob6[0][8], ob6[0][9]);
        *reinterpret_cast<float2*>(t + 14) = make_float2(ob6[1][4], ob6[1][5]);
        *reinterpret_cast<float4*>(t + 16) = make_float4(ob6[1][6], ob6[1][7], ob6[1][8], ob6[1][9]);
      }
      WS2_BARRIER();
    };
    {
      int s = 0;
      for (; s < 3 && s < n_iter; ++s) ball_iteration(s, std::false_type{});
      for (; s < n_steps + 2; ++s) ball_iteration(s, std::true_type{});
      for (; s < n_iter; ++s) ball_iteration(s, std::false_type{});
    }
    WS2_STAMP_STORE();
    {                                                      // last observation, ball half
      float* const d = o.obs + i0 * S2D_OBS_DIM;
      *reinterpret_cast<float4*>(d + 4) = make_float4(ob6[0][4], ob6[0][5], ob6[0][6], ob6[0][7]);
      *reinterpret_cast<float2*>(d + 8) = make_float2(ob6[0][8], ob6[0][9]);
      *reinterpret_cast<float2*>(d + 14) = make_float2(ob6[1][4], ob6[1][5]);
      *reinterpret_cast<float4*>(d + 16) = make_float4(ob6[1][6], ob6[1][7], ob6[1][8], ob6[1][9]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// host side (called by s2d_rollout in s2d_engine.hip; hidden symbol of the same library)
// ------------------------------------------------------------------------------------------
static bool aligned_to(const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// Launches the two-envs-per-lane pipeline when the batch and the record allow it and writes the instantiation's name to `name`
// (<= 95 characters).  Returns 1 when it launched (the caller checks hipGetLastError), 0 when the launch is not eligible.
extern "C" int s2d_internal_rollout2(int mode, int noise, const S2DHot* hot, const S2DRare* rare_dev, float* S, int64_t stride,
                                     int64_t n, int n_steps, const void* actions_dev, int kind, const RolloutOut* ro,
                                     const StepOut* o, void* stream, char* name) {
  if (n <= 0 || n % kGroup != 0) return 0;
  if (!ro->obs || !ro->action || !ro->reward || !ro->done || !ro->result) return 0;
  if (!aligned_to(ro->obs, 16) || !aligned_to(ro->reward, 8) || !aligned_to(ro->done, 2) || !aligned_to(ro->result, 2) ||
      !aligned_to(ro->action, mode == S2D_MODE_TURN4 ? 16 : 8))
    return 0;
  if (kind != S2D_ACT_RANDOM && !aligned_to(actions_dev, kind == S2D_ACT_DISCRETE_I64 || kind == S2D_ACT_TURNING ? 16 : 8)) return 0;
  using RollK = void (*)(S2DHot, const S2DRare*, float*, int64_t, int64_t, int, const void*, int, RolloutOut, StepOut);
  static const RollK table[3][2][2] = {
      {{s2d_reach_rollout_ws2_kernel<S2D_MODE_DISCRETE, false, false>, s2d_reach_rollout_ws2_kernel<S2D_MODE_DISCRETE, false, true>},
       {s2d_reach_rollout_ws2_kernel<S2D_MODE_DISCRETE, true, false>, s2d_reach_rollout_ws2_kernel<S2D_MODE_DISCRETE, true, true>}},
      {{s2d_reach_rollout_ws2_kernel<S2D_MODE_CONT1, false, false>, s2d_reach_rollout_ws2_kernel<S2D_MODE_CONT1, false, true>},
       {s2d_reach_rollout_ws2_kernel<S2D_MODE_CONT1, true, false>, s2d_reach_rollout_ws2_kernel<S2D_MODE_CONT1, true, true>}},
      {{s2d_reach_rollout_ws2_kernel<S2D_MODE_TURN4, false, false>, s2d_reach_rollout_ws2_kernel<S2D_MODE_TURN4, false, true>},
       {s2d_reach_rollout_ws2_kernel<S2D_MODE_TURN4, true, false>, s2d_reach_rollout_ws2_kernel<S2D_MODE_TURN4, true, true>}}};
  const int nt = ro->nt ? 1 : 0;
  hipLaunchKernelGGL(table[mode][noise ? 1 : 0][nt], dim3((unsigned)(n / kGroup)), dim3(kWsBlock), 0, static_cast<hipStream_t>(stream),
                     *hot, rare_dev, S, stride, n, n_steps, actions_dev, kind, *ro, *o);
  static const char* const mode_names[3] = {"discrete", "continuous", "turning"};
  if (name) std::snprintf(name, 96, "s2d_reach_rollout_ws2_kernel<%s,noise=%d,nt=%d>", mode_names[mode], noise ? 1 : 0, nt);
  return 1;
}
