// s2d_kernels.h -- device-side helpers shared by the kernels of s2d_engine.hip and s2d_rollout2.hip: launch constants, the
// per-step output block, record stores, the LDS observation tile, episode counters, action decoding and the prepared-episode
// slots of the wave-specialised rollout pipelines.  (Moved out of s2d_engine.hip in round 4, unchanged.)
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "s2d_device.h"

#define S2D_API extern "C" __attribute__((visibility("default")))

#ifndef S2D_BLOCK
#define S2D_BLOCK 256
#endif
static constexpr int kBlock = S2D_BLOCK;
static constexpr int kWave = 64;
static constexpr int kWavesPerBlock = kBlock / kWave;
static constexpr int kObsTile = kWave * S2D_OBS_DIM;  // 640 floats per wave
static constexpr int64_t kWsMaxEnvs = 524288;         // the wave-specialised rollout wins or ties up to here at steady clocks (profiles/r01/ws_vs_unified_sweep.txt)

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
struct StepOut {
  float* obs;            // [N][10]
  float* reward;         // [N]
  uint8_t* done;         // [N]
  uint8_t* result;       // [N]
  float* terminal_obs;   // [N][10]
  float* action_dir;     // [N]
  uint8_t* action_cmd;   // [N]
  unsigned long long* stats;
  float* prep;           // persistent prepared episodes of the per-step API: [2][PS_WORDS][stride] words + [2][stride] tags
};

// ---- persistent prepared episodes (per-step API) ------------------------------------------------------------------------
// s2d_step runs one wave per 64 envs, and a launch lasts as long as its slowest wave.  Drawing a reset inline (Philox blocks,
// the rejection loop, a simulator cycle, the first observation: ~2.8 us with one or two active lanes) therefore cost EVERY
// launch those 2.8 us, because some wave always has an episode ending (profiles/r02/ab_step.txt: 7.3 us per launch against
// 4.5 us for a workload whose episodes never end).  Episode j of env g is a function of (g, j) alone, so every env keeps its
// next two episodes prepared in the arena: slot j & 1 holds episode j, tagged with j.  A slot is the seven words of the
// post-reset state that depend on the draw (player x, y, body; ball x, y, vx, vy after the reset's command-less cycle); the
// rest is constant (player at rest, stamina model one cycle after a recover) or a function of those seven (first
// observation, reward carry) and is rebuilt in the few waves that reset -- every extra load of a step costs all waves
// ~15 ns (profiles/r02/ab_step.txt), a rebuilt word only the resetting ones.  A step loads both slots with the state (no
// load waits for another one's result); a reset is a register copy plus ~0.4 us of arithmetic.  Slots are refilled off the critical path by extra
// workgroups appended to the same launch's grid: each looks at its envs' `episode` e and prepares episode e + 2 if slot
// e & 1 does not hold it yet -- never the slot a main wave may be reading in the same launch -- and finishes well inside
// the launch.  s2d_reset prepares both slots of the envs it resets; a slot whose tag does not match (first use after the
// rollout kernels advanced the episode counter) is ignored and the reset is drawn inline, once.
enum { PS_PX, PS_PY, PS_BODY, PS_BX, PS_BY, PS_BVX, PS_BVY, PS_WORDS };
S2D_DEV uint32_t* prep_tags(float* prep, int64_t stride) { return reinterpret_cast<uint32_t*>(prep + 2 * PS_WORDS * stride); }
// A tag = the episode index the slot holds (bits 0..29) + the SIGNS of the player's post-reset velocity (bits 31, 30): the player of a
// fresh episode is at rest, but a collision in the reset's cycle multiplies its +0 velocity by collision_vel_rate < 0 and leaves -0
// (54 of 650 000 resets of the stock task) -- a state word the CPU checker holds bit for bit.  (Round 3 restored +0 always.)
static constexpr uint32_t kPrepTagMask = 0x3fffffffu;
S2D_DEV uint32_t prep_tag_of(uint32_t episode, const NextEpisode& q) {
  return (episode & kPrepTagMask) | ((uint32_t)__float_as_int(q.vx) & 0x80000000u) | (((uint32_t)__float_as_int(q.vy) & 0x80000000u) >> 1);
}

// LDS ops of one wave execute in order, so a wave-private tile needs no s_barrier; the
// wavefront-scope fences only stop the compiler from reordering the cross-lane accesses.
S2D_DEV void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Transpose a wave's [64][10] observation block through LDS and store it as one contiguous
// run.  `dst` = address of the block's first row (wave-uniform); `valid` = number of floats of
// the run that exist (640, or fewer in the last wave).  The two halves may run in different
// waves (tile_write by the observing wave, tile_flush by another one after an s_barrier).
S2D_DEV void tile_write(float* tile, const ObsOut& ob, int lane, bool active) {
  if (active) {
#pragma unroll
    for (int k = 0; k < S2D_OBS_DIM; ++k) tile[lane * S2D_OBS_DIM + k] = ob.o[k];
  }
}
// Record stores.  A launch whose record does not fit the 256 MiB Infinity Cache streams it out: non-temporal stores (`nt`, a
// wave-uniform flag the host sets from the record's size) keep those lines from being parked in L2 on their way -- 4 % on the
// 838 MB record of the headline configuration, steadier from region to region (profiles/r03/ab_store_policy_long.txt); a record
// that fits (64 cycles x 65 536 envs = 218 MB) keeps the plain stores, which are 3 % faster there.
typedef float v4f32_t __attribute__((ext_vector_type(4)));
S2D_DEV void rec_store16(float4* p, const float4& v, bool nt) {
  if (nt) {
    const v4f32_t w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(w) : "memory");
  } else {
    *p = v;
  }
}
template <typename T>
S2D_DEV void rec_store(T* p, T v, bool nt) {
  if (nt) __builtin_nontemporal_store(v, p); else *p = v;
}
// all_vec: the caller has established once that every row of its launch is a full, 16-byte-aligned tile (no per-call test)
S2D_DEV void tile_flush(const float* tile, int lane, float* __restrict__ dst, int valid, bool nt = false, bool all_vec = false) {
  const bool vec = all_vec || ((valid == kObsTile) && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0));
  if (vec) {
    const float4* t4 = reinterpret_cast<const float4*>(tile);
    float4* d4 = reinterpret_cast<float4*>(dst);
    rec_store16(d4 + lane, t4[lane], nt);
    rec_store16(d4 + kWave + lane, t4[kWave + lane], nt);
    if (lane < 32) rec_store16(d4 + 2 * kWave + lane, t4[2 * kWave + lane], nt);
  } else {
#pragma unroll
    for (int j = 0; j < S2D_OBS_DIM; ++j) {
      int idx = j * kWave + lane;
      if (idx < valid) dst[idx] = tile[idx];
    }
  }
}
S2D_DEV void store_obs_tile(float* tile, const ObsOut& ob, int lane, bool active, float* __restrict__ dst,
                            int valid) {
  tile_write(tile, ob, lane, active);
  wave_lds_fence();
  tile_flush(tile, lane, dst, valid);
  wave_lds_fence();
}

// Episode counters: every group of 64 envs owns one row of stats[S2D_STATS_ROWS(n)][8], and in every kernel that group is
// one wave (or the agent wave of one workgroup), so the row is updated with a plain load at the start of the launch and a plain
// store at its end -- lanes 0 .. 3 hold counters 0 .. 3; the env-step counter of the whole batch is kept by the first group, so a
// wave in which no episode ended stores nothing.  (Round 1 used striped atomics: no contention to speak of, yet the
// three atomics at the end of a wave cost s2d_step 0.3 us per launch, profiles/r02/ab_step_ablation.txt.)
S2D_DEV unsigned long long* stats_row(unsigned long long* stats, int64_t wave_first) { return stats + (wave_first / kWave) * 8; }
S2D_DEV unsigned long long stats_load(const unsigned long long* row, int lane) { return lane < 4 ? row[lane] : 0ull; }
// steps / goal / out / timeout: wave-uniform increments
S2D_DEV void stats_store(unsigned long long* row, int lane, unsigned long long old, unsigned long long steps, unsigned int goal,
                         unsigned int out, unsigned int timeout) {
  const unsigned long long add = lane == 0 ? steps : lane == 1 ? goal : lane == 2 ? out : timeout;
  if (lane < 4 && add != 0ull) row[lane] = old + add;
}
S2D_DEV unsigned int wave_count(bool pred) { return (unsigned int)__popcll(__ballot(pred)); }

// caller-provided action of env i at rollout step t (layouts of include/s2d.h)
template <int MODE>
S2D_DEV Action4 load_action(const void* __restrict__ actions, int kind, int64_t idx) {
  Action4 a{0.0f, 0.0f, 0.0f, 0.0f};
  if (MODE == S2D_MODE_DISCRETE) {
    a.a0 = (kind == S2D_ACT_DISCRETE_I64) ? (float)static_cast<const long long*>(actions)[idx]
                                          : (float)static_cast<const int32_t*>(actions)[idx];
  } else if (MODE == S2D_MODE_CONT1) {
    a.a0 = static_cast<const float*>(actions)[idx];
  } else {
    float4 v = static_cast<const float4*>(actions)[idx];
    a.a0 = v.x; a.a1 = v.y; a.a2 = v.z; a.a3 = v.w;
  }
  return a;
}
// in-kernel uniform random policy at policy step k (s2d_device.h: policy_quad).  `quad` caches
// the POLICY block of counter k >> 2; `refresh` = it has to be drawn now.
template <int MODE>
S2D_DEV Action4 random_action(const S2DHot& p, uint32_t gid_lo, uint32_t gid_hi, uint32_t k, U4& quad, bool refresh) {
  Action4 a{0.0f, 0.0f, 0.0f, 0.0f};
  if (MODE == S2D_MODE_TURN4) {
    U4 w = s2d_draw(p, gid_lo, gid_hi, k, S2D_ST_POLICY, 1);
    a.a0 = rnd_u01(w.x) * 2.0f - 1.0f; a.a1 = rnd_u01(w.y) * 2.0f - 1.0f;
    a.a2 = rnd_u01(w.z) * 2.0f - 1.0f; a.a3 = rnd_u01(w.w) * 2.0f - 1.0f;
  } else {
    if (refresh) quad = policy_quad(p, gid_lo, gid_hi, k, S2D_ST_POLICY);
    uint32_t w = quad_word(quad, k);
    if (MODE == S2D_MODE_DISCRETE) a.a0 = (float)rnd_below(w, (uint32_t)p.n_actions);
    else a.a0 = rnd_u01(w) * 2.0f - 1.0f;
  }
  return a;
}
// does a launch of this mode / action kind / noise setting consume the env's policy_step?  (wave-uniform)
template <int MODE, bool NOISE>
S2D_DEV bool uses_policy_step(int kind) { return kind == S2D_ACT_RANDOM || MODE == S2D_MODE_TURN4 || NOISE; }

template <int MODE>
S2D_DEV void store_rollout_action(void* __restrict__ dst, int64_t idx, const Action4& a) {
  if (MODE == S2D_MODE_DISCRETE) static_cast<int32_t*>(dst)[idx] = (int32_t)a.a0;
  else if (MODE == S2D_MODE_CONT1) static_cast<float*>(dst)[idx] = a.a0;
  else static_cast<float4*>(dst)[idx] = make_float4(a.a0, a.a1, a.a2, a.a3);
}

// action of one env for the step at policy step k -> decoded command (A2, reach_ball_env.py:53-85)
// with the command-only part of the dash already evaluated.  `quad` / `squad` cache the POLICY /
// SELECT blocks across the four steps they serve.
template <int MODE>
S2D_DEV CmdPrep decide(const S2DHot& p, const void* __restrict__ actions, int kind, int64_t idx, uint32_t gid_lo,
                       uint32_t gid_hi, uint32_t k, bool refresh, U4& quad, U4& squad, void* __restrict__ action_out,
                       int& cmd, float& dir) {
  if (kind == S2D_ACT_COMMAND) {                           // a decoded body command, executed as it is (wave-uniform branch)
    const float4 v = static_cast<const float4*>(actions)[idx];
    cmd = command_code(v.x); dir = v.z;
    return cmd_prepare(p, cmd, v.y, v.z);
  }
  Action4 a = (kind == S2D_ACT_RANDOM) ? random_action<MODE>(p, gid_lo, gid_hi, k, quad, refresh)
                                       : load_action<MODE>(actions, kind, idx);
  if (action_out) store_rollout_action<MODE>(action_out, idx, a);
  float u = 0.0f;
  if (MODE == S2D_MODE_TURN4) {                          // reach_ball_env.py:71
    if (refresh) squad = policy_quad(p, gid_lo, gid_hi, k, S2D_ST_SELECT);
    u = rnd_u01(quad_word(squad, k));
  }
  float power;
  action_map<MODE>(p, a, u, cmd, power, dir);
  return cmd_prepare(p, cmd, power, dir);
}

// action -> decoded command with the command-only half of the dash (decide() without the load and the store: for callers that
// keep the action, e.g. to park it for a storing wave)
template <int MODE>
S2D_DEV CmdPrep decode_action(const S2DHot& p, const Action4& a, uint32_t gl, uint32_t gh, uint32_t k, bool refresh, U4& squad,
                              int& cmd, float& dir) {
  float u = 0.0f;
  if (MODE == S2D_MODE_TURN4) {                            // reach_ball_env.py:71
    if (refresh) squad = policy_quad(p, gl, gh, k, S2D_ST_SELECT);
    u = rnd_u01(quad_word(squad, k));
  }
  float power;
  action_map<MODE>(p, a, u, cmd, power, dir);
  return cmd_prepare(p, cmd, power, dir);
}

// Prepared reset samples of one wave (LDS, struct-of-arrays over the 64 lanes).  The sample of
// an env's NEXT episode depends only on (gid, cycle at which the current episode began), so a
// rollout kernel draws them for many lanes at once -- a full wave at launch, then whenever
// kRefillMin lanes have used theirs -- instead of running the Philox + rejection loop with one
// or two active lanes each time an episode ends.
// Without noise the tile holds the whole post-reset state (NextEpisode, 13 words), with noise the sample (7).
struct PrepTile { float v[13 + S2D_OBS_DIM + 2][kWave]; };   // NextEpisode + FirstObs
#ifndef S2D_REFILL_MIN
#define S2D_REFILL_MIN 8
#endif
static constexpr int kRefillMin = S2D_REFILL_MIN;

template <bool NOISE>
S2D_DEV void prep_fill(const S2DHot& p, const S2DRare* __restrict__ rp, PrepTile& t, int lane, const Env& e,
                       uint32_t gid_lo, uint32_t gid_hi) {
  const S2DRare r = *rp;
  const NextEpisode q = episode_prepare<NOISE>(p, rp, r, gid_lo, gid_hi, reset_key(e));
  t.v[0][lane] = q.px; t.v[1][lane] = q.py; t.v[2][lane] = q.vx; t.v[3][lane] = q.vy; t.v[4][lane] = q.body;
  t.v[5][lane] = q.stamina; t.v[6][lane] = q.effort; t.v[7][lane] = q.recovery; t.v[8][lane] = q.capacity;
  t.v[9][lane] = q.bx; t.v[10][lane] = q.by; t.v[11][lane] = q.bvx; t.v[12][lane] = q.bvy;
  const FirstObs f = first_obs(p, q);
#pragma unroll
  for (int k = 0; k < S2D_OBS_DIM; ++k) t.v[13 + k][lane] = f.o[k];
  t.v[13 + S2D_OBS_DIM][lane] = f.dist; t.v[14 + S2D_OBS_DIM][lane] = f.rel;
}
// the same by the whole wave together (reset_sample_coop; wave-uniform call, `need` = this lane's tile entry is to be drawn;
// `scratch` = 64 wave-private LDS words)
template <bool NOISE>
S2D_DEV void prep_fill_coop(const S2DHot& p, const S2DRare* __restrict__ rp, PrepTile& t, int lane, uint32_t key,
                            uint32_t gid_lo, uint32_t gid_hi, bool need, uint32_t* scratch) {
  const S2DRare r = *rp;
  const NextEpisode q = episode_prepare_coop<NOISE>(p, rp, r, gid_lo, gid_hi, key, need, lane, scratch);
  const FirstObs f = first_obs(p, q);
  if (need) {
    t.v[0][lane] = q.px; t.v[1][lane] = q.py; t.v[2][lane] = q.vx; t.v[3][lane] = q.vy; t.v[4][lane] = q.body;
    t.v[5][lane] = q.stamina; t.v[6][lane] = q.effort; t.v[7][lane] = q.recovery; t.v[8][lane] = q.capacity;
    t.v[9][lane] = q.bx; t.v[10][lane] = q.by; t.v[11][lane] = q.bvx; t.v[12][lane] = q.bvy;
#pragma unroll
    for (int k = 0; k < S2D_OBS_DIM; ++k) t.v[13 + k][lane] = f.o[k];
    t.v[13 + S2D_OBS_DIM][lane] = f.dist; t.v[14 + S2D_OBS_DIM][lane] = f.rel;
  }
}
S2D_DEV NextEpisode prep_take_episode(const PrepTile& t, int lane) {
  return NextEpisode{t.v[0][lane], t.v[1][lane], t.v[2][lane], t.v[3][lane], t.v[4][lane], t.v[5][lane], t.v[6][lane],
                     t.v[7][lane], t.v[8][lane], t.v[9][lane], t.v[10][lane], t.v[11][lane], t.v[12][lane]};
}

S2D_DEV const S2DTables* tables_of(const S2DRare* rp) {
  return reinterpret_cast<const S2DTables*>(reinterpret_cast<const char*>(rp) + 256);
}

// A1: one Soccer2DEnv.step (soccer_2d_env.py:226-269) for the env held in registers, given the
// decoded command.  Returns the observation to hand back (post auto-reset), reward/done/result.
// prep == nullptr: the reset sample is drawn on the spot (per-step API).
// FAST: the dash-only fast path (s2d_device.h, S2DTables): ep_lds = effort * power by step number, sc_lut = (sin, cos) of the
// whole degrees -180 .. 180; the caller has checked that the env sits on the table.
template <bool NOISE, bool FAST = false>
S2D_DEV void step_env(const S2DHot& p, const S2DRare* __restrict__ rp, Env& e, uint32_t gid_lo, uint32_t gid_hi,
                      uint32_t k, int cmd, const CmdPrep& c, ObsOut& ob, float& reward, int& done, int& result,
                      float* __restrict__ terminal_row, PrepTile* prep, int lane, bool& have_prep,
                      const float* ep_lds = nullptr, const float2* sc_lut = nullptr) {
  NoiseIn nz{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  if (NOISE) nz = noise_prepare(p, gid_lo, gid_hi, k, S2D_ST_NOISE, cmd == S2D_CMD_TURN);
  float d2;
  if constexpr (FAST) {
    const float ep = ep_lds[e.step_number];
    const float2 sc = sc_lut[(int)norm_deg(e.body + c.dir) + 180];
    e.step_number += 1;                                  // reach_ball_env.py:55
    d2 = sim_cycle_dash_fast<NOISE>(p, rp, e, ep, c.dir_rate, sc.x, sc.y, nz);
  } else {
    e.step_number += 1;                                  // reach_ball_env.py:55
    d2 = sim_cycle<NOISE, true>(p, rp, e, cmd, c, nz);   // trainer forces PlayOn each cycle (:242)
  }
  observe_and_check(p, e, d2, ob, done, reward, result);
  if (done && p.auto_reset) {                  // SB3 VecEnv convention
#pragma unroll
    for (int k = 0; k < S2D_OBS_DIM; ++k) terminal_row[k] = ob.o[k];
    if (prep) {
      if (!have_prep) prep_fill<NOISE>(p, rp, *prep, lane, e, gid_lo, gid_hi);   // episode shorter than the refill cadence
      have_prep = false;
      episode_begin(e, prep_take_episode(*prep, lane));    // state, first observation and carry were prepared together
#pragma unroll
      for (int k = 0; k < S2D_OBS_DIM; ++k) ob.o[k] = prep->v[13 + k][lane];
      e.prev_dist = prep->v[13 + S2D_OBS_DIM][lane]; e.prev_angle = prep->v[14 + S2D_OBS_DIM][lane];
      return;
    } else {
      d2 = env_reset<NOISE>(p, rp, e, gid_lo, gid_hi);
    }
    int dn2, r2; float w2;
    observe_and_check(p, e, d2, ob, dn2, w2, r2);        // reach_ball_env.py:166: carry seeded, outputs dropped
  }
}

// prepare episode `episode` of the env and store it in its persistent slot (episode & 1), tag last
template <bool NOISE>
S2D_DEV void prep_store(const S2DHot& p, const S2DRare* __restrict__ rp, float* __restrict__ prep, int64_t stride, int64_t i,
                        uint32_t gl, uint32_t gh, uint32_t episode) {
  const S2DRare r = *rp;
  const NextEpisode q = episode_prepare<NOISE>(p, rp, r, gl, gh, episode);
  float* d = prep + (int64_t)(episode & 1u) * PS_WORDS * stride + i;
  const float w[PS_WORDS] = {q.px, q.py, q.body, q.bx, q.by, q.bvx, q.bvy};
#pragma unroll
  for (int k = 0; k < PS_WORDS; ++k) d[k * stride] = w[k];
  prep_tags(prep, stride)[(int64_t)(episode & 1u) * stride + i] = prep_tag_of(episode, q);
}
// the refill workgroups' form: the whole wave draws together (reset_sample_coop; `need` = this lane's slot is to be drawn)
template <bool NOISE>
S2D_DEV void prep_store_coop(const S2DHot& p, const S2DRare* __restrict__ rp, float* __restrict__ prep, int64_t stride, int64_t i,
                             uint32_t gl, uint32_t gh, uint32_t episode, bool need, int lane, uint32_t* scratch) {
  const S2DRare r = *rp;
  const NextEpisode q = episode_prepare_coop<NOISE>(p, rp, r, gl, gh, episode, need, lane, scratch);
  if (need) {
    float* d = prep + (int64_t)(episode & 1u) * PS_WORDS * stride + i;
    const float w[PS_WORDS] = {q.px, q.py, q.body, q.bx, q.by, q.bvx, q.bvy};
#pragma unroll
    for (int k = 0; k < PS_WORDS; ++k) d[k * stride] = w[k];
    prep_tags(prep, stride)[(int64_t)(episode & 1u) * stride + i] = prep_tag_of(episode, q);
  }
}
// the post-reset state from a slot: the drawn words + what every reset leaves behind (reset_apply: player at rest -- its
// velocity stays +0 through the command-less cycle, with noise on too: the noise magnitude is proportional to the speed --
// and the stamina model one update after a recover)
S2D_DEV NextEpisode prep_episode(const S2DHot& p, const S2DRare* __restrict__ rp, const float* w, uint32_t tag) {
  Env t{};
  t.stamina = p.stamina_max; t.recovery = rp->recover_init; t.effort = p.effort_init; t.capacity = p.stamina_capacity;
  update_stamina(p, t);
  return NextEpisode{w[PS_PX], w[PS_PY], __int_as_float((int)(tag & 0x80000000u)), __int_as_float((int)((tag << 1) & 0x80000000u)), w[PS_BODY],
                     t.stamina, t.effort, t.recovery, t.capacity,
                     w[PS_BX], w[PS_BY], w[PS_BVX], w[PS_BVY]};
}


// T fused cycles per launch: the 17 state words stay in registers, only the rollout record
// (obs 40 B + action 4 B + reward 4 B + done 1 B + result 1 B per env-step) streams out.
static constexpr int64_t kInfinityCacheBytes = 256ll << 20;   // MI355X
struct RolloutOut {
  float* obs; void* action; float* reward; uint8_t* done; uint8_t* result;
  int nt;                                                  // the record is larger than the Infinity Cache: stream it (rec_store)
};

enum { WS_PX, WS_PY, WS_BODY, WS_BX, WS_BY, WS_BVX, WS_BVY, WS_FLAGS, WS_WORDS };
enum { WA_CMD, WA_POWER, WA_DIR, WA_RATE, WA_NPM, WA_NPS, WA_NPC, WA_NBM, WA_NBS, WA_NBC, WA_NTU, WA_WORDS };   // command + prepared noise
// A prepared episode in LDS: the nine words of the post-reset state that depend on the draw (player x, y, vx, vy, body; ball x, y,
// vx, vy -- the player's velocity is +0 or, after a collision in the reset's cycle, -0) + the first observation and the reward carry.
// The four stamina words a reset leaves behind are the same for every episode (recover, then one update_stamina): reset_stamina().
enum { SL_BX = 5, SL_FIRST = 9, SL_DIST = SL_FIRST + S2D_OBS_DIM, SL_REL, SL_WORDS };
static constexpr int kSlots = 3;
static constexpr int kWsBlock = 4 * kWave;

struct ResetStamina { float stamina, effort, recovery, capacity; };
S2D_DEV ResetStamina reset_stamina(const S2DHot& p, const S2DRare* __restrict__ rp) {   // reset_apply()'s stamina block, one cycle later
  Env t{};
  t.stamina = p.stamina_max; t.recovery = rp->recover_init; t.effort = p.effort_init; t.capacity = p.stamina_capacity;
  update_stamina(p, t);
  return ResetStamina{t.stamina, t.effort, t.recovery, t.capacity};
}
// slot rows are `W` floats wide (64 envs per workgroup, or 128); col = this env's column
template <int W>
S2D_DEV void slot_put(float (*slot)[W], int col, const NextEpisode& q, const FirstObs& f) {
  slot[0][col] = q.px; slot[1][col] = q.py; slot[2][col] = q.vx; slot[3][col] = q.vy; slot[4][col] = q.body;
  slot[SL_BX][col] = q.bx; slot[SL_BX + 1][col] = q.by; slot[SL_BX + 2][col] = q.bvx; slot[SL_BX + 3][col] = q.bvy;
#pragma unroll
  for (int k = 0; k < S2D_OBS_DIM; ++k) slot[SL_FIRST + k][col] = f.o[k];
  slot[SL_DIST][col] = f.dist; slot[SL_REL][col] = f.rel;
}
template <int W>
S2D_DEV NextEpisode slot_take(const float (*slot)[W], int col, const ResetStamina& st) {
  return NextEpisode{slot[0][col], slot[1][col], slot[2][col], slot[3][col], slot[4][col], st.stamina, st.effort, st.recovery, st.capacity,
                     slot[SL_BX][col], slot[SL_BX + 1][col], slot[SL_BX + 2][col], slot[SL_BX + 3][col]};
}
// one prepared episode of this lane's env -> LDS slot
template <bool NOISE>
S2D_DEV void slot_fill(const S2DHot& p, const S2DRare* __restrict__ rp, float (*slot)[kWave], int lane, uint32_t gl,
                       uint32_t gh, uint32_t episode) {
  const S2DRare r = *rp;
  const NextEpisode q = episode_prepare<NOISE>(p, rp, r, gl, gh, episode);
  slot_put<kWave>(slot, lane, q, first_obs(p, q));
}
// the prologue's form: the whole wave draws together (reset_sample_coop; `need` = this lane has an env)
template <bool NOISE>
S2D_DEV void slot_fill_coop(const S2DHot& p, const S2DRare* __restrict__ rp, float (*slot)[kWave], int lane, uint32_t gl,
                            uint32_t gh, uint32_t episode, bool need, uint32_t* scratch) {
  const S2DRare r = *rp;
  const NextEpisode q = episode_prepare_coop<NOISE>(p, rp, r, gl, gh, episode, need, lane, scratch);
  const FirstObs f = first_obs(p, q);
  if (need) slot_put<kWave>(slot, lane, q, f);
}
