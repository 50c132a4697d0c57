// ws_skeleton.hip -- what does the STRUCTURE of the four-wave rollout pipeline cost, apart from its arithmetic?
// A workgroup = 4 role waves (policy | simulate | agent | ball) of one 64-env group, T + 3 iterations; per iteration each
// wave does U "steps" of: LDS hand-off read -> a dependent chain of c_role fma instructions -> LDS hand-off write ->
// its share of the rollout-record stores (same bytes and addresses as the engine: obs 2560 B / action 256 B / reward 256 B /
// done 64 B / result 64 B per group-step), then ONE s_barrier.  Sweeping chain lengths, U and the switches tells how the
// per-iteration time splits into barrier + LDS latency, single-wave issue rate of the longest chain, and store traffic.
//   hipcc -O3 --offload-arch=gfx950 -o ws_skeleton ws_skeleton.hip && ./ws_skeleton
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Cfg { int T, U, c[4], lds, stores, prio; };
struct Out { float* obs; int* action; float* reward; uint8_t* done; uint8_t* result; };

__device__ __forceinline__ float chain(float a, int n, float m, float c) {
  for (int k = 0; k < n; k += 8)
    asm volatile("v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\n"
                 "v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(c));
  return a;
}

__global__ __launch_bounds__(256) void skel(Cfg g, Out o, int64_t n, float seed) {
  __shared__ float hand[2][4][8][64];            // [buffer][u][word][lane]   (36 KB in all: four groups per CU, as the engine)
  __shared__ __attribute__((aligned(16))) float tile[2][4][640];
  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t first = (int64_t)blockIdx.x * 64, i = first + lane;
  if (g.prio) { if (role == 1) __builtin_amdgcn_s_setprio(2); else if (role >= 2) __builtin_amdgcn_s_setprio(1); }
  float a = seed + lane, m = 1.0001f, c = 0.5f;
  const int iters = g.T / g.U + 3;
  for (int s = 0; s < iters; ++s) {
    const int b = s & 1;
    for (int u = 0; u < g.U; ++u) {
      const int64_t row = ((int64_t)(s % (g.T / g.U)) * g.U + u) * n;
      if (g.lds) a += hand[b ^ 1][u][role][lane] + hand[b ^ 1][u][role + 4][lane];
      a = chain(a, g.c[role], m, c);
      if (g.lds) { hand[b][u][role][lane] = a; hand[b][u][role + 4][lane] = a + 1.0f; }
      if (g.lds && role >= 2) {                   // agent: 4 words, ball: 6 words of the observation tile
        float* t = &tile[b][u][lane * 10];
        if (role == 2) { t[0] = a; t[1] = a; t[2] = a; t[3] = a; } else { t[4] = a; t[5] = a; t[6] = a; t[7] = a; t[8] = a; t[9] = a; }
      }
      if (g.stores) {
        if (role == 0) o.action[row + i] = (int)a;
        if (role == 2) { o.reward[row + i] = a; o.done[row + i] = (uint8_t)(a > 3.0f); o.result[row + i] = (uint8_t)(a > 5.0f); }
        if (role == 3) {
          const float4* t4 = reinterpret_cast<const float4*>(tile[b ^ 1][u]);
          float4* d4 = reinterpret_cast<float4*>(o.obs + (row + first) * 10);
          float4 v0, v1, v2;
          if (g.lds) { v0 = t4[lane]; v1 = t4[64 + lane]; v2 = t4[128 + (lane & 31)]; }
          else { v0 = make_float4(a, a, a, a); v1 = v0; v2 = v0; }
          d4[lane] = v0; d4[64 + lane] = v1;
          if (lane < 32) d4[128 + lane] = v2;
        }
      }
    }
    __syncthreads();
  }
  if (a == 12345.678f) o.reward[i] = a;           // keep the chains alive
}

int main() {
  const int64_t n = 65536;
  const int T = 64;
  Out o;
  CHK(hipMalloc(&o.obs, (size_t)T * n * 40)); CHK(hipMalloc(&o.action, (size_t)T * n * 4)); CHK(hipMalloc(&o.reward, (size_t)T * n * 4));
  CHK(hipMalloc(&o.done, (size_t)T * n)); CHK(hipMalloc(&o.result, (size_t)T * n));
  struct Row { const char* name; Cfg g; };
  std::vector<Row> rows = {
      {"barrier only", {T, 1, {0, 0, 0, 0}, 0, 0, 0}},
      {"+ LDS hand-offs", {T, 1, {0, 0, 0, 0}, 1, 0, 0}},
      {"+ stores (no chains)", {T, 1, {0, 0, 0, 0}, 1, 1, 0}},
      {"stores only (no LDS)", {T, 1, {0, 0, 0, 0}, 0, 1, 0}},
      {"chains 40/180/100/90, no stores", {T, 1, {40, 184, 104, 88}, 1, 0, 0}},
      {"chains 40/180/100/90 + stores", {T, 1, {40, 184, 104, 88}, 1, 1, 0}},
      {"  same, priorities 0/2/1/1", {T, 1, {40, 184, 104, 88}, 1, 1, 1}},
      {"  same, U=2 steps per barrier", {T, 2, {40, 184, 104, 88}, 1, 1, 1}},
      {"  same, U=4 steps per barrier", {T, 4, {40, 184, 104, 88}, 1, 1, 1}},
      {"chains 40/120/100/90 + stores", {T, 1, {40, 120, 104, 88}, 1, 1, 1}},
      {"chains 40/80/64/40 + stores", {T, 1, {40, 80, 64, 40}, 1, 1, 1}},
      {"  same, U=2", {T, 2, {40, 80, 64, 40}, 1, 1, 1}},
      {"  same, U=4", {T, 4, {40, 80, 64, 40}, 1, 1, 1}},
      {"balanced 104/104/104/104 + stores", {T, 1, {104, 104, 104, 104}, 1, 1, 0}},
      {"  same, U=4", {T, 4, {104, 104, 104, 104}, 1, 1, 0}},
      {"balanced 64/64/64/64 + stores", {T, 1, {64, 64, 64, 64}, 1, 1, 0}},
      {"  same, U=4", {T, 4, {64, 64, 64, 64}, 1, 1, 0}},
      {"chains 40/360/100/90 + stores", {T, 1, {40, 360, 104, 88}, 1, 1, 1}},
  };
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  // settle the clock under load
  for (int k = 0; k < 4000; ++k) hipLaunchKernelGGL(skel, dim3(1024), dim3(256), 0, 0, rows[5].g, o, n, 1.0f);
  CHK(hipDeviceSynchronize());
  printf("%-40s %10s %12s %10s\n", "configuration (65536 envs, T = 64)", "us/launch", "ns/env-cycle", "TB/s");
  for (auto& r : rows) {
    for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(skel, dim3(1024), dim3(256), 0, 0, r.g, o, n, 1.0f);
    CHK(hipDeviceSynchronize());
    const int reps = 400;
    CHK(hipEventRecord(e0));
    for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(skel, dim3(1024), dim3(256), 0, 0, r.g, o, n, 1.0f);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("%-40s %10.2f %12.1f %10.2f\n", r.name, us, us * 1e3 / T, r.g.stores ? 50.0 * n * T / us / 1e6 : 0.0);
    fflush(stdout);
  }
  return 0;
}
