// ws_roles.hip -- how many role waves per 64-env group?  Same skeleton as ws_skeleton.hip (LDS hand-off read -> dependent fma
// chain -> hand-off write -> record stores -> one s_barrier per cycle), with R = 3 .. 8 waves per group sharing a fixed total of
// chain instructions, each extra wave paying its own loop / hand-off overhead.  1024 groups, T = 64.
//   hipcc -O3 --offload-arch=gfx950 -o ws_roles ws_roles.hip && ./ws_roles
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Cfg { int T, R, c[8], prio[8], store_role_obs, store_role_small; };
struct Out { float* obs; int* action; float* reward; uint8_t* done; uint8_t* result; };

__device__ __forceinline__ float chain(float a, int n, float m, float c) {
  for (int k = 0; k < n; k += 8)
    asm volatile("v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\n"
                 "v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(c));
  return a;
}

__global__ __launch_bounds__(512) void skel(Cfg g, Out o, int64_t n, float seed) {
  __shared__ float hand[2][8][2][64];
  __shared__ __attribute__((aligned(16))) float tile[2][640];
  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t first = (int64_t)blockIdx.x * 64, i = first + lane;
  if (g.prio[role] == 1) __builtin_amdgcn_s_setprio(1); else if (g.prio[role] == 2) __builtin_amdgcn_s_setprio(2);
  float a = seed + lane, m = 1.0001f, c = 0.5f;
  const int iters = g.T + g.R - 1;
  const int prev = role == 0 ? g.R - 1 : role - 1;
  int64_t row = 0;
  for (int s = 0; s < iters; ++s) {
    const int b = s & 1;
    a += hand[b ^ 1][prev][0][lane] + hand[b ^ 1][prev][1][lane];
    a = chain(a, g.c[role], m, c);
    hand[b][role][0][lane] = a; hand[b][role][1][lane] = a + 1.0f;
    if (role == g.store_role_small) {
      o.action[row + i] = (int)a; o.reward[row + i] = a; o.done[row + i] = (uint8_t)(a > 3.0f); o.result[row + i] = (uint8_t)(a > 5.0f);
      float* t = &tile[b][lane * 10]; t[0] = a; t[1] = a; t[2] = a; t[3] = a;
    }
    if (role == g.store_role_obs) {
      float* t = &tile[b][lane * 10]; t[4] = a; t[5] = a; t[6] = a; t[7] = a; t[8] = a; t[9] = a;
      const float4* t4 = reinterpret_cast<const float4*>(tile[b ^ 1]);
      float4* d4 = reinterpret_cast<float4*>(o.obs + (row + first) * 10);
      d4[lane] = t4[lane]; d4[64 + lane] = t4[64 + lane];
      if (lane < 32) d4[128 + lane] = t4[128 + lane];
    }
    if (s < g.T - 1) row += n;
    __syncthreads();
  }
  if (a == 12345.678f) o.reward[i] = a;
}

int main() {
  const int64_t n = 65536; const int T = 64;
  Out o;
  CHK(hipMalloc(&o.obs, (size_t)T * n * 40)); CHK(hipMalloc(&o.action, (size_t)T * n * 4)); CHK(hipMalloc(&o.reward, (size_t)T * n * 4));
  CHK(hipMalloc(&o.done, (size_t)T * n)); CHK(hipMalloc(&o.result, (size_t)T * n));
  struct Row { const char* name; Cfg g; };
  std::vector<Row> rows = {
      {"4 waves 40/104/96/88 prio 0/1/2/2", {T, 4, {40, 104, 96, 88}, {0, 1, 2, 2}, 3, 2}},
      {"4 waves 40/104/96/88 prio 0/0/0/0", {T, 4, {40, 104, 96, 88}, {0, 0, 0, 0}, 3, 2}},
      {"3 waves 144/96/88 prio 1/2/2", {T, 3, {144, 96, 88}, {1, 2, 2}, 2, 1}},
      {"5 waves 40/56/48/96/88", {T, 5, {40, 56, 48, 96, 88}, {0, 1, 1, 2, 2}, 4, 3}},
      {"6 waves 40/104/48/48/48/40", {T, 6, {40, 104, 48, 48, 48, 40}, {0, 2, 1, 1, 1, 1}, 5, 3}},
      {"6 waves 40/56/48/64/64/56", {T, 6, {40, 56, 48, 64, 64, 56}, {0, 1, 1, 1, 1, 1}, 5, 3}},
      {"8 waves 40/56/48/48/48/48/40/0", {T, 8, {40, 56, 48, 48, 48, 48, 40, 0}, {0, 1, 1, 1, 1, 1, 1, 1}, 6, 4}},
      {"8 waves 48/48/48/40/40/40/40/24 (balanced)", {T, 8, {48, 48, 48, 40, 40, 40, 40, 24}, {0, 0, 0, 0, 0, 0, 0, 0}, 7, 4}},
      {"2 waves 144/184", {T, 2, {144, 184}, {0, 0}, 1, 1}},
  };
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int k = 0; k < 4000; ++k) hipLaunchKernelGGL(skel, dim3(1024), dim3(256), 0, 0, rows[0].g, o, n, 1.0f);
  CHK(hipDeviceSynchronize());
  printf("%-48s %10s %12s\n", "roles (chain instructions per wave), 65536 x 64", "us/launch", "ns/cycle");
  for (auto& r : rows) {
    const dim3 blk(64 * r.g.R);
    for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(skel, dim3(1024), blk, 0, 0, r.g, o, n, 1.0f);
    CHK(hipDeviceSynchronize());
    const int reps = 400;
    CHK(hipEventRecord(e0));
    for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(skel, dim3(1024), blk, 0, 0, r.g, o, n, 1.0f);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("%-48s %10.2f %12.1f\n", r.name, us, us * 1e3 / (T + r.g.R - 1));
    fflush(stdout);
  }
  return 0;
}
