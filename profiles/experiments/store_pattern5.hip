// store_pattern5.hip -- the rollout record's store pattern (no arithmetic) with a cache policy PER ARRAY: whole-line streams
// (observations, reward, action) plain or nt, the half-line streams (done, result; 64-env groups) plain or nt.  E = envs per lane.
//   hipcc --offload-arch=gfx950 -O3 -o store_pattern5 store_pattern5.hip && ./store_pattern5 256
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st16(float4* p, float a, bool nt) {
  const v4f w = {a, a, a, a};
  if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(w) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ void st8(void* p, float a, bool nt) {
  const v2f w = {a, a};
  if (nt) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(w) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ void st4(void* p, float a, bool nt) {
  if (nt) asm volatile("global_store_dword %0, %1, off nt" ::"v"(p), "v"(a) : "memory");
  else asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(a) : "memory");
}
__device__ __forceinline__ void st2(void* p, int a, bool nt) {
  if (nt) asm volatile("global_store_short %0, %1, off nt" ::"v"(p), "v"(a) : "memory");
  else asm volatile("global_store_short %0, %1, off" ::"v"(p), "v"(a) : "memory");
}
__device__ __forceinline__ void st1(void* p, int a, bool nt) {
  if (nt) asm volatile("global_store_byte %0, %1, off nt" ::"v"(p), "v"(a) : "memory");
  else asm volatile("global_store_byte %0, %1, off" ::"v"(p), "v"(a) : "memory");
}
// pol bits: 1 obs nt, 2 reward/action nt, 4 done/result nt; flushw: 0 = wave 3 stores the observations, 1 = waves 1 and 3 half each
template <int E>
__global__ __launch_bounds__(256) void k(float* obs, int* act, float* rew, unsigned char* done, unsigned char* res, long n, int T, int pol, int flushw, float* sink) {
  const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
  const long first = (long)blockIdx.x * 64 * E;
  const bool nto = pol & 1, ntr = pol & 2, ntd = pol & 4;
  float acc = (float)lane;
  for (int t = 0; t < T; ++t) {
    const long row = (long)t * n + first;
    float4* d = reinterpret_cast<float4*>(obs + row * 10);
    if (E == 2) {
      for (int j = 0; j < 5; ++j) { const bool mine = flushw ? ((j & 1) ? role == 1 : role == 3) : role == 3; if (mine) st16(d + j * 64 + lane, acc, nto); }
    } else {
      if (flushw == 0) { if (role == 3) { st16(d + lane, acc, nto); st16(d + 64 + lane, acc, nto); if (lane < 32) st16(d + 128 + lane, acc, nto); } }
      else { if (role == 3) { st16(d + lane, acc, nto); if (lane < 32) st16(d + 128 + lane, acc, nto); } if (role == 1) st16(d + 64 + lane, acc, nto); }
    }
    if (role == 2) {
      if (E == 2) { st8(rew + row + 2 * lane, acc, ntr); st2(done + row + 2 * lane, t, ntd); st2(res + row + 2 * lane, t, ntd); }
      else { st4(rew + row + lane, acc, ntr); st1(done + row + lane, t, ntd); st1(res + row + lane, t, ntd); }
    } else if (role == 0) {
      if (E == 2) st8(act + row + 2 * lane, acc, ntr); else st4(act + row + lane, acc, ntr);
    }
    __syncthreads();
  }
  if (acc == 123.456f) sink[0] = acc;
}
int main(int argc, char** argv) {
  const long n = 65536; const int T = argc > 1 ? atoi(argv[1]) : 256;
  float* obs[2]; int* act[2]; float* rew[2]; unsigned char *done[2], *res[2]; float* sink;
  for (int b = 0; b < 2; ++b) { (void)hipMalloc(&obs[b], n * T * 40); (void)hipMalloc(&act[b], n * T * 4); (void)hipMalloc(&rew[b], n * T * 4); (void)hipMalloc(&done[b], n * T); (void)hipMalloc(&res[b], n * T); }
  (void)hipMalloc(&sink, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  struct Case { int e, pol, fw; const char* name; };
  const Case cases[] = {{1, 7, 0, "E=1 all nt (the round-3 kernel's policy)"}, {1, 0, 0, "E=1 all plain"}, {1, 4, 0, "E=1 done/result nt, the rest plain"},
                        {1, 6, 0, "E=1 small arrays nt, observations plain"}, {1, 4, 1, "E=1 done/result nt, rest plain, 2 waves flush"},
                        {2, 7, 0, "E=2 all nt"}, {2, 0, 0, "E=2 all plain"}, {2, 4, 0, "E=2 done/result nt, the rest plain"}, {2, 0, 1, "E=2 all plain, 2 waves flush"},
                        {1, 7, 0, "E=1 all nt (again)"}, {1, 4, 0, "E=1 done/result nt, the rest plain (again)"}, {2, 0, 0, "E=2 all plain (again)"}};
  for (const Case& c : cases) {
    auto go = [&](int i) {
      const int b = i & 1;
      if (c.e == 2) hipLaunchKernelGGL(k<2>, dim3(n / 128), dim3(256), 0, 0, obs[b], act[b], rew[b], done[b], res[b], n, T, c.pol, c.fw, sink);
      else hipLaunchKernelGGL(k<1>, dim3(n / 64), dim3(256), 0, 0, obs[b], act[b], rew[b], done[b], res[b], n, T, c.pol, c.fw, sink);
    };
    for (int i = 0; i < 40; ++i) go(i);
    (void)hipDeviceSynchronize();
    std::vector<float> ms;
    for (int r = 0; r < 5; ++r) {
      (void)hipEventRecord(e0);
      for (int i = 0; i < 16; ++i) go(i);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float m; (void)hipEventElapsedTime(&m, e0, e1); ms.push_back(m / 16);
    }
    std::sort(ms.begin(), ms.end());
    const double us = ms[2] * 1e3;
    printf("%-52s %8.1f us/launch (min %6.1f max %6.1f) %6.3f us/iter %6.2f TB/s\n", c.name, us, ms[0] * 1e3, ms[4] * 1e3, us / T, n * T * 50.0 / us / 1e6);
  }
  return 0;
}
