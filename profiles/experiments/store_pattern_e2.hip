// store_pattern_e2.hip -- the rollout record's STORE PATTERN with two envs per lane (round 4): a workgroup of four waves per 128 envs;
// per iteration and group 5120 B of observations (five 16-byte-per-lane stores by wave 3), action + reward as 8 B per lane
// (waves 0 and 2: 512 contiguous bytes), done + result as 2 B per lane (wave 2: one whole 128-byte line each); one s_barrier per
// iteration.  Compared in the same process with the 64-env pattern of store_pattern.hip (case "e1").  No arithmetic.
//   hipcc --offload-arch=gfx950 -O3 -o store_pattern_e2 store_pattern_e2.hip && ./store_pattern_e2 256
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ void st16(float4* p, float a) {
  const v4f w = {a, a, a, a};
  if (NT) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(w) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(w) : "memory");
}
template <bool NT> __device__ __forceinline__ void st8(void* p, float a) {
  const v2f w = {a, a};
  if (NT) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(w) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(w) : "memory");
}
template <bool NT> __device__ __forceinline__ void st4(void* p, float a) {
  if (NT) asm volatile("global_store_dword %0, %1, off nt" ::"v"(p), "v"(a) : "memory");
  else asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(a) : "memory");
}
template <bool NT> __device__ __forceinline__ void st2(void* p, int a) {
  if (NT) asm volatile("global_store_short %0, %1, off nt" ::"v"(p), "v"(a) : "memory");
  else asm volatile("global_store_short %0, %1, off" ::"v"(p), "v"(a) : "memory");
}
template <bool NT> __device__ __forceinline__ void st1(void* p, int a) {
  if (NT) asm volatile("global_store_byte %0, %1, off nt" ::"v"(p), "v"(a) : "memory");
  else asm volatile("global_store_byte %0, %1, off" ::"v"(p), "v"(a) : "memory");
}

// E = envs per lane (1: 64-env groups as the round-3 kernel; 2: 128-env groups)
template <int E, bool NT>
__global__ __launch_bounds__(256) void k(float* obs, int* act, float* rew, unsigned char* done, unsigned char* res, long n, int T,
                                         int which, float* sink) {
  const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
  const long first = (long)blockIdx.x * 64 * E;
  float acc = (float)lane;
  for (int t = 0; t < T; ++t) {
    const long row = (long)t * n + first;
    if (role == 3) {
      if (which & 1) {
        float4* d = reinterpret_cast<float4*>(obs + row * 10);
        if (E == 2) { for (int j = 0; j < 5; ++j) st16<NT>(d + j * 64 + lane, acc); }
        else { st16<NT>(d + lane, acc); st16<NT>(d + 64 + lane, acc); if (lane < 32) st16<NT>(d + 128 + lane, acc); }
      }
    } else if (role == 2) {
      if (which & 2) { if (E == 2) st8<NT>(rew + row + 2 * lane, acc); else st4<NT>(rew + row + lane, acc); }
      if (which & 8) { if (E == 2) st2<NT>(done + row + 2 * lane, t); else st1<NT>(done + row + lane, t); }
      if (which & 16) { if (E == 2) st2<NT>(res + row + 2 * lane, t); else st1<NT>(res + row + lane, t); }
    } else if (role == 0) {
      if (which & 4) { if (E == 2) st8<NT>(act + row + 2 * lane, acc); else st4<NT>(act + row + lane, acc); }
    }
    __syncthreads();
  }
  if (acc == 123.456f) sink[0] = acc;
}

int main(int argc, char** argv) {
  const long n = argc > 2 ? atol(argv[2]) : 65536; const int T = argc > 1 ? atoi(argv[1]) : 256; const int nbuf = 2;
  float* obs[2]; int* act[2]; float* rew[2]; unsigned char *done[2], *res[2]; float* sink;
  for (int b = 0; b < nbuf; ++b) {
    (void)hipMalloc(&obs[b], n * T * 40); (void)hipMalloc(&act[b], n * T * 4); (void)hipMalloc(&rew[b], n * T * 4);
    (void)hipMalloc(&done[b], n * T); (void)hipMalloc(&res[b], n * T);
  }
  (void)hipMalloc(&sink, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("T=%d  N=%ld  bytes per iteration %.2f MB\n", T, n, n * 50 / 1e6);
  struct Case { int e, nt, which; const char* name; };
  const Case cases[] = {{1, 1, 31, "e1 nt  all (64-env groups, byte done/result)"}, {2, 1, 31, "e2 nt  all (128-env groups, whole lines)"},
                        {1, 0, 31, "e1 plain all"}, {2, 0, 31, "e2 plain all"},
                        {2, 1, 1, "e2 nt  obs only"}, {2, 1, 7, "e2 nt  obs + reward + action"}, {2, 1, 0, "e2 no stores (barrier loop)"},
                        {1, 1, 31, "e1 nt  all (again)"}, {2, 1, 31, "e2 nt  all (again)"}};
  for (const Case& c : cases) {
    auto go = [&](int i) {
      const int b = i & 1;
      if (c.e == 2 && c.nt) hipLaunchKernelGGL((k<2, true>), dim3(n / 128), dim3(256), 0, 0, obs[b], act[b], rew[b], done[b], res[b], n, T, c.which, sink);
      else if (c.e == 2) hipLaunchKernelGGL((k<2, false>), dim3(n / 128), dim3(256), 0, 0, obs[b], act[b], rew[b], done[b], res[b], n, T, c.which, sink);
      else if (c.nt) hipLaunchKernelGGL((k<1, true>), dim3(n / 64), dim3(256), 0, 0, obs[b], act[b], rew[b], done[b], res[b], n, T, c.which, sink);
      else hipLaunchKernelGGL((k<1, false>), dim3(n / 64), dim3(256), 0, 0, obs[b], act[b], rew[b], done[b], res[b], n, T, c.which, sink);
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
    const double per = ((c.which & 1) ? 40 : 0) + ((c.which & 2) ? 4 : 0) + ((c.which & 4) ? 4 : 0) + ((c.which & 8) ? 1 : 0) + ((c.which & 16) ? 1 : 0);
    printf("%-52s %8.1f us/launch (min %6.1f max %6.1f) %6.3f us/iter %6.2f TB/s\n", c.name, us, ms[0] * 1e3, ms[4] * 1e3, us / T, n * T * per / us / 1e6);
  }
  return 0;
}
