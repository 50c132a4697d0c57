// store_pattern.hip -- what the rollout record's STORE PATTERN costs on its own, in the pure-HBM regime (no arithmetic).
// Grid = N/64 workgroups of four waves (as the four-wave rollout pipeline); per iteration and group: 2560 B of observations
// (three 16-byte-per-lane stores by wave 3), action + reward (4 B per lane, waves 0 and 2), done + result (1 B per lane, wave 2);
// one s_barrier per iteration.  Time-major layout [T][N][...] (the API's layout), two buffers of T x 3.28 MB alternating.
// which: bit 0 obs, 1 reward, 2 action, 3 done, 4 result.
// small: 0 = as the kernel does it; 1 = done / result as one dword per lane from 16 lanes; 2 = the four small arrays stored for FOUR
// neighbouring groups by one of them (1 KB / 1 KB / 256 B / 256 B per instruction: what a 256-env workgroup could do);
// 3 = as 2 for done / result only; 4 = done / result for TWO neighbouring groups by one of them (128 B = one line).
//   hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip && ./store_pattern 256
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

__device__ __forceinline__ void store_byte(unsigned char* p, int v, int pol) {
  if (pol == 1) asm volatile("global_store_byte %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  else if (pol == 2) asm volatile("global_store_byte %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if (pol == 3) asm volatile("global_store_byte %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else if (pol == 4) asm volatile("global_store_byte %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  else *p = (unsigned char)v;
}
__global__ __launch_bounds__(256) void k(float* obs, int* act, float* rew, unsigned char* done, unsigned char* res, long n, int T,
                                         int which, int small, float* sink, int map = 0, int pol = 0) {
  const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
  // which env group a block works on: 0 = its index; 1 = blocks b and b + 8 (same XCD under round-robin dispatch) get neighbouring
  // groups 2m, 2m + 1; 2 = four neighbouring groups per XCD; 3 = every XCD a contiguous eighth of the groups
  long g = blockIdx.x;
  const long b = blockIdx.x, nb = gridDim.x;
  if (map == 1) g = (b / 16) * 16 + (b % 8) * 2 + (b / 8) % 2;
  else if (map == 2) g = (b / 32) * 32 + (b % 8) * 4 + (b / 8) % 4;
  else if (map == 3) g = (b % 8) * (nb / 8) + b / 8;
  const long first = g * 64;
  float acc = (float)lane;
  for (int t = 0; t < T; ++t) {
    const long row = (long)t * n + first;
    if (role == 3) {
      if (which & 1) {
        float4* d = reinterpret_cast<float4*>(obs + row * 10);
        const float4 v = make_float4(acc, acc, acc, acc);
        d[lane] = v; d[64 + lane] = v;
        if (lane < 32) d[128 + lane] = v;
      }
    } else if (role == 2) {
      const bool lead = small == 4 ? (g & 1) == 0 : (g & 3) == 0;
      const int wide = small == 4 ? 32 : 64;
      if (which & 2) {
        if (small == 2) { if (lead) reinterpret_cast<float4*>(rew + row)[lane] = make_float4(acc, acc, acc, acc); }
        else rew[row + lane] = acc;
      }
      if (which & 8) {
        if (small >= 2) { if (lead && lane < wide) reinterpret_cast<int*>(done + row)[lane] = t; }
        else if (small == 1) { if (lane < 16) reinterpret_cast<int*>(done + row)[lane] = t; }
        else store_byte(done + row + lane, t, pol);
      }
      if (which & 16) {
        if (small >= 2) { if (lead && lane < wide) reinterpret_cast<int*>(res + row)[lane] = t; }
        else if (small == 1) { if (lane < 16) reinterpret_cast<int*>(res + row)[lane] = t; }
        else store_byte(res + row + lane, lane, pol);
      }
    } else if (role == 0) {
      if (which & 4) {
        if (small == 2) { if ((g & 3) == 0) reinterpret_cast<int4*>(act + row)[lane] = make_int4(t, t, t, t); }
        else act[row + lane] = t;
      }
    }
    __syncthreads();
  }
  if (acc == 123.456f) sink[0] = acc;
}

int main(int argc, char** argv) {
  const long n = 65536; const int T = argc > 1 ? atoi(argv[1]) : 256; const int nbuf = 2;
  float* obs[2]; int* act[2]; float* rew[2]; unsigned char *done[2], *res[2]; float* sink;
  for (int b = 0; b < nbuf; ++b) {
    (void)hipMalloc(&obs[b], n * T * 40); (void)hipMalloc(&act[b], n * T * 4); (void)hipMalloc(&rew[b], n * T * 4);
    (void)hipMalloc(&done[b], n * T); (void)hipMalloc(&res[b], n * T);
  }
  (void)hipMalloc(&sink, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("T=%d  N=%ld  bytes per iteration %.2f MB\n", T, n, n * 50 / 1e6);
  struct Case { int which, small; const char* name; int map = 0; int pol = 0; };
  const Case cases[] = {{0, 0, "no stores (barrier loop)"}, {1, 0, "obs only"}, {31, 0, "all, as the kernel stores them"}, {3, 0, "obs + reward"},
                        {7, 0, "obs + reward + action"}, {15, 0, "obs + reward + action + done"}, {25, 0, "obs + done + result"},
                        {31, 1, "all, done/result as 16-lane dword stores"}, {31, 3, "all, done/result for 4 groups by one (256 B)"},
                        {31, 2, "all, small arrays for 4 groups by one (1 KB / 256 B)"}, {31, 4, "all, done/result for 2 groups by one (128 B)"}, {30, 0, "small arrays only"}, {31, 0, "all, done/result stores nt", 0, 1}, {31, 0, "all, done/result stores sc1", 0, 2},
                        {31, 0, "all, done/result stores sc0 sc1", 0, 3}, {31, 0, "all, done/result stores sc0", 0, 4}, {31, 0, "all, groups 2m / 2m+1 on one XCD", 1}, {31, 0, "all, four neighbouring groups per XCD", 2},
                        {31, 0, "all, a contiguous eighth of the groups per XCD", 3}, {30, 2, "small arrays only, 4 groups by one"}};
  for (const Case& c : cases) {
    auto go = [&](int i) { hipLaunchKernelGGL(k, dim3(n / 64), dim3(256), 0, 0, obs[i & 1], act[i & 1], rew[i & 1], done[i & 1], res[i & 1], n, T, c.which, c.small, sink, c.map, c.pol); };
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
    printf("%-58s %8.1f us/launch %6.3f us/iter %6.2f TB/s\n", c.name, us, us / T, n * T * per / us / 1e6);
  }
  return 0;
}
