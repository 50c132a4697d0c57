// store_pattern3.hip -- why does the rollout record's observation stream (one wave per workgroup storing a contiguous block per
// iteration, one s_barrier per iteration) stop at ~5.5 TB/s when a plain fill of the same buffer runs at ~6.9 TB/s on the same box?
// Observation stream only ([T][N][10] floats), variants of WHO stores and HOW (round 4).
//   hipcc --offload-arch=gfx950 -O3 -o store_pattern3 store_pattern3.hip && ./store_pattern3 256
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ void st16(float4* p, float a) {
  const v4f w = {a, a, a, a};
  if (NT) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(w) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ float chain(float a, int n) {
  for (int k = 0; k < n; k += 8)
    asm volatile("v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\n"
                 "v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(1.0001f), "v"(0.5f));
  return a;
}

// G = envs per workgroup (64, 128, 256); waves = 4; `spread`: 0 = wave 3 stores the whole block, 1 = all four waves store a quarter
// each (interleaved 1 KB pieces), 2 = waves 2 and 3 store half each; `sync`: 1 = s_barrier per iteration, 0 = none;
// `work` = dependent fma per wave and iteration (emulates the arithmetic)
template <int G, bool NT>
__global__ __launch_bounds__(256) void k(float* obs, long n, int T, int spread, int sync, int work, float* sink) {
  const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
  const long first = (long)blockIdx.x * G;
  constexpr int PIECES = G * 10 / 4 / 64;      // 1 KB pieces per block: 2.5 -> handled as 3 with a half (G = 64), 5, 10
  float acc = (float)lane;
  for (int t = 0; t < T; ++t) {
    float4* d = reinterpret_cast<float4*>(obs + ((long)t * n + first) * 10);
    acc = chain(acc, work);
    if (G == 64) {
      if (spread == 0 ? role == 3 : true) {
        if (spread == 0) { st16<NT>(d + lane, acc); st16<NT>(d + 64 + lane, acc); if (lane < 32) st16<NT>(d + 128 + lane, acc); }
        else if (role < 2) st16<NT>(d + role * 64 + lane, acc);
        else if (role == 2 && lane < 32) st16<NT>(d + 128 + lane, acc);
      }
    } else {
      for (int j = 0; j < PIECES; ++j) {
        const bool mine = spread == 0 ? role == 3 : spread == 1 ? (j & 3) == role : (role >= 2 && (j & 1) == (role & 1));
        if (mine) st16<NT>(d + j * 64 + lane, acc);
      }
    }
    if (sync) __syncthreads();
  }
  if (acc == 123.456f) sink[0] = acc;
}
// reference: the same bytes in plain linear order (a fill)
template <bool NT>
__global__ __launch_bounds__(256) void fill(float* obs, long n4, float* sink) {
  float4* d = reinterpret_cast<float4*>(obs);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) st16<NT>(d + i, 1.0f);
}

int main(int argc, char** argv) {
  const long n = 65536; const int T = argc > 1 ? atoi(argv[1]) : 256;
  float* obs[2]; float* sink;
  for (int b = 0; b < 2; ++b) (void)hipMalloc(&obs[b], n * T * 40);
  (void)hipMalloc(&sink, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  struct Case { int g, nt, spread, sync, work; const char* name; };
  const Case cases[] = {
      {0, 0, 0, 0, 0, "fill (linear order), plain"}, {0, 1, 0, 0, 0, "fill (linear order), nt"},
      {64, 1, 0, 1, 0, "G=64  nt wave 3 stores, barrier"}, {128, 1, 0, 1, 0, "G=128 nt wave 3 stores, barrier"},
      {256, 1, 0, 1, 0, "G=256 nt wave 3 stores, barrier"}, {128, 0, 0, 1, 0, "G=128 plain wave 3 stores, barrier"},
      {128, 1, 1, 1, 0, "G=128 nt all four waves store, barrier"}, {128, 1, 2, 1, 0, "G=128 nt waves 2+3 store, barrier"},
      {128, 1, 0, 0, 0, "G=128 nt wave 3 stores, NO barrier"}, {128, 1, 1, 0, 0, "G=128 nt all four waves, NO barrier"},
      {64, 1, 1, 1, 0, "G=64  nt waves 0-2 store, barrier"},
      {128, 1, 0, 1, 128, "G=128 nt wave 3 stores, barrier, 128 fma/wave"}, {128, 1, 1, 1, 128, "G=128 nt four waves, barrier, 128 fma/wave"},
      {128, 1, 0, 1, 256, "G=128 nt wave 3 stores, barrier, 256 fma/wave"}, {64, 1, 0, 1, 128, "G=64  nt wave 3 stores, barrier, 128 fma/wave"},
      {0, 1, 0, 0, 0, "fill (linear order), nt (again)"}, {128, 1, 0, 1, 0, "G=128 nt wave 3 stores, barrier (again)"}};
  for (const Case& c : cases) {
    auto go = [&](int i) {
      float* o = obs[i & 1];
      if (c.g == 0) { if (c.nt) hipLaunchKernelGGL(fill<true>, dim3(2048), dim3(256), 0, 0, o, n * T * 10 / 4, sink); else hipLaunchKernelGGL(fill<false>, dim3(2048), dim3(256), 0, 0, o, n * T * 10 / 4, sink); }
      else if (c.g == 64) hipLaunchKernelGGL((k<64, true>), dim3(n / 64), dim3(256), 0, 0, o, n, T, c.spread, c.sync, c.work, sink);
      else if (c.g == 256) hipLaunchKernelGGL((k<256, true>), dim3(n / 256), dim3(256), 0, 0, o, n, T, c.spread, c.sync, c.work, sink);
      else if (c.nt) hipLaunchKernelGGL((k<128, true>), dim3(n / 128), dim3(256), 0, 0, o, n, T, c.spread, c.sync, c.work, sink);
      else hipLaunchKernelGGL((k<128, false>), dim3(n / 128), dim3(256), 0, 0, o, n, T, c.spread, c.sync, c.work, sink);
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
    printf("%-52s %8.1f us/launch (min %6.1f max %6.1f) %6.3f us/iter %6.2f TB/s\n", c.name, us, ms[0] * 1e3, ms[4] * 1e3, us / T, n * T * 40.0 / us / 1e6);
  }
  return 0;
}
