// store_pattern4.hip -- what is the streaming-WRITE ceiling of this box for the observation stream (671 MB per launch-equivalent),
// by kernel shape?  (torch's fill_ reads 6.8-6.9 TB/s in bench.py's box_fill_gbs; the rollout pipeline's own pattern 5.2-5.7.)
//   hipcc --offload-arch=gfx950 -O3 -o store_pattern4 store_pattern4.hip && ./store_pattern4
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
// A: torch-like: a block of 256 threads writes a contiguous 16 KB region (4 x 16 B per thread at 4 KB stride)
template <bool NT> __global__ __launch_bounds__(256) void fillA(float4* d, long n4) {
  const long base = (long)blockIdx.x * 1024 + threadIdx.x;
  for (int j = 0; j < 4; ++j) if (base + j * 256 < n4) st16<NT>(d + base + j * 256, 1.0f);
}
// B: one wave per (step, group) block of 2560 B, launched in time-major order (grid = T x 1024)
template <bool NT> __global__ __launch_bounds__(64) void fillB(float4* d) {
  float4* p = d + (long)blockIdx.x * 160;
  const int lane = threadIdx.x;
  st16<NT>(p + lane, 1.0f); st16<NT>(p + 64 + lane, 1.0f); if (lane < 32) st16<NT>(p + 128 + lane, 1.0f);
}
// C: persistent: 1024 blocks x 256 threads, block g loops over t and ALL four waves write a quarter of the 2560-B block (640 B each:
// 40 lanes x 16 B), no barrier
template <bool NT> __global__ __launch_bounds__(256) void fillC(float4* d, int T) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int t = 0; t < T; ++t) { float4* p = d + ((long)t * 1024 + blockIdx.x) * 160 + w * 40; if (lane < 40) st16<NT>(p + lane, 1.0f); }
}
// D: persistent, one wave per block stores the whole 2560 B, 4 waves per block handle 4 consecutive groups? no: 4096 blocks of 64 threads
template <bool NT> __global__ __launch_bounds__(64) void fillD(float4* d, int T) {
  const int lane = threadIdx.x;
  for (int t = 0; t < T; ++t) { float4* p = d + ((long)t * 1024 + (blockIdx.x & 1023)) * 160; if ((blockIdx.x >> 10) == (t & 3)) { st16<NT>(p + lane, 1.0f); st16<NT>(p + 64 + lane, 1.0f); if (lane < 32) st16<NT>(p + 128 + lane, 1.0f); } }
}
int main() {
  const long n = 65536; const int T = 256; const long n4 = n * T * 10 / 4;
  float4* obs[2];
  for (int b = 0; b < 2; ++b) (void)hipMalloc(&obs[b], n4 * 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const char* names[] = {"A torch-like 16 KB per block, plain", "A torch-like 16 KB per block, nt", "B one wave per (step, group) block, plain", "B one wave per (step, group) block, nt",
                         "C persistent, 4 waves x 640 B per step, plain", "C persistent, 4 waves x 640 B per step, nt", "D persistent 4096 waves, each every 4th step, plain", "D persistent 4096 waves, each every 4th step, nt"};
  for (int c = 0; c < 8; ++c) {
    auto go = [&](int i) {
      float4* o = obs[i & 1];
      switch (c) {
        case 0: hipLaunchKernelGGL(fillA<false>, dim3((n4 + 1023) / 1024), dim3(256), 0, 0, o, n4); break;
        case 1: hipLaunchKernelGGL(fillA<true>, dim3((n4 + 1023) / 1024), dim3(256), 0, 0, o, n4); break;
        case 2: hipLaunchKernelGGL(fillB<false>, dim3(T * 1024), dim3(64), 0, 0, o); break;
        case 3: hipLaunchKernelGGL(fillB<true>, dim3(T * 1024), dim3(64), 0, 0, o); break;
        case 4: hipLaunchKernelGGL(fillC<false>, dim3(1024), dim3(256), 0, 0, o, T); break;
        case 5: hipLaunchKernelGGL(fillC<true>, dim3(1024), dim3(256), 0, 0, o, T); break;
        case 6: hipLaunchKernelGGL(fillD<false>, dim3(4096), dim3(64), 0, 0, o, T); break;
        case 7: hipLaunchKernelGGL(fillD<true>, dim3(4096), dim3(64), 0, 0, o, T); break;
      }
    };
    for (int i = 0; i < 20; ++i) go(i);
    (void)hipDeviceSynchronize();
    std::vector<float> ms;
    for (int r = 0; r < 5; ++r) {
      (void)hipEventRecord(e0);
      for (int i = 0; i < 16; ++i) go(i);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float m; (void)hipEventElapsedTime(&m, e0, e1); ms.push_back(m / 16);
    }
    std::sort(ms.begin(), ms.end());
    printf("%-56s %8.1f us (min %6.1f max %6.1f) %6.2f TB/s\n", names[c], ms[2] * 1e3, ms[0] * 1e3, ms[4] * 1e3, n4 * 16.0 / (ms[2] * 1e3) / 1e6);
  }
  return 0;
}
