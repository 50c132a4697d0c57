// sync_cost.hip -- what does one hand-off between the role waves of a 4-wave group cost on gfx950?
//   (a) s_barrier alone, (b) LDS write -> s_barrier -> LDS read (the engine's per-cycle hand-off),
//   (c) barrier-free: sequence flags in LDS (producer: data then flag, same wave => in order; consumer: spins on the flag),
//       ring of depth D, with credits so that a producer never overruns its consumers.
// 1024 groups x 4 waves (one group per SIMD quartet, as the rollout kernel at 65 536 envs).
//   hipcc -O3 --offload-arch=gfx950 -o sync_cost sync_cost.hip && ./sync_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define ITERS 2048

__device__ __forceinline__ float work(float a, int n) {
  for (int k = 0; k < n; k += 8)
    asm volatile("v_fma_f32 %0, %0, %1, %1\nv_fma_f32 %0, %0, %1, %1\nv_fma_f32 %0, %0, %1, %1\nv_fma_f32 %0, %0, %1, %1\n"
                 "v_fma_f32 %0, %0, %1, %1\nv_fma_f32 %0, %0, %1, %1\nv_fma_f32 %0, %0, %1, %1\nv_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(1.0001f));
  return a;
}

__global__ __launch_bounds__(256) void k_barrier(float* out, int n_work) {
  float a = threadIdx.x;
  for (int s = 0; s < ITERS; ++s) { a = work(a, n_work); __builtin_amdgcn_s_barrier(); }
  if (a == 1234.5f) out[threadIdx.x] = a;
}

__global__ __launch_bounds__(256) void k_barrier_lds(float* out, int n_work) {
  __shared__ float buf[2][4][64];
  const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
  float a = threadIdx.x;
  for (int s = 0; s < ITERS; ++s) {
    a += buf[(s + 1) & 1][(role + 3) & 3][lane];           // what the previous stage wrote in the previous iteration
    a = work(a, n_work);
    buf[s & 1][role][lane] = a;
    __syncthreads();
  }
  if (a == 1234.5f) out[threadIdx.x] = a;
}

// barrier-free chain 0 -> 1 -> 2 -> 3 (each stage consumes what the previous one produced for the same step)
template <int D>
__global__ __launch_bounds__(256) void k_flags(float* out, int n_work) {
  __shared__ float buf[4][D][64];
  __shared__ volatile int produced[4];                     // steps stage r has published
  __shared__ volatile int consumed[4];                     // steps stage r has finished reading from stage r-1
  const int lane = threadIdx.x & 63, role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (threadIdx.x < 4) { produced[threadIdx.x] = 0; consumed[threadIdx.x] = 0; }
  __syncthreads();
  float a = threadIdx.x;
  for (int s = 0; s < ITERS; ++s) {
    if (role > 0) {                                        // wait for the producer's step s
      while (produced[role - 1] <= s) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      a += buf[role - 1][s % D][lane];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      if (lane == 0) consumed[role] = s + 1;
    }
    a = work(a, n_work);
    if (role < 3) {                                        // wait for a free ring entry, publish
      while (consumed[role + 1] + D <= s) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      buf[role][s % D][lane] = a;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      if (lane == 0) produced[role] = s + 1;
    }
  }
  if (a == 1234.5f) out[threadIdx.x] = a;
}

template <typename K>
static double time_kernel(K kernel, float* out, int n_work) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(kernel, dim3(1024), dim3(256), 0, 0, out, n_work);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(kernel, dim3(1024), dim3(256), 0, 0, out, n_work);
  CHK(hipEventRecord(e1));
  CHK(hipDeviceSynchronize());
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e6 / 20 / ITERS;                            // ns per iteration
}

int main() {
  float* out;
  CHK(hipMalloc(&out, 4096));
  for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(k_barrier_lds, dim3(1024), dim3(256), 0, 0, out, 64);
  CHK(hipDeviceSynchronize());
  printf("%-44s %8s %8s %8s %8s\n", "ns per iteration, dependent fma per wave:", "0", "64", "128", "256");
  const int works[4] = {0, 64, 128, 256};
  double r[4];
  for (int w = 0; w < 4; ++w) r[w] = time_kernel(k_barrier, out, works[w]);
  printf("%-44s %8.1f %8.1f %8.1f %8.1f\n", "s_barrier", r[0], r[1], r[2], r[3]);
  for (int w = 0; w < 4; ++w) r[w] = time_kernel(k_barrier_lds, out, works[w]);
  printf("%-44s %8.1f %8.1f %8.1f %8.1f\n", "LDS write -> s_barrier -> LDS read", r[0], r[1], r[2], r[3]);
  for (int w = 0; w < 4; ++w) r[w] = time_kernel(k_flags<2>, out, works[w]);
  printf("%-44s %8.1f %8.1f %8.1f %8.1f\n", "LDS sequence flags, ring depth 2", r[0], r[1], r[2], r[3]);
  for (int w = 0; w < 4; ++w) r[w] = time_kernel(k_flags<4>, out, works[w]);
  printf("%-44s %8.1f %8.1f %8.1f %8.1f\n", "LDS sequence flags, ring depth 4", r[0], r[1], r[2], r[3]);
  for (int w = 0; w < 4; ++w) r[w] = time_kernel(k_flags<8>, out, works[w]);
  printf("%-44s %8.1f %8.1f %8.1f %8.1f\n", "LDS sequence flags, ring depth 8", r[0], r[1], r[2], r[3]);
  return 0;
}
