// instr_clocks.hip -- the issue rate of instr_rate.hip in SHADER CLOCKS (s_memtime around the loop, per wave), so that the figure
// does not depend on the clock the chip happens to grant: clocks per wave-instruction seen by ONE wave, and per SIMD (= that / waves
// per SIMD), for 1, 2, 4, 8 waves per SIMD.  8 independent chains per wave (throughput, not latency).
//   hipcc -O3 --offload-arch=gfx950 -o instr_clocks instr_clocks.hip && ./instr_clocks
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#define ITERS 2048
#define REP8(x) x x x x x x x x
#define F8 float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; float m = 1.0001f, c = 0.5f
#define U8 uint32_t a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; uint32_t m = 0xD2511F53u
#define OP8(op) asm volatile(op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
#define OP8_2(op) asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
#define OP8_1(op) asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#define KERNEL(name, decl, body, sink)                                                    \
  __global__ void name(uint32_t* out, uint64_t* clk, uint32_t seed) {                     \
    decl;                                                                                 \
    const uint64_t t0 = __builtin_amdgcn_s_memtime();                                     \
    for (int it = 0; it < ITERS; ++it) { REP8(body) }                                     \
    const uint64_t t1 = __builtin_amdgcn_s_memtime();                                     \
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;                                    \
    if ((threadIdx.x & 63) == 0) clk[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; \
  }
KERNEL(k_fma, F8, OP8("v_fma_f32"), (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
KERNEL(k_mul, F8, OP8_2("v_mul_f32"), (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
KERNEL(k_rcp, F8, OP8_1("v_rcp_f32"), (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
KERNEL(k_sqrt, F8, OP8_1("v_sqrt_f32"), (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
KERNEL(k_xor, U8, OP8_2("v_xor_b32"), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)
KERNEL(k_mul_hi, U8, OP8_2("v_mul_hi_u32"), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)
typedef void (*K)(uint32_t*, uint64_t*, uint32_t);
int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  uint32_t* out; uint64_t* clk; hipMalloc(&out, (size_t)cus * 4 * 8 * 64 * 4); hipMalloc(&clk, (size_t)cus * 4 * 8 * 8);
  struct { const char* n; K k; } ks[] = {{"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_xor_b32", k_xor}, {"v_mul_hi_u32", k_mul_hi}, {"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt}};
  for (int wps : {1, 2, 4, 8}) {
    printf("waves_per_simd=%d\n", wps);
    for (auto& e : ks) {
      const int waves = cus * 4 * wps;                    // blocks of 4 waves (one per SIMD), wps blocks per CU
      for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(e.k, dim3(cus * wps), dim3(256), 0, 0, out, clk, 1u);
      hipDeviceSynchronize();
      std::vector<uint64_t> h(waves); hipMemcpy(h.data(), clk, waves * 8, hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.end());
      const double per_wave = (double)h[waves / 2] / (ITERS * 64.0);
      printf("  %-14s median %7.2f clocks per instruction and wave   = %5.2f per SIMD   (p99 %.2f)\n", e.n, per_wave, per_wave / wps, (double)h[waves * 99 / 100] / (ITERS * 64.0));
    }
  }
  return 0;
}
