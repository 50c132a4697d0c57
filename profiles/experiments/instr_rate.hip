// instr_rate.hip -- issue-rate microbenchmark for the VALU instructions the engine's hot loops
// are made of (gfx950).  Each kernel runs 8 independent dependency chains of one instruction,
// so the figure is THROUGHPUT (issue slots), not latency.  Output: ns per wave-instruction per
// SIMD and the ratio to v_fma_f32.  Used to price Philox (v_mad_u64_u32) against fp32 work.
//   hipcc -O3 --offload-arch=gfx950 -o instr_rate instr_rate.hip && ./instr_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITERS 4096
#define REP8(x) x x x x x x x x

#define KERNEL(name, decl, body, sink)                                          \
  __global__ void name(uint32_t* out, uint32_t seed) {                          \
    decl;                                                                       \
    for (int it = 0; it < ITERS; ++it) { REP8(body) }                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;                          \
  }

#define F8 float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; float m = 1.0001f, c = 0.5f
#define U8 uint32_t a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; uint32_t m = 0xD2511F53u
#define OP8(op) asm volatile(op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
#define OP8_2(op) asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
#define OP8_1(op) asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));

KERNEL(k_fma, F8, OP8("v_fma_f32"), (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
KERNEL(k_mul, F8, OP8_2("v_mul_f32"), (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
KERNEL(k_rcp, F8, OP8_1("v_rcp_f32"), (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
KERNEL(k_sqrt, F8, OP8_1("v_sqrt_f32"), (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
KERNEL(k_xor, U8, OP8_2("v_xor_b32"), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)
KERNEL(k_mul_lo, U8, OP8_2("v_mul_lo_u32"), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)
KERNEL(k_mul_hi, U8, OP8_2("v_mul_hi_u32"), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)
KERNEL(k_mul_u24, U8, OP8_2("v_mul_u32_u24"), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)

// v_mad_u64_u32 vdst[2], vcc, a, b, c[2]: 8 chains on 64-bit accumulators
__global__ void k_mad_u64(uint32_t* out, uint32_t seed) {
  uint64_t a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  uint32_t m = 0xD2511F53u;
  for (int it = 0; it < ITERS; ++it) {
    REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %8, %0\nv_mad_u64_u32 %1, vcc, %8, %8, %1\nv_mad_u64_u32 %2, vcc, %8, %8, %2\nv_mad_u64_u32 %3, vcc, %8, %8, %3\n"
                      "v_mad_u64_u32 %4, vcc, %8, %8, %4\nv_mad_u64_u32 %5, vcc, %8, %8, %5\nv_mad_u64_u32 %6, vcc, %8, %8, %6\nv_mad_u64_u32 %7, vcc, %8, %8, %7"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
// v_pk_fma_f32 on register pairs
__global__ void k_pk_fma(uint32_t* out, uint32_t seed) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a0 = {(float)seed, 1}, a1 = {(float)seed, 2}, a2 = {(float)seed, 3}, a3 = {(float)seed, 4}, a4 = {(float)seed, 5}, a5 = {(float)seed, 6}, a6 = {(float)seed, 7}, a7 = {(float)seed, 8};
  f2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
  for (int it = 0; it < ITERS; ++it) { REP8(OP8("v_pk_fma_f32")) }
  f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s.x + s.y);
}

// single dependency chains: latency of a dependent issue (one wave per SIMD shows it directly)
__global__ void k_fma_dep(uint32_t* out, uint32_t seed) {
  float a0 = seed, m = 1.0001f, c = 0.5f;
  for (int it = 0; it < ITERS; ++it) {
    REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\n"
                      "v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)a0;
}
__global__ void k_xor_dep(uint32_t* out, uint32_t seed) {
  uint32_t a0 = seed, m = 0xD2511F53u;
  for (int it = 0; it < ITERS; ++it) {
    REP8(asm volatile("v_xor_b32 %0, %0, %1\nv_xor_b32 %0, %0, %1\nv_xor_b32 %0, %0, %1\nv_xor_b32 %0, %0, %1\n"
                      "v_xor_b32 %0, %0, %1\nv_xor_b32 %0, %0, %1\nv_xor_b32 %0, %0, %1\nv_xor_b32 %0, %0, %1" : "+v"(a0) : "v"(m));)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0;
}
__global__ void k_mad_u64_dep(uint32_t* out, uint32_t seed) {
  uint64_t a0 = seed; uint32_t m = 0xD2511F53u;
  for (int it = 0; it < ITERS; ++it) {
    REP8(asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0\nv_mad_u64_u32 %0, vcc, %1, %1, %0\nv_mad_u64_u32 %0, vcc, %1, %1, %0\nv_mad_u64_u32 %0, vcc, %1, %1, %0\n"
                      "v_mad_u64_u32 %0, vcc, %1, %1, %0\nv_mad_u64_u32 %0, vcc, %1, %1, %0\nv_mad_u64_u32 %0, vcc, %1, %1, %0\nv_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(a0) : "v"(m) : "vcc");)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)a0;
}
__global__ void k_rcp_dep(uint32_t* out, uint32_t seed) {
  float a0 = seed + 2.0f;
  for (int it = 0; it < ITERS; ++it) {
    REP8(asm volatile("v_rcp_f32 %0, %0\nv_rcp_f32 %0, %0\nv_rcp_f32 %0, %0\nv_rcp_f32 %0, %0\nv_rcp_f32 %0, %0\nv_rcp_f32 %0, %0\nv_rcp_f32 %0, %0\nv_rcp_f32 %0, %0" : "+v"(a0));)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)a0;
}
// the Philox round as the engine writes it (two multiplies, xors), 10 rounds per call, dependent calls
__global__ void k_philox_dep(uint32_t* out, uint32_t seed) {
  uint32_t c0 = seed, c1 = threadIdx.x, c2 = blockIdx.x, c3 = 7, k0s = seed * 3, k1s = seed * 5;
  for (int it = 0; it < ITERS * 64 / 10; ++it) {
    uint32_t k0 = k0s, k1 = k1s;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
      uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
      c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
      k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c1 ^ c2 ^ c3;
}

// same chains with only the lower 32 lanes of every wave active: does a half-empty wave64 issue faster?
__global__ void k_fma_half(uint32_t* out, uint32_t seed) {
  if (threadIdx.x & 32) return;
  F8;
  for (int it = 0; it < ITERS; ++it) { REP8(OP8("v_fma_f32")) }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__global__ void k_xor_half(uint32_t* out, uint32_t seed) {
  if (threadIdx.x & 32) return;
  U8;
  for (int it = 0; it < ITERS; ++it) { REP8(OP8_2("v_xor_b32")) }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void k_rcp_half(uint32_t* out, uint32_t seed) {
  if (threadIdx.x & 32) return;
  F8;
  for (int it = 0; it < ITERS; ++it) { REP8(OP8_1("v_rcp_f32")) }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}

// a never-taken guarded region as the compiler emits it for a rare per-lane event:
//   v_cmp -> s_and_saveexec -> s_cbranch_execz -> (skipped body) -> s_or exec ; one v_fma between regions
__global__ void k_guard(uint32_t* out, uint32_t seed) {
  float a0 = seed, m = 1.0001f, c = 0.5f, big = 1e30f;
  for (int it = 0; it < ITERS; ++it) {
    REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n"
                      "v_cmp_gt_f32 vcc, %0, %3\n"
                      "s_and_saveexec_b64 s[10:11], vcc\n"
                      "s_cbranch_execz 1f\n"
                      "v_mov_b32 %0, 0\n"
                      "1: s_or_b64 exec, exec, s[10:11]\n" : "+v"(a0) : "v"(m), "v"(c), "v"(big) : "vcc", "s10", "s11");)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)a0;
}
// the same decision as a select (no branch): v_cmp + v_cndmask
__global__ void k_select(uint32_t* out, uint32_t seed) {
  float a0 = seed, m = 1.0001f, c = 0.5f, big = 1e30f;
  for (int it = 0; it < ITERS; ++it) {
    REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n"
                      "v_cmp_gt_f32 vcc, %0, %3\n"
                      "v_cndmask_b32 %0, %0, %1, vcc\n" : "+v"(a0) : "v"(m), "v"(c), "v"(big) : "vcc");)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)a0;
}
// wave-uniform scalar branch, never taken: s_cmp + s_cbranch_scc1
__global__ void k_sbranch(uint32_t* out, uint32_t seed) {
  float a0 = seed, m = 1.0001f, c = 0.5f;
  for (int it = 0; it < ITERS; ++it) {
    REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n"
                      "s_cmp_eq_u32 %3, 12345\n"
                      "s_cbranch_scc1 1f\n"
                      "v_fma_f32 %0, %0, %1, %2\n"
                      "1:\n" : "+v"(a0) : "v"(m), "v"(c), "s"(seed) : "scc");)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)a0;
}
// LDS round trip: ds_write_b32 + ds_read_b32 + s_waitcnt
__global__ void k_lds_rt(uint32_t* out, uint32_t seed) {
  __shared__ float buf[256];
  float a0 = seed; uint32_t addr = threadIdx.x * 4;
  for (int it = 0; it < ITERS; ++it) {
    REP8(asm volatile("ds_write_b32 %1, %0\nds_read_b32 %0, %1\ns_waitcnt lgkmcnt(0)\n" : "+v"(a0) : "v"(addr) : "memory");)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)a0 + (uint32_t)buf[0];
}

typedef void (*kern_t)(uint32_t*, uint32_t);
static double run(kern_t k, uint32_t* out, int blocks, int threads) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, 1u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5.0;
}

int main() {
  uint32_t* out; hipMalloc(&out, 1 << 24);
  struct { const char* name; kern_t k; } ks[] = {
    {"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_pk_fma_f32", k_pk_fma}, {"v_xor_b32", k_xor},
    {"v_mul_u32_u24", k_mul_u24}, {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi},
    {"v_mad_u64_u32", k_mad_u64}, {"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt},
    {"dep v_fma_f32", k_fma_dep}, {"dep v_xor_b32", k_xor_dep}, {"dep v_mad_u64_u32", k_mad_u64_dep},
    {"dep v_rcp_f32", k_rcp_dep}, {"guarded region (fma+cmp+saveexec+branch+or)/8", k_guard}, {"select (fma+cmp+cndmask)/8", k_select},
    {"scalar branch (fma+s_cmp+s_cbranch+fma)/8", k_sbranch}, {"LDS write+read+wait /8", k_lds_rt},
    {"half-wave v_fma_f32", k_fma_half}, {"half-wave v_xor_b32", k_xor_half}, {"half-wave v_rcp_f32", k_rcp_half}, {"philox round (per 1/64 it)", k_philox_dep}};
  // 1024 SIMDs; W waves per SIMD
  for (int W = 1; W <= 8; W *= 2) {
    int threads = 256, blocks = 256 * W;   // 256 CUs x W blocks of 4 waves -> W waves per SIMD
    double base = 0;
    printf("waves_per_simd=%d\n", W);
    for (auto& k : ks) {
      double ms = run(k.k, out, blocks, threads);
      double per = ms * 1e6 / ((double)ITERS * 64 * W);   // ns per wave-instruction per SIMD
      if (k.k == k_fma) base = per;
      printf("  %-16s %8.3f ms  %7.3f ns/instr/SIMD  x%.2f of v_fma_f32\n", k.name, ms, per, per / base);
    }
  }
  return 0;
}
