// wave_placement.hip -- where does the dispatcher put the waves of a small grid?  Launches
// <blocks> workgroups of <threads> threads with <lds> bytes of LDS (the geometry of the
// wave-specialised rollout kernel: 512 x 256, 26 KB), every wave records HW_ID / XCC_ID and spins
// long enough for the whole grid to be co-resident.  Prints waves per SIMD and workgroups per CU.
//   hipcc -O3 --offload-arch=gfx950 -o wave_placement wave_placement.hip && ./wave_placement 512 256 26112
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <map>
#include <tuple>

__global__ void probe(uint32_t* out, int spin) {
  extern __shared__ float lds[];
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float acc = threadIdx.x;
  for (int i = 0; i < spin; ++i) acc = fmaf(acc, 1.0001f, 0.5f);
  lds[threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) {
    int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    out[2 * w] = hw; out[2 * w + 1] = xcc;
  }
  if (acc == 12345.0f) out[0] = (uint32_t)lds[(threadIdx.x + 1) % blockDim.x];
}

int main(int argc, char** argv) {
  int blocks = argc > 1 ? atoi(argv[1]) : 512, threads = argc > 2 ? atoi(argv[2]) : 256;
  int lds = argc > 3 ? atoi(argv[3]) : 26112;
  int waves = blocks * threads / 64;
  uint32_t* d; hipMalloc(&d, waves * 8);
  uint32_t* h = (uint32_t*)malloc(waves * 8);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), lds, 0, d, 20000);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, d, waves * 8, hipMemcpyDeviceToHost);
  std::map<std::tuple<int, int, int, int>, int> per_simd;   // xcc, se, cu, simd
  std::map<std::tuple<int, int, int>, int> per_cu;
  for (int w = 0; w < waves; ++w) {
    uint32_t hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
    int simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_simd[{(int)xcc, se * 2 + sh, cu, simd}]++;
    per_cu[{(int)xcc, se * 2 + sh, cu}]++;
  }
  const int wpb = threads / 64;
  for (int k = 0; k < wpb; ++k) {           // which SIMD does wave k of a workgroup get?
    int cnt[4] = {0, 0, 0, 0};
    for (int b = 0; b < blocks; ++b) cnt[(h[2 * (b * wpb + k)] >> 4) & 3]++;
    printf("  wave %d of a workgroup -> SIMD0 %d, SIMD1 %d, SIMD2 %d, SIMD3 %d\n", k, cnt[0], cnt[1], cnt[2], cnt[3]);
  }
  // per CU: how many waves with the SAME index k share a SIMD (k = role in the pipeline kernel), and the TG_ID values
  std::map<std::tuple<int, int, int, int, int>, int> same_k;  // xcc, se, cu, simd, k
  std::map<std::tuple<int, int, int>, std::map<int, int>> tg_of_cu;
  for (int w = 0; w < waves; ++w) {
    uint32_t hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
    int simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, tg = (hw >> 16) & 0xf;
    same_k[{(int)xcc, se * 2 + sh, cu, simd, w % wpb}]++;
    if (w % wpb == 0) tg_of_cu[{(int)xcc, se * 2 + sh, cu}][tg]++;
  }
  std::map<int, int> hist_same, hist_tg;
  for (auto& kv : same_k) hist_same[kv.second]++;
  for (auto& kv : hist_same) printf("  (SIMD, wave index) pairs holding %d waves of that index: %d\n", kv.first, kv.second);
  for (auto& kv : tg_of_cu) { int key = 0; for (auto& t : kv.second) key |= 1 << t.first; hist_tg[key]++; }
  for (auto& kv : hist_tg) printf("  CUs whose workgroups have TG_ID set 0x%x: %d\n", kv.first, kv.second);
  std::map<int, int> hist_simd, hist_cu;
  for (auto& kv : per_simd) hist_simd[kv.second]++;
  for (auto& kv : per_cu) hist_cu[kv.second]++;
  printf("grid %d x %d, lds %d B: %d waves on %zu SIMDs of %zu CUs\n", blocks, threads, lds, waves, per_simd.size(), per_cu.size());
  for (auto& kv : hist_simd) printf("  SIMDs holding %d waves: %d\n", kv.first, kv.second);
  for (auto& kv : hist_cu) printf("  CUs holding %d waves: %d\n", kv.first, kv.second);
  return 0;
}
