// Where do the waves of small workgroups land?  Launches `blocks` workgroups of `threads` threads with `lds` bytes
// of LDS each, all resident at once (every wave spins until a deadline), and prints how many distinct SIMDs the
// waves of a CU occupy.  Answers whether two 2-wave workgroups per CU put one wave on each of the four SIMDs.
//   wave_placement [threads=128] [lds_kib=50] [blocks_per_cu=2]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <map>
#include <set>
#include <vector>

__global__ void where(unsigned* out, int lds_bytes, unsigned long long hold_ticks) {
  extern __shared__ char smem[];
  if (lds_bytes > 0 && threadIdx.x == 0) smem[0] = 1;
  const unsigned long long t0 = wall_clock64();
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  while (wall_clock64() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(8);
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    out[2 * w] = hw;
    out[2 * w + 1] = (xcc & 0xf) | (static_cast<unsigned>(t0 & 0x0fffffff) << 4);  // start time (100 MHz ticks, 28 bits)
  }
}

int main(int argc, char** argv) {
  const int threads = argc > 1 ? atoi(argv[1]) : 128;
  const int lds = (argc > 2 ? atoi(argv[2]) : 50) * 1024;
  const int per_cu = argc > 3 ? atoi(argv[3]) : 2;
  const int blocks = 256 * per_cu, waves = blocks * threads / 64;
  unsigned* d;
  hipMalloc(&d, waves * 2 * sizeof(unsigned));
  hipMemset(d, 0xff, waves * 2 * sizeof(unsigned));
  hipFuncSetAttribute(reinterpret_cast<const void*>(where), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(where, dim3(blocks), dim3(threads), lds, 0, d, lds, 100000ull /* 1 ms at 100 MHz */);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
  std::vector<unsigned> h(waves * 2);
  hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> per_cu_simds;  // key: xcc | se | sh | cu
  for (int w = 0; w < waves; ++w) {
    const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
    const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_cu_simds[(xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(simd);
  }
  std::map<std::string, int> patterns;
  for (auto& kv : per_cu_simds) {
    int cnt[4] = {0, 0, 0, 0};
    for (int s : kv.second) cnt[s]++;
    char buf[64];
    snprintf(buf, sizeof buf, "%d %d %d %d", cnt[0], cnt[1], cnt[2], cnt[3]);
    patterns[buf]++;
  }
  unsigned t_min = ~0u;
  for (int w = 0; w < waves; ++w) t_min = std::min(t_min, h[2 * w + 1] >> 4);
  int late = 0;
  for (int w = 0; w < waves; ++w) late += ((h[2 * w + 1] >> 4) - t_min) > 50000u;  // started more than 0.5 ms after the first: a second round
  printf("%d workgroups x %d threads, %d KiB LDS: %zu CUs seen; %d of %d waves started in a second round; waves per SIMD pattern -> number of CUs\n",
         blocks, threads, lds / 1024, per_cu_simds.size(), late, waves);
  for (auto& kv : patterns) printf("  [%s] : %d\n", kv.first.c_str(), kv.second);
  return 0;
}
