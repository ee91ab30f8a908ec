// Which XCD does workgroup b of a launch run on?  Every workgroup records the XCC_ID hardware register.
//   hipcc --offload-arch=gfx950 -O3 xcd_map_probe.hip -o xcd_map_probe ; ./xcd_map_probe [workgroups] [threads] [lds_kib]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void probe(unsigned* out, int spin) {
  extern __shared__ char lds[];
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (threadIdx.x == 0) { lds[0] = 1; out[blockIdx.x] = id; }
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(100);  // keep the workgroups resident together
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 256, threads = argc > 2 ? atoi(argv[2]) : 512, lds = (argc > 3 ? atoi(argv[3]) : 150) * 1024;
  unsigned* d;
  if (hipMalloc(&d, n * 4) != hipSuccess) return 1;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(probe, dim3(n), dim3(threads), lds, 0, d, 200);
    std::vector<unsigned> h(n);
    if (hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    int match = 0;
    for (int b = 0; b < n; ++b) match += int((h[b] & 0xf) == unsigned(b & 7));
    printf("launch %d: %d of %d workgroups on XCD (b %% 8); first 24 ids:", rep, match, n);
    for (int b = 0; b < 24 && b < n; ++b) printf(" %u", h[b] & 0xf);
    printf("  (raw register of b=0: 0x%x)\n", h[0]);
  }
  return 0;
}
