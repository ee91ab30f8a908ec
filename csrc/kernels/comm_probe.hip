// Small kernels around the communication stream: a bounded spin and a touch (used to find out whether two
// streams can make progress independently), and a stand-in for an N-rank all-reduce on ONE GPU.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace dpa {
namespace kernels {

namespace {

// wall_clock64() is the 100 MHz constant clock: the same rate under any power state.
__global__ void spin_kernel(float* p, long long ticks) {
  const long long t0 = wall_clock64();
  int guard = 0;
  while (wall_clock64() - t0 < ticks && guard < (1 << 27)) {  // bounded: every wave leaves
    __builtin_amdgcn_s_sleep(8);
    ++guard;
  }
  if (threadIdx.x == 0) p[0] = static_cast<float>(guard);
}

__global__ void touch_kernel(float* p) {
  if (threadIdx.x == 0) p[0] += 1.0f;
}

// One-GPU stand-in for a collective over `bytes` of `buf`: `gridDim.x` workgroups (the CU footprint of a
// collective kernel) stay resident for `total_ticks` and move the buffer `passes` times through the chip
// (16-byte loads and stores per lane, values unchanged), paced so the traffic is spread over the whole time
// like the chunks of a transfer that waits for the links.  Nothing here measures a link.
__global__ void __launch_bounds__(256) emulated_collective_kernel(uint4* buf, long long n_vec, int passes,
                                                                  long long total_ticks) {
  const long long t0 = wall_clock64();
  const long long per_wg = (n_vec + gridDim.x - 1) / gridDim.x;
  const long long lo = per_wg * blockIdx.x;
  const long long hi = lo + per_wg < n_vec ? lo + per_wg : n_vec;
  constexpr long long kChunkVec = 256 * 16;  // 64 KiB per workgroup step
  const long long n_chunks = hi > lo ? (hi - lo + kChunkVec - 1) / kChunkVec : 0;
  const long long steps = n_chunks * passes;
  long long step = 0;
  for (int pass = 0; pass < passes; ++pass) {
    for (long long c = 0; c < n_chunks; ++c, ++step) {
      // chunk `step` may not start before its share of the time has passed
      const long long due = steps > 0 ? total_ticks * step / steps : 0;
      int guard = 0;
      while (wall_clock64() - t0 < due && guard < (1 << 24)) {
        __builtin_amdgcn_s_sleep(16);
        ++guard;
      }
      const long long base = lo + c * kChunkVec;
#pragma unroll 4
      for (int k = 0; k < 16; ++k) {
        const long long i = base + k * 256 + threadIdx.x;
        if (i < hi) {
          uint4 v = buf[i];
          buf[i] = v;
        }
      }
    }
  }
  int guard = 0;
  while (wall_clock64() - t0 < total_ticks && guard < (1 << 24)) {
    __builtin_amdgcn_s_sleep(16);
    ++guard;
  }
}

}  // namespace

void launch_spin(float* scratch, double milliseconds, hipStream_t stream) {
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, stream, scratch, static_cast<long long>(milliseconds * 1e5));
}

void launch_touch(float* scratch, hipStream_t stream) {
  hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, stream, scratch);
}

void launch_emulated_collective(void* buf, long long bytes, int workgroups, int passes, double milliseconds,
                                hipStream_t stream) {
  const long long n_vec = bytes / 16;
  if (workgroups < 1) workgroups = 1;
  hipLaunchKernelGGL(emulated_collective_kernel, dim3(workgroups), dim3(256), 0, stream, static_cast<uint4*>(buf), n_vec,
                     passes, static_cast<long long>(milliseconds * 1e5));
}

}  // namespace kernels
}  // namespace dpa
