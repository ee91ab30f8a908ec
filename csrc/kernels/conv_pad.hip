// Zero-haloed copy of a channels-last image for the implicit 3x3 convolution (GemmArgs::conv_mode):
// X[b][1 + h][1 + w][c] = x[b][h][w][c], zero on the one-pixel frame of every image and on the guard rows in front of
// and behind the whole batch.  One pass: every 16-byte piece of the destination is either copied or zeroed.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace dpa {
namespace kernels {

namespace {

// dst row r (of `rows_total`, each `chunks` 16-byte pieces) is haloed pixel q = r - guard_front.
__global__ __launch_bounds__(256) void pad_halo_kernel(const uint4* __restrict__ x, uint4* __restrict__ dst, long long rows_total,
                                                       long long guard_front, int B, int H, int W, int chunks) {
  const long long idx = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= rows_total * chunks) return;
  const long long r = idx / chunks;
  const int ch = static_cast<int>(idx - r * chunks);
  const long long q = r - guard_front;
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  const long long per_image = static_cast<long long>(H + 2) * (W + 2);
  if (q >= 0 && q < per_image * B) {
    const long long b = q / per_image;
    const int rem = static_cast<int>(q - b * per_image);
    const int hh = rem / (W + 2), ww = rem - hh * (W + 2);
    if (hh >= 1 && hh <= H && ww >= 1 && ww <= W) v = x[((b * H + (hh - 1)) * W + (ww - 1)) * chunks + ch];
  }
  dst[idx] = v;
}

}  // namespace

void launch_pad_halo(const void* x, void* dst, long long rows_total, long long guard_front, int B, int H, int W, int C,
                     hipStream_t stream) {
  const int chunks = C / 8;  // bf16: 8 elements per 16 bytes
  const long long n = rows_total * chunks;
  hipLaunchKernelGGL(pad_halo_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, static_cast<const uint4*>(x),
                     static_cast<uint4*>(dst), rows_total, guard_front, B, H, W, chunks);
}

}  // namespace kernels
}  // namespace dpa
