// Token + position embedding of a GPT-style model (bf16), forward and backward.
//
// The reference model has no embedding (two linear layers, reference min_DDP.py:41-49); this op serves
// the GPT-2 benchmark configuration, whose first layer is  x[b, t, :] = wte[idx[b, t], :] + wpe[t, :].
//   * forward: one pass, 16-byte vectors, the sum is taken in fp32 and rounded once (what the two stock
//     operators -- gather, then a broadcast add over the [B, T, C] tensor -- do in two passes);
//   * position gradient: dwpe[t, :] = sum over b of g[b, t, :], added up in the fixed order b = 0, 1, ...;
//   * token gradient: the token ids arrive sorted (stable) together with the rows they came from; the
//     workgroup that finds itself at the first occurrence of a token adds that token's rows up in sorted
//     order and writes the finished row, every other workgroup leaves at once.  No atomics: the result
//     does not depend on scheduling.  Rows of tokens that do not occur are left as the caller zeroed them.
// Token ids outside [0, V) are clamped in the forward and skipped in the backward; callers that want an
// error for them check on the host (ops/embedding.py).
#include "device_utils.h"
#include "kernels.h"

namespace dpa {
namespace kernels {

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ void unpack8(const u32x4& v, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v[i] << 16);
    f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
  }
}

__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    v[i] = static_cast<unsigned>(BF16::from_f32(f[2 * i])) | (static_cast<unsigned>(BF16::from_f32(f[2 * i + 1])) << 16);
  return v;
}

constexpr int kEmbedThreads = 256;

__global__ __launch_bounds__(kEmbedThreads) void embed_fwd_kernel(const long long* __restrict__ idx, const u32x4* __restrict__ wte,
                                                                  const u32x4* __restrict__ wpe, u32x4* __restrict__ out,
                                                                  long long rows, int T, int vpr, long long V) {
  const long long total = rows * vpr;
  const long long step = static_cast<long long>(gridDim.x) * kEmbedThreads;
  for (long long v = static_cast<long long>(blockIdx.x) * kEmbedThreads + threadIdx.x; v < total; v += step) {
    const long long r = v / vpr;
    const int c = static_cast<int>(v - r * vpr);
    long long tok = idx[r];
    tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
    const int pos = static_cast<int>(r % T);
    float a[8], b[8];
    unpack8(wte[tok * vpr + c], a);
    unpack8(wpe[static_cast<long long>(pos) * vpr + c], b);
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] += b[i];
    out[v] = pack8(a);
  }
}

// One thread per (position, 8-column vector); the batch is walked in order with eight loads in flight.
__global__ __launch_bounds__(kEmbedThreads) void embed_bwd_pos_kernel(const u32x4* __restrict__ g, u32x4* __restrict__ dwpe, int B,
                                                                      long long per_batch /* T * vpr */) {
  const long long v = static_cast<long long>(blockIdx.x) * kEmbedThreads + threadIdx.x;
  if (v >= per_batch) return;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int b = 0; b < B; ++b) {
    float x[8];
    unpack8(g[static_cast<long long>(b) * per_batch + v], x);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += x[i];
  }
  dwpe[v] = pack8(acc);
}

constexpr int kTokThreads = 128;
constexpr int kRunChunk = 256;  // sorted positions per chunk: no workgroup walks more occurrences than this

// Token rows.  The occurrences of a token are consecutive in the stable-sorted id list.  The list is cut at every
// multiple of kRunChunk, and a workgroup sums the piece of a run that starts at its position i -- i is a run's first
// occurrence or a chunk boundary inside a run -- up to the run's end or the next boundary, whichever comes first.
// A run that ends inside the chunk it started in (every run of a near-uniform id distribution) is finished by that
// one workgroup, as a single fixed-order sum.  A longer run (a padding or very frequent token: 10^4 .. 10^5
// occurrences in a batch of real text) leaves fp32 partial rows instead -- head_part[c] for the piece from its first
// occurrence to the end of chunk c, body_part[c] for each later chunk -- and embed_bwd_tok_join_kernel adds them
// up in chunk order.  Which positions belong to which sum is a function of the sorted list alone: deterministic.
__global__ __launch_bounds__(kTokThreads) void embed_bwd_tok_kernel(const long long* __restrict__ tok_sorted, const long long* __restrict__ perm,
                                                                    const u32x4* __restrict__ g, u32x4* __restrict__ dwte, long long n,
                                                                    int vpr, long long V, float* __restrict__ head_part,
                                                                    float* __restrict__ body_part, long long* __restrict__ head_tok) {
  const long long i = blockIdx.x;
  const long long tok = tok_sorted[i];
  const bool is_head = i == 0 || tok_sorted[i - 1] != tok;
  const bool is_boundary = (i % kRunChunk) == 0;
  if (!is_head && !is_boundary) return;
  if (tok < 0 || tok >= V) return;
  const long long c = i / kRunChunk;
  const long long stop = (c + 1) * kRunChunk < n ? (c + 1) * kRunChunk : n;  // next boundary
  const bool continues = stop < n && tok_sorted[stop] == tok && tok_sorted[stop - 1] == tok;
  const bool whole = is_head && !continues;  // the run ends before the next boundary: this sum is final
  if (threadIdx.x == 0 && is_head && continues) head_tok[c] = tok;
  for (int col = threadIdx.x; col < vpr; col += kTokThreads) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long long j = i; j < stop && tok_sorted[j] == tok; ++j) {
      float x[8];
      unpack8(g[perm[j] * vpr + col], x);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += x[e];
    }
    if (whole) {
      dwte[tok * vpr + col] = pack8(acc);
    } else {
      float* dst = (is_head ? head_part : body_part) + (c * vpr + col) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) dst[e] = acc[e];
    }
  }
}

// One workgroup per chunk c in which a long run starts (head_tok[c] >= 0): head piece, then the body pieces of the
// following chunks while they still belong to the run, in chunk order.
__global__ __launch_bounds__(kTokThreads) void embed_bwd_tok_join_kernel(const long long* __restrict__ tok_sorted, u32x4* __restrict__ dwte,
                                                                         long long n, int vpr, const float* __restrict__ head_part,
                                                                         const float* __restrict__ body_part,
                                                                         const long long* __restrict__ head_tok) {
  const long long c0 = blockIdx.x;
  const long long tok = head_tok[c0];
  if (tok < 0) return;
  for (int col = threadIdx.x; col < vpr; col += kTokThreads) {
    float acc[8];
    const float* hp = head_part + (c0 * vpr + col) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = hp[e];
    for (long long c = c0 + 1; c * kRunChunk < n && tok_sorted[c * kRunChunk] == tok; ++c) {
      const float* bp = body_part + (c * vpr + col) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += bp[e];
    }
    dwte[tok * vpr + col] = pack8(acc);
  }
}

}  // namespace

bool embedding_supported(long long rows, long long T, long long C, long long V, long long P) {
  return rows > 0 && T > 0 && rows % T == 0 && T <= P && C > 0 && C % 8 == 0 && V > 0 && rows < (1LL << 31) && rows * (C / 8) < (1LL << 40);
}

void launch_embedding_fwd(const long long* idx, const void* wte, const void* wpe, void* out, long long rows, int T, int C, long long V,
                          hipStream_t stream) {
  const int vpr = C / 8;
  const long long total = rows * vpr;
  long long blocks = (total + kEmbedThreads - 1) / kEmbedThreads;
  if (blocks > 256LL * 64) blocks = 256LL * 64;  // grid-stride beyond 64 workgroups per CU
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kEmbedThreads), 0, stream, idx,
                     static_cast<const u32x4*>(wte), static_cast<const u32x4*>(wpe), static_cast<u32x4*>(out), rows, T, vpr, V);
}

long long embedding_bwd_chunks(long long rows) { return (rows + kRunChunk - 1) / kRunChunk; }

// head_part / body_part: fp32 [chunks][C] each; head_tok: int64 [chunks] (embedding_bwd_chunks(rows) chunks).
void launch_embedding_bwd(const void* grad, const long long* tok_sorted, const long long* perm, void* dwte, void* dwpe, long long rows,
                          int B, int T, int C, long long V, float* head_part, float* body_part, long long* head_tok, hipStream_t stream) {
  const int vpr = C / 8;
  const long long per_batch = static_cast<long long>(T) * vpr;
  hipLaunchKernelGGL(embed_bwd_pos_kernel, dim3(static_cast<unsigned>((per_batch + kEmbedThreads - 1) / kEmbedThreads)), dim3(kEmbedThreads),
                     0, stream, static_cast<const u32x4*>(grad), static_cast<u32x4*>(dwpe), B, per_batch);
  const long long chunks = embedding_bwd_chunks(rows);
  (void)hipMemsetAsync(head_tok, 0xFF, static_cast<size_t>(chunks) * sizeof(long long), stream);  // -1: no long run starts in the chunk
  hipLaunchKernelGGL(embed_bwd_tok_kernel, dim3(static_cast<unsigned>(rows)), dim3(kTokThreads), 0, stream, tok_sorted, perm,
                     static_cast<const u32x4*>(grad), static_cast<u32x4*>(dwte), rows, vpr, V, head_part, body_part, head_tok);
  hipLaunchKernelGGL(embed_bwd_tok_join_kernel, dim3(static_cast<unsigned>(chunks)), dim3(kTokThreads), 0, stream, tok_sorted,
                     static_cast<u32x4*>(dwte), rows, vpr, head_part, body_part, head_tok);
}

}  // namespace kernels
}  // namespace dpa
