// Host-callable launchers of the gfx950 HIP kernels.  Raw pointers + hipStream_t only, so
// the kernel translation units never see a torch header (they compile in seconds) and the
// launchers can be captured into a hipGraph: none of them allocates, copies or synchronises.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace dpa {
namespace kernels {

enum DType : int { kF32 = 0, kBF16 = 1, kF16 = 2 };

inline int dtype_size(int dt) { return dt == kF32 ? 4 : 2; }

// ------------------------------------------------------------------ bucket pack / unpack
// Multi-tensor "dst[i] = cast(src[i] * scale)".  One launch covers up to kMaxTensors
// tensors; tensor t owns blocks [block_start[t], block_start[t+1]).
struct CopyTable {
  static constexpr int kMaxTensors = 32;
  static constexpr int kChunk = 8192;  // elements per workgroup
  const void* src[kMaxTensors];
  void* dst[kMaxTensors];
  long long numel[kMaxTensors];
  int block_start[kMaxTensors + 1];
  int n;
};
void launch_multi_copy_scale(const CopyTable& t, int src_dtype, int dst_dtype, float scale,
                             hipStream_t stream);
void launch_scale_inplace(void* p, long long n, int dtype, float scale, hipStream_t stream);

// ------------------------------------------------------------------ communication-stream helpers
// A bounded spin of `milliseconds` (one wave, sleeps between clock reads) and a one-lane increment: the pair
// answers "can stream B make progress while stream A has a kernel running".
void launch_spin(float* scratch, double milliseconds, hipStream_t stream);
void launch_touch(float* scratch, hipStream_t stream);
// One-GPU stand-in for a multi-rank collective: `workgroups` workgroups stay resident for `milliseconds` and
// move `bytes` of `buf` `passes` times through HBM (values unchanged), paced over that time.
void launch_emulated_collective(void* buf, long long bytes, int workgroups, int passes, double milliseconds,
                                hipStream_t stream);

// ------------------------------------------------------------------ fused AdamW
// Hyper-parameters live in device memory so a captured graph replays with the current
// learning rate and step: [lr, beta1, beta2, eps, weight_decay, step(as int bits),
// 1-beta1^step, sqrt(1-beta2^step)].
struct AdamWTable {
  static constexpr int kMaxTensors = 24;
  static constexpr int kChunk = 4096;
  void* param[kMaxTensors];
  const void* grad[kMaxTensors];
  float* exp_avg[kMaxTensors];
  float* exp_avg_sq[kMaxTensors];
  float* master[kMaxTensors];  // fp32 master copy for 16-bit params, else nullptr
  long long numel[kMaxTensors];
  int block_start[kMaxTensors + 1];
  int n;
};
constexpr int kAdamWHparamFloats = 8;
void launch_adamw_tick(float* hparams, hipStream_t stream);
void launch_adamw(const AdamWTable& t, int param_dtype, int grad_dtype, const float* hparams,
                  const float* grad_scale /*device scalar or nullptr*/, hipStream_t stream);
// Sum of squares of up to kMaxTensors tensors accumulated into out[0] (fp32, pre-zeroed),
// and the clip coefficient min(1, max_norm / (sqrt(out[0]) + 1e-6)) written by the finisher.
// Writes one partial sum of squares per workgroup (t.block_start[t.n] of them) to `partials`.
void launch_sumsq(const CopyTable& t, int dtype, float* partials, hipStream_t stream);
// out[0] = sum of `n` partials, added up in a fixed order by one workgroup.
void launch_sum_partials(const float* partials, int n, float* out, hipStream_t stream);
void launch_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm_out,
                      hipStream_t stream);

// ------------------------------------------------------------------ softmax cross-entropy
// logits: [N, ld] with the first C columns valid.  Per row: loss (0 for ignored rows), the
// pair (row max, log of the sum of exp(x - max)) in row_lse[2r], row_lse[2r+1], and a correct
// flag (argmax == target, lowest index wins ties).
void launch_xent_fwd(const void* logits, int dtype, const long long* targets, long long N,
                     long long C, long long ld, long long ignore_index, float* row_loss,
                     float* row_lse, unsigned char* correct, hipStream_t stream);
// metrics[0]=sum of row losses, [1]=correct count, [2]=non-ignored rows, [3]=mean loss.
void launch_xent_reduce(const float* row_loss, const unsigned char* correct,
                        const long long* targets, long long N, long long ignore_index,
                        float* metrics, hipStream_t stream);
// dlogits = (softmax - onehot) * grad_out[0] * (mean ? 1/metrics[2] : 1); padded columns
// [C, ld) are written as zero.  dlogits may alias logits.
void launch_xent_bwd(const void* logits, int dtype, const long long* targets, const float* row_lse,
                     const float* grad_out, const float* metrics, int mean, long long N,
                     long long C, long long ld, long long ignore_index, void* dlogits,
                     hipStream_t stream);
// Fused forward + backward for training: `logits` [N, ld] is overwritten with
// (softmax - onehot) * scale, scale = 1/#counted rows (mean) or 1 (sum), written to grad_scale[0].
bool xent_fused_supported(long long C);
void launch_xent_fused(void* logits, int dtype, const long long* targets, long long N, long long C,
                       long long ld, long long ignore_index, int mean, float* grad_scale,
                       float* row_loss, unsigned char* correct, hipStream_t stream);

// ------------------------------------------------------------------ dense layers
enum Act : int { kActNone = 0, kActRelu = 1, kActGelu = 2 };

// C[M,N] = act(A * B^T + bias) style products on the matrix cores.  Layout flags say which
// dimension of each operand is contiguous:  a_kmajor: A is [M,K] row-major (K contiguous),
// else A is stored [K,M];  b_kmajor: B is [N,K] row-major, else B is stored [K,N].
//   forward  y  = x W^T + b : A=x [M,K] kmajor, B=W [N,K] kmajor
//   grad-in  dx = dy W      : A=dy [M,N'] kmajor, B=W stored [K'=N', N=K_in]  (not kmajor)
//   grad-w   dW = dy^T x    : A=dy stored [K'=M, M'=N] (not kmajor), B=x stored [K'=M, N'=K_in]
// pre_act (optional) receives A*B^T + bias before the activation (needed by GELU backward).
struct GemmArgs {
  const void* A;
  const void* B;
  void* C;
  const void* bias;  // [N] or nullptr
  void* pre_act;     // [M,N] or nullptr
  long long M, N, K;
  long long lda, ldb, ldc;
  int a_kmajor, b_kmajor;
  int act;
  float alpha;
  float beta;        // C = alpha*acc + beta*C_old (beta != 0 only with act == none, no bias)
  // Split-K (bf16 only): with split_k > 1 the contraction is cut into split_k ranges computed by
  // separate workgroups into fp32 partials `workspace[split_k][M][N]`, summed in a fixed order by
  // a second kernel that applies the epilogue.  Both are set by the caller from
  // gemm_bf16_split_k(); zero / nullptr means a single pass.
  float* workspace;
  int split_k;
  // Activation-backward epilogue (bf16, 256x256-tile kernel, M and N multiples of 256, no bias /
  // act / pre_act / beta): C = (alpha*acc) * act'(aux) with aux [M,N] at leading dimension ldc --
  // the pre-activation for GELU (aux_mode = kActGelu), the activation output for ReLU
  // (kActRelu).  colsum_partial [2*M/256][N] fp32, when given, receives per-128-row column sums of
  // the bf16-rounded result (the bias gradient of the layer whose activation this is).
  const void* aux;
  int aux_mode;
  float* colsum_partial;
  // Implicit 3x3 convolution, stride 1, zero padding 1, over a channels-last image X that carries a one-pixel
  // zero halo: X[conv_B][conv_H + 2][conv_W + 2][conv_C], conv_C % 64 == 0 (bf16, large-tile kernel only).
  //   conv_mode 1 (forward / input gradient, NT layout): row m of A is output pixel (b, h, w) and its k index runs
  //     over (dh, dw, c):  A[m][(3*dh + dw)*C + c] = X[b][h + dh][w + dw][c]  (dh, dw = 0..2 in halo coordinates).
  //     A points to X[0][0][0][0]; M = B*H*W, K = 9*C; B is the weight as [N][9*C] (a channels-last [N, C, 3, 3]).
  //   conv_mode 2 (weight gradient, split-K TN layout): the contraction runs over ALL haloed pixels q; A is the
  //     haloed output gradient [Q][M] (zero halo), column n = (3*dh + dw)*C + c of B reads X row q + (dh-1)*(W+2) + (dw-1),
  //     channel c.  B points to X row 0; (W + 3) rows before and after it must be readable zeros.  N = 9*C, ldb = C.
  int conv_mode;
  int conv_B, conv_H, conv_W, conv_C;
};
// Number of contraction ranges worth using for this product (1 = do not split): > 1 when the
// output has too few 256x256 tiles to occupy the chip but a long contraction (weight gradients).
int gemm_bf16_split_k(const GemmArgs& a);
bool gemm_bf16_act_backward_supported(const GemmArgs& a);
bool gemm_f32_supported(const GemmArgs& a);
void launch_gemm_f32(const GemmArgs& a, hipStream_t stream);
bool gemm_bf16_supported(const GemmArgs& a);
// Argument check for the implicit-convolution views (GemmArgs::conv_mode 1 / 2); launch_gemm_bf16 runs them.
bool gemm_bf16_conv_supported(const GemmArgs& a);
void launch_gemm_bf16(const GemmArgs& a, hipStream_t stream);
// fp16 storage (fp32 accumulate) on the 128x128 MFMA kernel; same argument conventions.
bool gemm_f16_supported(const GemmArgs& a);
void launch_gemm_f16(const GemmArgs& a, hipStream_t stream);

// dst [rows_total][C] (bf16, 16-byte aligned): row guard_front + q is haloed pixel q of x [B][H][W][C] -- zero on the
// one-pixel frame of every image -- and every row outside [guard_front, guard_front + B*(H+2)*(W+2)) is zero.
void launch_pad_halo(const void* x, void* dst, long long rows_total, long long guard_front, int B, int H, int W, int C,
                     hipStream_t stream);

// Degenerate shapes of the reference MLP (in_dim = 1, n_classes = 4), fp32 or bf16 storage:
//   y[M,N] = act(x[M,1] * w[N,1]^T + b)                     (K = 1 outer product)
void launch_outer_bias_act(const void* x, const void* w, const void* bias, void* y, long long M,
                           long long N, int dtype, int act, hipStream_t stream);
//   y[M,N] = x[M,K] W[N,K]^T + b  with N <= 16               (skinny output)
void launch_skinny_n(const void* x, const void* w, const void* bias, void* y, long long M,
                     long long N, long long K, int dtype, hipStream_t stream);
//   dx[M,K] = dy[M,N] W[N,K]  with N <= 16
void launch_skinny_n_dgrad(const void* dy, const void* w, void* dx, long long M, long long N,
                           long long K, int dtype, hipStream_t stream);
//   dW[N,K] = dy[M,N]^T x[M,K] with N <= 16 (split over M, fp32 partials in `workspace`),
//   db[N] = colsum(dy).  workspace: at least skinny_wgrad_workspace(M,N,K) floats.
long long skinny_wgrad_workspace(long long M, long long N, long long K);
void launch_skinny_n_wgrad(const void* dy, const void* x, void* dW, void* db, float* workspace,
                           long long M, long long N, long long K, int dtype, hipStream_t stream);
//   dw[N] = sum_m dy[m,n] * x[m] ; db[N] = sum_m dy[m,n]      (K = 1 weight grad)
long long colreduce_workspace(long long M, long long N);
void launch_outer_wgrad(const void* dy, const void* x, void* dw, void* db, float* workspace,
                        long long M, long long N, int dtype, hipStream_t stream);
//   db[N] = colsum(dy[M,N])  (also used by the large-GEMM path)
void launch_colsum(const void* dy, void* db, float* workspace, long long M, long long N, int dtype,
                   hipStream_t stream);
// Linear(1, H) -> act -> Linear(H, N <= 16) without materialising the hidden layer (rank1_mlp.hip).
//   forward : y[M,N] = b2 + act(x w1^T + b1) W2^T        x: [M] scalars, w1/b1: [H], W2: [N,H]
//   backward: grads (length rank1_mlp_bwd_len(H,N), NT = 4 | 8 | 16 >= N) holds
//             dW2 rows at [n*H, (n+1)*H), dw1 at [NT*H, ...), db1 at [(NT+1)*H, ...), db2 at [(NT+2)*H, +N);
//             partial: rank1_mlp_bwd_chunks(M,H) * rank1_mlp_bwd_len(H,N) floats of workspace.
//             pk: rank1_mlp_fwd_workspace(H, N) floats (the weights repacked for scalar streaming).
long long rank1_mlp_fwd_workspace(long long H, long long N);
void launch_rank1_mlp_fwd(const void* x, const void* w1, const void* b1, const void* W2, const void* b2, void* y, float* pk,
                          long long M, long long H, long long N, int dtype, int act, hipStream_t stream);
int rank1_mlp_bwd_chunks(long long M, long long H);
long long rank1_mlp_bwd_len(long long H, long long N);
void launch_rank1_mlp_bwd(const void* dy, const void* x, const void* w1, const void* b1, const void* W2, float* partial,
                          void* grads, long long M, long long H, long long N, int dtype, int act, hipStream_t stream);
//   out[i] = sum over c < chunks of partial[c][i]   (fixed order)
// Token + position embedding (bf16): out[r, :] = wte[idx[r], :] + wpe[r % T, :] and its two gradients
// (token ids sorted by the caller, rows of absent tokens left as zeroed by the caller).
bool embedding_supported(long long rows, long long T, long long C, long long V, long long P);
void launch_embedding_fwd(const long long* idx, const void* wte, const void* wpe, void* out, long long rows, int T, int C, long long V,
                          hipStream_t stream);
long long embedding_bwd_chunks(long long rows);
// head_part / body_part: fp32 [embedding_bwd_chunks(rows)][C] scratch, head_tok: int64 [chunks] scratch (long runs of one token).
void launch_embedding_bwd(const void* grad, const long long* tok_sorted, const long long* perm, void* dwte, void* dwpe, long long rows,
                          int B, int T, int C, long long V, float* head_part, float* body_part, long long* head_tok, hipStream_t stream);

void launch_reduce_partials(const float* partial, void* out, long long len, int chunks, int dtype,
                            hipStream_t stream);
//   g = dy * act'(saved)  (+ db = colsum(g)); saved = y for ReLU, pre-activation for GELU.
void launch_act_bwd(const void* dy, const void* saved, void* g, void* db, float* workspace,
                    long long M, long long N, int dtype, int act, hipStream_t stream);

// ------------------------------------------------------------------ layer norm
// With `res` != nullptr: s = x + res (+ res_bias, a [cols] vector) is written to `sum_out` and
// y = LN(s); else y = LN(x).
void launch_layernorm_fwd(const void* x, const void* res, const void* res_bias, void* sum_out,
                          const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                          long long rows, long long cols, float eps, int dtype, hipStream_t stream);
long long layernorm_bwd_workspace(long long rows, long long cols);
// Any row width up to 8192 columns: rows of whole 16-byte vectors that fit a wave's registers take the wave-per-row
// kernels, everything else a workgroup-per-row pair.
bool layernorm_supported(long long cols);
// dx = LN'(dy) (+ dres when given: the gradient reaching the normalised tensor by another path).
// dsum != nullptr: also the column sums of dx (gradient of res_bias); needs cols <= 1365.
void launch_layernorm_bwd(const void* dy, const void* x, const void* dres, const void* gamma,
                          const float* mean, const float* rstd, void* dx, void* dgamma, void* dbeta,
                          void* dsum, float* workspace, long long rows, long long cols, int dtype,
                          hipStream_t stream);

// ------------------------------------------------------------------ batch normalisation (training mode, channels last)
// x, y, residual, dy, dx: row-major [R, C] (R = batch * height * width positions) in `dtype`; gamma, beta and their
// gradients: [C] in `param_dtype`; running statistics: [C] in `stat_dtype`; mean, rstd: fp32 [C].  relu != 0 folds
// y = max(., 0) into the forward and its mask (y > 0) into the backward; a residual is added before the ReLU.
bool batchnorm_supported(long long R, long long C, int dtype);
long long batchnorm_workspace(long long R, long long C, int dtype);  // floats, for either direction
// Statistics of x, y = relu(x * a + b (+ res)); a | b are left in scale_shift [2*C]; gamma / beta / running_* may be null.
void launch_batchnorm_fwd(const void* x, const void* res, void* y, const void* gamma, const void* beta, int param_dtype, void* running_mean,
                          void* running_var, int stat_dtype, float momentum, float eps, float* mean, float* rstd, float* scale_shift,
                          float* workspace, long long R, long long C, int relu, int dtype, hipStream_t stream);
// y = relu(x * scale + shift (+ res)) with given per-channel fp32 coefficients (evaluation mode).
void launch_batchnorm_apply(const void* x, const void* res, void* y, const float* scale, const float* shift, long long R, long long C, int relu,
                            int dtype, hipStream_t stream);
// dx, dgamma, dbeta (either may be null) and dres = masked dy when dres != nullptr.  With relu != 0 the mask is
// y > 0, read from y -- or, when scale_shift (the forward's a | b, [2*C]) is given, recomputed from x: valid when
// no residual was added before the ReLU, and then y is not touched.
void launch_batchnorm_bwd(const void* dy, const void* x, const void* y, const float* scale_shift, const void* gamma, int param_dtype,
                          const float* mean, const float* rstd, void* dx, void* dres, void* dgamma, void* dbeta, float* workspace, long long R,
                          long long C, int relu, int dtype, hipStream_t stream);

// ------------------------------------------------------------------ causal attention (head dim 64, bf16)
// q/k/v element (b, t, h, d) lives at ptr[(b*T + t) * row_stride + h*64 + d] (a packed
// [B, T, 3, H, 64] projection has row_stride = 3*H*64 and k = q + H*64, v = q + 2*H*64).
// o / do are [B, T, H, 64] contiguous; lse is [B, H, T] fp32; delta is a workspace of 2*B*H*T floats.
bool attention_supported(long long T, long long head_dim, long long row_stride);
// `kernel`: -1 = choose by shape (environment DPA_ATTN_FWD=classic|wide|tile32 overrides), 0 = the first kernel (32 queries
// per wave, 16x16x32 matrix instructions), 1 = 64 queries per wave on 32x32x16 (needs B*H even; falls back otherwise),
// 2 = 32 queries per wave on 32x32x16.
void launch_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                          long long row_stride, int B, int H, int T, float softmax_scale,
                          hipStream_t stream, int kernel = -1);
void launch_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                          const float* lse, float* delta, void* dq, void* dk, void* dv,
                          long long row_stride, long long grad_row_stride, int B, int H, int T,
                          float softmax_scale, float* bias_partial, hipStream_t stream, int kernel = -1);
// kernel: -1 = choose (environment DPA_ATTN_BWD=classic|wide overrides), 0 = 16x16x32 matrix instructions, 1 = 32x32x16.
// bias_partial: nullptr, or [B*T/128][3*H*64] fp32 receiving the column sums of dq | dk | dv per block
// of 128 positions (summed over the first index they give the bias gradient of the q|k|v projection).

}  // namespace kernels
}  // namespace dpa
