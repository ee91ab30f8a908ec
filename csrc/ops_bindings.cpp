// Tensor-level entry points of the HIP kernels.  Every function validates shapes, dtypes,
// devices and contiguity on the host before a launch, allocates outputs through torch's
// caching allocator, and launches on torch's current HIP stream.
#include <c10/hip/HIPStream.h>
#include <torch/extension.h>

#include <tuple>

#include "kernels/kernels.h"

namespace py = pybind11;
using namespace dpa;
using at::Tensor;

namespace {

int kdtype(const Tensor& t) {
  switch (t.scalar_type()) {
    case at::kFloat: return kernels::kF32;
    case at::kBFloat16: return kernels::kBF16;
    case at::kHalf: return kernels::kF16;
    default: TORCH_CHECK(false, "unsupported dtype ", t.scalar_type(), " (float32 / bfloat16 / float16 expected)");
  }
}

hipStream_t stream_of(const Tensor& t) {
  return c10::hip::getCurrentHIPStream(static_cast<c10::DeviceIndex>(t.get_device())).stream();
}

void check_gpu(const Tensor& t, const char* name) {
  TORCH_CHECK(t.defined(), name, " is undefined");
  TORCH_CHECK(t.is_cuda(), name, " must be a GPU tensor");
  TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}

struct DevGuard {
  int prev = -1;
  explicit DevGuard(const Tensor& t) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != t.get_device()) {
      prev = cur;
      (void)hipSetDevice(t.get_device());
    }
  }
  ~DevGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// ------------------------------------------------------------------ pack / scale
void multi_copy_scale(const std::vector<Tensor>& srcs, const std::vector<Tensor>& dsts, double scale) {
  TORCH_CHECK(srcs.size() == dsts.size(), "multi_copy_scale: list sizes differ");
  if (srcs.empty()) return;
  DevGuard g(srcs[0]);
  hipStream_t stream = stream_of(srcs[0]);
  for (int sd = 0; sd < 3; ++sd)
    for (int dd = 0; dd < 3; ++dd) {
      kernels::CopyTable tab;
      tab.n = 0;
      tab.block_start[0] = 0;
      auto flush = [&]() {
        if (tab.n > 0) kernels::launch_multi_copy_scale(tab, sd, dd, static_cast<float>(scale), stream);
        tab.n = 0;
        tab.block_start[0] = 0;
      };
      for (size_t i = 0; i < srcs.size(); ++i) {
        check_gpu(srcs[i], "src");
        check_gpu(dsts[i], "dst");
        TORCH_CHECK(srcs[i].numel() == dsts[i].numel(), "multi_copy_scale: element counts differ at index ", i);
        TORCH_CHECK(srcs[i].get_device() == dsts[i].get_device(), "multi_copy_scale: device mismatch at index ", i);
        if (kdtype(srcs[i]) != sd || kdtype(dsts[i]) != dd || srcs[i].numel() == 0) continue;
        if (tab.n == kernels::CopyTable::kMaxTensors) flush();
        const int k = tab.n;
        tab.src[k] = srcs[i].data_ptr();
        tab.dst[k] = dsts[i].data_ptr();
        tab.numel[k] = srcs[i].numel();
        tab.block_start[k + 1] = tab.block_start[k] + static_cast<int>((srcs[i].numel() + kernels::CopyTable::kChunk - 1) / kernels::CopyTable::kChunk);
        tab.n = k + 1;
      }
      flush();
    }
}

void scale_inplace(Tensor t, double scale) {
  check_gpu(t, "tensor");
  DevGuard g(t);
  kernels::launch_scale_inplace(t.data_ptr(), t.numel(), kdtype(t), static_cast<float>(scale), stream_of(t));
}

// ------------------------------------------------------------------ AdamW
void adamw_tick(Tensor hparams) {
  check_gpu(hparams, "hparams");
  TORCH_CHECK(hparams.scalar_type() == at::kFloat && hparams.numel() >= kernels::kAdamWHparamFloats, "hparams: float32[8] expected");
  DevGuard g(hparams);
  kernels::launch_adamw_tick(hparams.data_ptr<float>(), stream_of(hparams));
}

void adamw_step(const std::vector<Tensor>& params, const std::vector<Tensor>& grads, const std::vector<Tensor>& exp_avgs,
                const std::vector<Tensor>& exp_avg_sqs, const std::vector<c10::optional<Tensor>>& masters, Tensor hparams,
                c10::optional<Tensor> grad_scale) {
  const size_t n = params.size();
  TORCH_CHECK(grads.size() == n && exp_avgs.size() == n && exp_avg_sqs.size() == n && masters.size() == n,
              "adamw_step: list sizes differ");
  if (n == 0) return;
  check_gpu(hparams, "hparams");
  TORCH_CHECK(hparams.scalar_type() == at::kFloat && hparams.numel() >= kernels::kAdamWHparamFloats, "hparams: float32[8] expected");
  const float* gs = nullptr;
  if (grad_scale.has_value() && grad_scale->defined()) {
    check_gpu(*grad_scale, "grad_scale");
    TORCH_CHECK(grad_scale->scalar_type() == at::kFloat && grad_scale->numel() >= 1, "grad_scale: float32 scalar expected");
    gs = grad_scale->data_ptr<float>();
  }
  DevGuard g(params[0]);
  hipStream_t stream = stream_of(params[0]);
  for (int pd = 0; pd < 3; ++pd)
    for (int gd = 0; gd < 3; ++gd) {
      kernels::AdamWTable tab;
      tab.n = 0;
      tab.block_start[0] = 0;
      auto flush = [&]() {
        if (tab.n > 0) kernels::launch_adamw(tab, pd, gd, hparams.data_ptr<float>(), gs, stream);
        tab.n = 0;
        tab.block_start[0] = 0;
      };
      for (size_t i = 0; i < n; ++i) {
        check_gpu(params[i], "param");
        check_gpu(grads[i], "grad");
        check_gpu(exp_avgs[i], "exp_avg");
        check_gpu(exp_avg_sqs[i], "exp_avg_sq");
        const int64_t ne = params[i].numel();
        TORCH_CHECK(grads[i].numel() == ne && exp_avgs[i].numel() == ne && exp_avg_sqs[i].numel() == ne,
                    "adamw_step: element counts differ at index ", i);
        TORCH_CHECK(exp_avgs[i].scalar_type() == at::kFloat && exp_avg_sqs[i].scalar_type() == at::kFloat,
                    "adamw_step: moments must be float32");
        TORCH_CHECK(grads[i].get_device() == params[i].get_device() && exp_avgs[i].get_device() == params[i].get_device() &&
                        exp_avg_sqs[i].get_device() == params[i].get_device(), "adamw_step: device mismatch at index ", i);
        if (kdtype(params[i]) != pd || kdtype(grads[i]) != gd || ne == 0) continue;
        float* master = nullptr;
        if (masters[i].has_value() && masters[i]->defined()) {
          check_gpu(*masters[i], "master");
          TORCH_CHECK(masters[i]->scalar_type() == at::kFloat && masters[i]->numel() == ne, "adamw_step: master must be float32 of the same size");
          master = masters[i]->data_ptr<float>();
        }
        if (tab.n == kernels::AdamWTable::kMaxTensors) flush();
        const int k = tab.n;
        tab.param[k] = params[i].data_ptr();
        tab.grad[k] = grads[i].data_ptr();
        tab.exp_avg[k] = exp_avgs[i].data_ptr<float>();
        tab.exp_avg_sq[k] = exp_avg_sqs[i].data_ptr<float>();
        tab.master[k] = master;
        tab.numel[k] = ne;
        tab.block_start[k + 1] = tab.block_start[k] + static_cast<int>((ne + kernels::AdamWTable::kChunk - 1) / kernels::AdamWTable::kChunk);
        tab.n = k + 1;
      }
      flush();
    }
}

// out[0] = sum of squares over all tensors.  Two stages with a fixed summation order: every
// workgroup writes its partial into a scratch vector, one workgroup adds the partials up.
void grad_sumsq(const std::vector<Tensor>& tensors, Tensor out) {
  check_gpu(out, "out");
  TORCH_CHECK(out.scalar_type() == at::kFloat && out.numel() >= 1, "grad_sumsq: float32 accumulator expected");
  if (tensors.empty()) return;
  DevGuard g(out);
  hipStream_t stream = stream_of(out);
  int64_t total_blocks = 0;
  for (const Tensor& t : tensors) {
    check_gpu(t, "tensor");
    TORCH_CHECK(t.get_device() == out.get_device(), "grad_sumsq: device mismatch");
    TORCH_CHECK(kdtype(t) >= 0, "grad_sumsq: unsupported dtype");
    total_blocks += (t.numel() + kernels::CopyTable::kChunk - 1) / kernels::CopyTable::kChunk;
  }
  if (total_blocks == 0) {
    out.zero_();
    return;
  }
  TORCH_CHECK(total_blocks < (1ll << 31), "grad_sumsq: too many elements");
  Tensor partials = at::empty({total_blocks}, out.options());
  float* pp = partials.data_ptr<float>();
  int64_t done = 0;
  for (int dt = 0; dt < 3; ++dt) {
    kernels::CopyTable tab;
    tab.n = 0;
    tab.block_start[0] = 0;
    auto flush = [&]() {
      if (tab.n > 0) {
        kernels::launch_sumsq(tab, dt, pp + done, stream);
        done += tab.block_start[tab.n];
      }
      tab.n = 0;
      tab.block_start[0] = 0;
    };
    for (const Tensor& t : tensors) {
      if (kdtype(t) != dt || t.numel() == 0) continue;
      if (tab.n == kernels::CopyTable::kMaxTensors) flush();
      const int k = tab.n;
      tab.src[k] = t.data_ptr();
      tab.dst[k] = nullptr;
      tab.numel[k] = t.numel();
      tab.block_start[k + 1] = tab.block_start[k] + static_cast<int>((t.numel() + kernels::CopyTable::kChunk - 1) / kernels::CopyTable::kChunk);
      tab.n = k + 1;
    }
    flush();
  }
  kernels::launch_sum_partials(pp, static_cast<int>(total_blocks), out.data_ptr<float>(), stream);
}

void clip_coef(Tensor sumsq, double max_norm, Tensor coef, Tensor norm_out) {
  check_gpu(sumsq, "sumsq");
  check_gpu(coef, "coef");
  check_gpu(norm_out, "norm_out");
  TORCH_CHECK(sumsq.scalar_type() == at::kFloat && coef.scalar_type() == at::kFloat && norm_out.scalar_type() == at::kFloat,
              "clip_coef: float32 scalars expected");
  DevGuard g(sumsq);
  kernels::launch_clip_coef(sumsq.data_ptr<float>(), static_cast<float>(max_norm), coef.data_ptr<float>(), norm_out.data_ptr<float>(), stream_of(sumsq));
}

// ------------------------------------------------------------------ softmax cross-entropy
// logits: [N, ld]; only the first n_classes columns are real (n_classes <= 0 means all).
std::tuple<Tensor, Tensor, Tensor, Tensor> xent_forward(Tensor logits, Tensor targets, int64_t n_classes, int64_t ignore_index) {
  check_gpu(logits, "logits");
  check_gpu(targets, "targets");
  TORCH_CHECK(logits.dim() == 2, "logits must be [N, C]");
  TORCH_CHECK(targets.dim() == 1 && targets.size(0) == logits.size(0), "targets must be [N]");
  TORCH_CHECK(targets.scalar_type() == at::kLong, "targets must be int64");
  TORCH_CHECK(targets.get_device() == logits.get_device(), "logits and targets on different devices");
  const int64_t N = logits.size(0), ld = logits.size(1);
  const int64_t C = n_classes > 0 ? n_classes : ld;
  TORCH_CHECK(C <= ld, "n_classes exceeds the logits width");
  TORCH_CHECK(N < (1ll << 31), "too many rows");
  auto fopt = logits.options().dtype(at::kFloat);
  Tensor row_loss = at::empty({N}, fopt), row_lse = at::empty({N, 2}, fopt);
  Tensor correct = at::empty({N}, logits.options().dtype(at::kBool));
  Tensor metrics = at::empty({4}, fopt);
  DevGuard g(logits);
  hipStream_t s = stream_of(logits);
  kernels::launch_xent_fwd(logits.data_ptr(), kdtype(logits), reinterpret_cast<const long long*>(targets.data_ptr<int64_t>()), N, C, ld,
                           ignore_index, row_loss.data_ptr<float>(), row_lse.data_ptr<float>(),
                           reinterpret_cast<unsigned char*>(correct.data_ptr<bool>()), s);
  kernels::launch_xent_reduce(row_loss.data_ptr<float>(), reinterpret_cast<const unsigned char*>(correct.data_ptr<bool>()),
                              reinterpret_cast<const long long*>(targets.data_ptr<int64_t>()), N, ignore_index,
                              metrics.data_ptr<float>(), s);
  return {row_loss, row_lse, correct, metrics};
}

Tensor xent_backward(Tensor logits, Tensor targets, Tensor row_lse, Tensor grad_out, Tensor metrics, bool mean,
                     int64_t n_classes, int64_t ignore_index, bool inplace) {
  check_gpu(logits, "logits");
  check_gpu(targets, "targets");
  check_gpu(row_lse, "row_lse");
  check_gpu(grad_out, "grad_out");
  check_gpu(metrics, "metrics");
  TORCH_CHECK(logits.dim() == 2 && targets.numel() == logits.size(0) && row_lse.numel() == 2 * logits.size(0), "xent_backward: shape mismatch");
  TORCH_CHECK(targets.scalar_type() == at::kLong && row_lse.scalar_type() == at::kFloat && metrics.scalar_type() == at::kFloat &&
                  metrics.numel() >= 4, "xent_backward: dtype mismatch");
  TORCH_CHECK(grad_out.scalar_type() == at::kFloat && grad_out.numel() == 1, "xent_backward: grad_out must be a float32 scalar");
  const int64_t N = logits.size(0), ld = logits.size(1);
  const int64_t C = n_classes > 0 ? n_classes : ld;
  TORCH_CHECK(C <= ld, "n_classes exceeds the logits width");
  Tensor dlogits = inplace ? logits : at::empty_like(logits);
  DevGuard g(logits);
  kernels::launch_xent_bwd(logits.data_ptr(), kdtype(logits), reinterpret_cast<const long long*>(targets.data_ptr<int64_t>()),
                           row_lse.data_ptr<float>(), grad_out.data_ptr<float>(), metrics.data_ptr<float>(), mean ? 1 : 0, N, C, ld,
                           ignore_index, dlogits.data_ptr(), stream_of(logits));
  return dlogits;
}

// Training-only fused pass: overwrites `logits` with d(loss)/d(logits) for a unit upstream gradient
// (mean or sum reduction) and returns (correct, metrics); metrics = {loss sum, #correct, #counted, mean loss}.
std::tuple<Tensor, Tensor> xent_fused(Tensor logits, Tensor targets, int64_t n_classes, int64_t ignore_index, bool mean) {
  check_gpu(logits, "logits");
  check_gpu(targets, "targets");
  TORCH_CHECK(logits.dim() == 2 && logits.is_contiguous(), "xent_fused: logits must be a contiguous [N, C] tensor");
  TORCH_CHECK(targets.dim() == 1 && targets.size(0) == logits.size(0) && targets.scalar_type() == at::kLong, "xent_fused: targets must be int64 [N]");
  TORCH_CHECK(targets.get_device() == logits.get_device(), "logits and targets on different devices");
  const int64_t N = logits.size(0), ld = logits.size(1);
  const int64_t C = n_classes > 0 ? n_classes : ld;
  TORCH_CHECK(C <= ld && N < (1ll << 31), "xent_fused: bad sizes");
  TORCH_CHECK(kernels::xent_fused_supported(C), "xent_fused: needs more than 64 classes");
  auto fopt = logits.options().dtype(at::kFloat);
  Tensor row_loss = at::empty({N}, fopt), scale = at::empty({1}, fopt), metrics = at::empty({4}, fopt);
  Tensor correct = at::empty({N}, logits.options().dtype(at::kBool));
  DevGuard g(logits);
  hipStream_t s = stream_of(logits);
  const long long* tg = reinterpret_cast<const long long*>(targets.data_ptr<int64_t>());
  kernels::launch_xent_fused(logits.data_ptr(), kdtype(logits), tg, N, C, ld, ignore_index, mean ? 1 : 0, scale.data_ptr<float>(),
                             row_loss.data_ptr<float>(), reinterpret_cast<unsigned char*>(correct.data_ptr<bool>()), s);
  kernels::launch_xent_reduce(row_loss.data_ptr<float>(), reinterpret_cast<const unsigned char*>(correct.data_ptr<bool>()), tg, N,
                              ignore_index, metrics.data_ptr<float>(), s);
  return {correct, metrics};
}

// ------------------------------------------------------------------ batch normalisation (training mode, channels last)
// x: [R, C] (the channels-last activation flattened over batch and space).  Returns (y, mean, rstd, scale | shift).
// running_mean / running_var are updated in place when given.
std::tuple<Tensor, Tensor, Tensor, Tensor> batchnorm_forward(Tensor x, c10::optional<Tensor> gamma, c10::optional<Tensor> beta,
                                                     c10::optional<Tensor> running_mean, c10::optional<Tensor> running_var, double momentum,
                                                     double eps, c10::optional<Tensor> residual, bool relu) {
  check_gpu(x, "x");
  TORCH_CHECK(x.dim() == 2 && x.is_contiguous(), "batchnorm_forward: x must be a contiguous [R, C] tensor");
  const int64_t R = x.size(0), C = x.size(1);
  const int dt = kdtype(x);
  TORCH_CHECK(kernels::batchnorm_supported(R, C, dt), "batchnorm_forward: unsupported shape [", R, ", ", C, "]");
  int pdt = kernels::kF32, sdt = kernels::kF32;
  const void *gp = nullptr, *bp = nullptr;
  if (gamma.has_value() && gamma->defined()) {
    check_gpu(*gamma, "weight");
    TORCH_CHECK(gamma->numel() == C && gamma->is_contiguous(), "batchnorm_forward: weight must be [C]");
    pdt = kdtype(*gamma);
    gp = gamma->data_ptr();
  }
  if (beta.has_value() && beta->defined()) {
    check_gpu(*beta, "bias");
    TORCH_CHECK(beta->numel() == C && beta->is_contiguous(), "batchnorm_forward: bias must be [C]");
    TORCH_CHECK(gp == nullptr || kdtype(*beta) == pdt, "batchnorm_forward: weight and bias dtypes differ");
    pdt = kdtype(*beta);
    bp = beta->data_ptr();
  }
  void *rm = nullptr, *rv = nullptr;
  if (running_mean.has_value() && running_mean->defined()) {
    check_gpu(*running_mean, "running_mean");
    TORCH_CHECK(running_var.has_value() && running_var->defined(), "batchnorm_forward: running_mean without running_var");
    check_gpu(*running_var, "running_var");
    TORCH_CHECK(running_mean->numel() == C && running_var->numel() == C && running_mean->is_contiguous() && running_var->is_contiguous() &&
                    running_mean->scalar_type() == running_var->scalar_type(),
                "batchnorm_forward: running statistics must be contiguous [C] tensors of one dtype");
    sdt = kdtype(*running_mean);
    rm = running_mean->data_ptr();
    rv = running_var->data_ptr();
  }
  const void* rp = nullptr;
  if (residual.has_value() && residual->defined()) {
    check_gpu(*residual, "residual");
    TORCH_CHECK(residual->sizes() == x.sizes() && residual->scalar_type() == x.scalar_type() && residual->is_contiguous() &&
                    residual->get_device() == x.get_device(),
                "batchnorm_forward: residual must match x");
    rp = residual->data_ptr();
  }
  auto fopt = x.options().dtype(at::kFloat);
  Tensor y = at::empty_like(x), mean = at::empty({C}, fopt), rstd = at::empty({C}, fopt), ss = at::empty({2 * C}, fopt);
  Tensor ws = at::empty({kernels::batchnorm_workspace(R, C, dt)}, fopt);
  DevGuard g(x);
  kernels::launch_batchnorm_fwd(x.data_ptr(), rp, y.data_ptr(), gp, bp, pdt, rm, rv, sdt, static_cast<float>(momentum), static_cast<float>(eps),
                                mean.data_ptr<float>(), rstd.data_ptr<float>(), ss.data_ptr<float>(), ws.data_ptr<float>(), R, C, relu ? 1 : 0, dt,
                                stream_of(x));
  return {y, mean, rstd, ss};
}

// y = relu(x * scale + shift (+ residual)) with fp32 per-channel coefficients (evaluation mode).
Tensor batchnorm_apply(Tensor x, Tensor scale, Tensor shift, c10::optional<Tensor> residual, bool relu) {
  check_gpu(x, "x");
  check_gpu(scale, "scale");
  check_gpu(shift, "shift");
  TORCH_CHECK(x.dim() == 2 && x.is_contiguous(), "batchnorm_apply: x must be a contiguous [R, C] tensor");
  const int64_t R = x.size(0), C = x.size(1);
  const int dt = kdtype(x);
  TORCH_CHECK(kernels::batchnorm_supported(R, C, dt), "batchnorm_apply: unsupported shape");
  TORCH_CHECK(scale.scalar_type() == at::kFloat && shift.scalar_type() == at::kFloat && scale.numel() == C && shift.numel() == C &&
                  scale.is_contiguous() && shift.is_contiguous(),
              "batchnorm_apply: scale and shift must be contiguous fp32 [C]");
  const void* rp = nullptr;
  if (residual.has_value() && residual->defined()) {
    check_gpu(*residual, "residual");
    TORCH_CHECK(residual->sizes() == x.sizes() && residual->scalar_type() == x.scalar_type() && residual->is_contiguous(),
                "batchnorm_apply: residual must match x");
    rp = residual->data_ptr();
  }
  Tensor y = at::empty_like(x);
  DevGuard g(x);
  kernels::launch_batchnorm_apply(x.data_ptr(), rp, y.data_ptr(), scale.data_ptr<float>(), shift.data_ptr<float>(), R, C, relu ? 1 : 0, dt,
                                  stream_of(x));
  return y;
}

// Returns (dx, dgamma, dbeta, dres); dgamma / dbeta in the dtype of gamma (fp32 without one), dres undefined unless wanted.
// The ReLU mask comes from y, or from x and the forward's coefficients `scale_shift` when those are given instead.
std::tuple<Tensor, Tensor, Tensor, Tensor> batchnorm_backward(Tensor dy, Tensor x, c10::optional<Tensor> y, c10::optional<Tensor> scale_shift,
                                                              c10::optional<Tensor> gamma, Tensor mean, Tensor rstd, bool relu, bool want_dres) {
  check_gpu(dy, "dy");
  check_gpu(x, "x");
  TORCH_CHECK(x.dim() == 2 && x.is_contiguous() && dy.is_contiguous() && dy.sizes() == x.sizes() && dy.scalar_type() == x.scalar_type() &&
                  dy.get_device() == x.get_device(),
              "batchnorm_backward: dy and x must be matching contiguous [R, C] tensors");
  const int64_t R = x.size(0), C = x.size(1);
  const int dt = kdtype(x);
  TORCH_CHECK(kernels::batchnorm_supported(R, C, dt), "batchnorm_backward: unsupported shape");
  TORCH_CHECK(mean.scalar_type() == at::kFloat && rstd.scalar_type() == at::kFloat && mean.numel() == C && rstd.numel() == C,
              "batchnorm_backward: mean and rstd must be fp32 [C]");
  const void* yp = nullptr;
  const float* ssp = nullptr;
  if (relu && scale_shift.has_value() && scale_shift->defined()) {
    check_gpu(*scale_shift, "scale_shift");
    TORCH_CHECK(scale_shift->scalar_type() == at::kFloat && scale_shift->numel() == 2 * C && scale_shift->is_contiguous(),
                "batchnorm_backward: scale_shift must be contiguous fp32 [2*C]");
    ssp = scale_shift->data_ptr<float>();
  } else if (relu) {
    TORCH_CHECK(y.has_value() && y->defined() && y->sizes() == x.sizes() && y->scalar_type() == x.scalar_type() && y->is_contiguous(),
                "batchnorm_backward: the fused ReLU needs the forward output or the forward's coefficients");
    yp = y->data_ptr();
  }
  int pdt = kernels::kF32;
  const void* gp = nullptr;
  auto popt = x.options().dtype(at::kFloat);
  if (gamma.has_value() && gamma->defined()) {
    check_gpu(*gamma, "weight");
    TORCH_CHECK(gamma->numel() == C && gamma->is_contiguous(), "batchnorm_backward: weight must be [C]");
    pdt = kdtype(*gamma);
    gp = gamma->data_ptr();
    popt = gamma->options();
  }
  Tensor dx = at::empty_like(x), dgamma = at::empty({C}, popt), dbeta = at::empty({C}, popt), dres;
  if (want_dres) dres = at::empty_like(x);
  Tensor ws = at::empty({kernels::batchnorm_workspace(R, C, dt)}, x.options().dtype(at::kFloat));
  DevGuard g(x);
  kernels::launch_batchnorm_bwd(dy.data_ptr(), x.data_ptr(), yp, ssp, gp, pdt, mean.data_ptr<float>(), rstd.data_ptr<float>(), dx.data_ptr(),
                                want_dres ? dres.data_ptr() : nullptr, dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr<float>(), R, C,
                                relu ? 1 : 0, dt, stream_of(x));
  return {dx, dgamma, dbeta, dres};
}

// ------------------------------------------------------------------ linear layers
bool gemm_path_ok(const kernels::GemmArgs& a, int dt) {
  return dt == kernels::kF32 ? kernels::gemm_f32_supported(a)
                             : (dt == kernels::kBF16 ? kernels::gemm_bf16_supported(a) : (dt == kernels::kF16 ? kernels::gemm_f16_supported(a) : false));
}

// `like` supplies the device for the split-K workspace (fp32 partials, caching allocator).
void run_gemm(const kernels::GemmArgs& a, int dt, hipStream_t s, const Tensor& like) {
  if (dt == kernels::kF32) {
    kernels::launch_gemm_f32(a, s);
    return;
  }
  if (dt == kernels::kF16) {
    kernels::launch_gemm_f16(a, s);
    return;
  }
  const int split = kernels::gemm_bf16_split_k(a);
  if (split > 1) {
    kernels::GemmArgs b = a;
    Tensor ws = at::empty({static_cast<int64_t>(split) * a.M * a.N}, like.options().dtype(at::kFloat));
    b.workspace = ws.data_ptr<float>();
    b.split_k = split;
    kernels::launch_gemm_bf16(b, s);
    return;
  }
  kernels::launch_gemm_bf16(a, s);
}

// Whether the MFMA GEMMs take C[M,N] = op(A) op(B)^T with contiguous operands in the given layouts.
bool gemm_shape_ok(int64_t M, int64_t N, int64_t K, bool a_kmajor, bool b_kmajor, int dt) {
  kernels::GemmArgs a{};
  a.M = M; a.N = N; a.K = K; a.lda = a_kmajor ? K : M; a.ldb = b_kmajor ? K : N; a.ldc = N;
  a.a_kmajor = a_kmajor; a.b_kmajor = b_kmajor;
  return gemm_path_ok(a, dt);
}

// Which kernel family a linear layer y = x W^T (+b) of this shape runs on.  "mfma" only when the forward AND both
// backward products (dW = g^T x with N and K as contiguous extents, dx = g W) fit the MFMA GEMMs' 16-byte
// vector layout, so a layer never passes its forward and then fails in its backward.
std::string linear_route(int64_t M, int64_t N, int64_t K, int dt) {
  if (K == 1) return "outer";
  if (N <= 16) return "skinny";
  const bool ok = gemm_shape_ok(M, N, K, true, true, dt) && gemm_shape_ok(N, K, M, false, false, dt) && gemm_shape_ok(M, K, N, true, false, dt);
  return ok ? "mfma" : "unsupported";
}

// y = act(x W^T + b).  x: [M, K], W: [N, K], b: [N] or None.  Returns (y, pre_activation or None).
std::tuple<Tensor, c10::optional<Tensor>> linear_forward(Tensor x, Tensor w, c10::optional<Tensor> bias, int64_t act, bool save_pre_act) {
  check_gpu(x, "x");
  check_gpu(w, "weight");
  TORCH_CHECK(x.dim() == 2 && w.dim() == 2 && x.size(1) == w.size(1), "linear_forward: x [M,K] and weight [N,K] expected");
  TORCH_CHECK(x.scalar_type() == w.scalar_type(), "linear_forward: x and weight dtypes differ");
  TORCH_CHECK(x.get_device() == w.get_device(), "linear_forward: device mismatch");
  const void* bptr = nullptr;
  if (bias.has_value() && bias->defined()) {
    check_gpu(*bias, "bias");
    TORCH_CHECK(bias->numel() == w.size(0) && bias->scalar_type() == w.scalar_type(), "linear_forward: bias must be [N] in the weight dtype");
    bptr = bias->data_ptr();
  }
  const int64_t M = x.size(0), K = x.size(1), N = w.size(0);
  const int dt = kdtype(x);
  Tensor y = at::empty({M, N}, x.options());
  c10::optional<Tensor> pre;
  DevGuard g(x);
  hipStream_t s = stream_of(x);
  if (M == 0 || N == 0) return {y, pre};
  if (K == 1) {
    TORCH_CHECK(!save_pre_act, "linear_forward: pre-activation output is not provided for K == 1");
    kernels::launch_outer_bias_act(x.data_ptr(), w.data_ptr(), bptr, y.data_ptr(), M, N, dt, static_cast<int>(act), s);
  } else if (N <= 16) {
    TORCH_CHECK(act == kernels::kActNone, "linear_forward: no fused activation on the N <= 16 path");
    kernels::launch_skinny_n(x.data_ptr(), w.data_ptr(), bptr, y.data_ptr(), M, N, K, dt, s);
  } else {
    kernels::GemmArgs a{};
    a.A = x.data_ptr(); a.B = w.data_ptr(); a.C = y.data_ptr(); a.bias = bptr;
    a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = K; a.ldc = N;
    a.a_kmajor = 1; a.b_kmajor = 1; a.act = static_cast<int>(act); a.alpha = 1.f; a.beta = 0.f;
    if (save_pre_act && act != kernels::kActNone) {
      pre = at::empty({M, N}, x.options());
      a.pre_act = pre->data_ptr();
    }
    TORCH_CHECK(gemm_path_ok(a, dt), "linear_forward: shape M=", M, " N=", N, " K=", K, " is not supported by the MFMA GEMM for dtype ", x.scalar_type());
    run_gemm(a, dt, s, x);
  }
  return {y, pre};
}

// ------------------------------------------------------------------ Linear(1,H) -> act -> Linear(H,N<=16), fused
namespace {
const void* opt_ptr(const c10::optional<Tensor>& t, const Tensor& like, int64_t numel, const char* what) {
  if (!t.has_value() || !t->defined()) return nullptr;
  check_gpu(*t, what);
  TORCH_CHECK(t->numel() == numel && t->scalar_type() == like.scalar_type() && t->get_device() == like.get_device(),
              "rank1_mlp: ", what, " must hold ", numel, " elements of the weight dtype on the same device");
  return t->data_ptr();
}
void check_rank1(const Tensor& x, const Tensor& w1, const Tensor& W2, int64_t act) {
  check_gpu(x, "x");
  check_gpu(w1, "w1");
  check_gpu(W2, "W2");
  TORCH_CHECK(W2.dim() == 2 && w1.numel() == W2.size(1), "rank1_mlp: w1 [H] (or [H,1]) and W2 [N,H] expected");
  TORCH_CHECK(x.dim() <= 2 && (x.dim() < 2 || x.size(1) == 1), "rank1_mlp: x must be [M] or [M,1]");
  TORCH_CHECK(W2.size(0) >= 1 && W2.size(0) <= 16, "rank1_mlp: 1 <= N <= 16 outputs");
  TORCH_CHECK(W2.size(1) < (1ll << 30), "rank1_mlp: hidden size too large");
  TORCH_CHECK(x.scalar_type() == w1.scalar_type() && x.scalar_type() == W2.scalar_type(), "rank1_mlp: dtypes differ");
  TORCH_CHECK(x.get_device() == w1.get_device() && x.get_device() == W2.get_device(), "rank1_mlp: device mismatch");
  TORCH_CHECK(kdtype(x) >= 0, "rank1_mlp: unsupported dtype");
  TORCH_CHECK(act == kernels::kActNone || act == kernels::kActRelu, "rank1_mlp: activation must be none or relu");
}
}  // namespace

Tensor rank1_mlp_forward(Tensor x, Tensor w1, c10::optional<Tensor> b1, Tensor W2, c10::optional<Tensor> b2, int64_t act) {
  check_rank1(x, w1, W2, act);
  const int64_t M = x.numel(), H = W2.size(1), N = W2.size(0);
  const void* b1p = opt_ptr(b1, w1, H, "b1");
  const void* b2p = opt_ptr(b2, w1, N, "b2");
  Tensor y = at::empty({M, N}, x.options());
  DevGuard g(x);
  Tensor pk = at::empty({kernels::rank1_mlp_fwd_workspace(H, N)}, x.options().dtype(at::kFloat));
  kernels::launch_rank1_mlp_fwd(x.data_ptr(), w1.data_ptr(), b1p, W2.data_ptr(), b2p, y.data_ptr(), pk.data_ptr<float>(), M, H, N,
                                kdtype(x), static_cast<int>(act), stream_of(x));
  return y;
}

// Returns (dw1 [H,1], db1 [H], dW2 [N,H], db2 [N]) as views of one buffer.
std::tuple<Tensor, Tensor, Tensor, Tensor> rank1_mlp_backward(Tensor dy, Tensor x, Tensor w1, c10::optional<Tensor> b1, Tensor W2, int64_t act) {
  check_rank1(x, w1, W2, act);
  check_gpu(dy, "dy");
  const int64_t M = x.numel(), H = W2.size(1), N = W2.size(0);
  TORCH_CHECK(dy.dim() == 2 && dy.size(0) == M && dy.size(1) == N && dy.scalar_type() == x.scalar_type() && dy.get_device() == x.get_device(),
              "rank1_mlp_backward: dy must be [M,N] in the dtype and on the device of x");
  const void* b1p = opt_ptr(b1, w1, H, "b1");
  const int64_t len = kernels::rank1_mlp_bwd_len(H, N);
  const int64_t NT = (len - 2 * H) / (H + 1);
  Tensor grads = at::empty({len}, x.options());
  DevGuard g(x);
  if (M == 0) {
    grads.zero_();
  } else {
    Tensor partial = at::empty({static_cast<int64_t>(kernels::rank1_mlp_bwd_chunks(M, H)) * len}, x.options().dtype(at::kFloat));
    kernels::launch_rank1_mlp_bwd(dy.data_ptr(), x.data_ptr(), w1.data_ptr(), b1p, W2.data_ptr(), partial.data_ptr<float>(),
                                  grads.data_ptr(), M, H, N, kdtype(x), static_cast<int>(act), stream_of(x));
  }
  return {grads.narrow(0, NT * H, H).view({H, 1}), grads.narrow(0, (NT + 1) * H, H), grads.narrow(0, 0, N * H).view({N, H}),
          grads.narrow(0, (NT + 2) * H, N)};
}

// Gradients of y = x W^T + b given g = dL/d(xW^T+b).  Returns (dx or None, dW, db or None).
// dw_out: write the weight gradient into this tensor (a slice of a gradient bucket) instead of a fresh one.
std::tuple<c10::optional<Tensor>, Tensor, c10::optional<Tensor>> linear_backward(Tensor g, Tensor x, Tensor w, bool need_dx, bool need_db,
                                                                                   c10::optional<Tensor> dw_out) {
  check_gpu(g, "grad");
  check_gpu(x, "x");
  check_gpu(w, "weight");
  TORCH_CHECK(g.dim() == 2 && x.dim() == 2 && w.dim() == 2, "linear_backward: 2-D tensors expected");
  const int64_t M = x.size(0), K = x.size(1), N = w.size(0);
  TORCH_CHECK(g.size(0) == M && g.size(1) == N && w.size(1) == K, "linear_backward: shape mismatch");
  TORCH_CHECK(g.scalar_type() == x.scalar_type() && x.scalar_type() == w.scalar_type(), "linear_backward: dtype mismatch");
  TORCH_CHECK(g.get_device() == x.get_device() && x.get_device() == w.get_device(), "linear_backward: device mismatch");
  const int dt = kdtype(x);
  DevGuard guard(x);
  hipStream_t s = stream_of(x);
  Tensor dW;
  if (dw_out.has_value() && dw_out->defined()) {
    check_gpu(*dw_out, "dw_out");
    TORCH_CHECK(dw_out->sizes() == w.sizes() && dw_out->scalar_type() == w.scalar_type() && dw_out->is_contiguous() &&
                    dw_out->get_device() == w.get_device(), "linear_backward: dw_out must be a contiguous tensor like the weight");
    dW = *dw_out;
  } else {
    dW = at::empty_like(w);
  }
  c10::optional<Tensor> dx, db;
  if (need_db) db = at::empty({N}, w.options());
  if (need_dx) dx = at::empty_like(x);
  auto fopt = x.options().dtype(at::kFloat);
  if (K == 1) {
    Tensor ws = at::empty({kernels::colreduce_workspace(M, N)}, fopt);
    Tensor db_tmp = need_db ? *db : at::empty({N}, w.options());
    kernels::launch_outer_wgrad(g.data_ptr(), x.data_ptr(), dW.data_ptr(), db_tmp.data_ptr(), ws.data_ptr<float>(), M, N, dt, s);
    if (need_dx) {
      // dx[m] = sum_n g[m,n] w[n]: a skinny product with a single output column.
      kernels::launch_skinny_n(g.data_ptr(), w.data_ptr(), nullptr, dx->data_ptr(), M, 1, N, dt, s);
    }
  } else if (N <= 16) {
    Tensor ws = at::empty({kernels::skinny_wgrad_workspace(M, N, K)}, fopt);
    kernels::launch_skinny_n_wgrad(g.data_ptr(), x.data_ptr(), dW.data_ptr(), need_db ? db->data_ptr() : nullptr, ws.data_ptr<float>(), M, N, K, dt, s);
    if (need_dx) kernels::launch_skinny_n_dgrad(g.data_ptr(), w.data_ptr(), dx->data_ptr(), M, N, K, dt, s);
  } else {
    // dW[N,K] = g^T x : contraction over M; both operands are stored with the contraction dim slowest.
    kernels::GemmArgs a{};
    a.A = g.data_ptr(); a.B = x.data_ptr(); a.C = dW.data_ptr();
    a.M = N; a.N = K; a.K = M; a.lda = N; a.ldb = K; a.ldc = K; a.a_kmajor = 0; a.b_kmajor = 0; a.alpha = 1.f;
    TORCH_CHECK(gemm_path_ok(a, dt), "linear_backward: weight-gradient shape not supported by the MFMA GEMM");
    run_gemm(a, dt, s, g);
    if (need_dx) {
      // dx[M,K] = g W : contraction over N; A = g is contraction-contiguous, B = W is stored [N(contraction), K].
      kernels::GemmArgs b{};
      b.A = g.data_ptr(); b.B = w.data_ptr(); b.C = dx->data_ptr();
      b.M = M; b.N = K; b.K = N; b.lda = N; b.ldb = K; b.ldc = K; b.a_kmajor = 1; b.b_kmajor = 0; b.alpha = 1.f;
      TORCH_CHECK(gemm_path_ok(b, dt), "linear_backward: input-gradient shape not supported by the MFMA GEMM");
      run_gemm(b, dt, s, g);
    }
    if (need_db) {
      Tensor ws = at::empty({kernels::colreduce_workspace(M, N)}, fopt);
      kernels::launch_colsum(g.data_ptr(), db->data_ptr(), ws.data_ptr<float>(), M, N, dt, s);
    }
  }
  return {dx, dW, db};
}

// g = dy * act'(saved) and optionally db = colsum(g).  saved: output y for ReLU, pre-activation for GELU.
std::tuple<Tensor, c10::optional<Tensor>> act_backward(Tensor dy, Tensor saved, int64_t act, bool need_db) {
  check_gpu(dy, "dy");
  check_gpu(saved, "saved");
  TORCH_CHECK(dy.dim() == 2 && saved.sizes() == dy.sizes() && saved.scalar_type() == dy.scalar_type(), "act_backward: shape/dtype mismatch");
  TORCH_CHECK(dy.get_device() == saved.get_device(), "act_backward: device mismatch");
  const int64_t M = dy.size(0), N = dy.size(1);
  Tensor gout = at::empty_like(dy);
  c10::optional<Tensor> db;
  if (need_db) db = at::empty({N}, dy.options());
  Tensor ws = at::empty({kernels::colreduce_workspace(M, N)}, dy.options().dtype(at::kFloat));
  DevGuard guard(dy);
  kernels::launch_act_bwd(dy.data_ptr(), saved.data_ptr(), gout.data_ptr(), need_db ? db->data_ptr() : nullptr, ws.data_ptr<float>(), M, N,
                          kdtype(dy), static_cast<int>(act), stream_of(dy));
  return {gout, db};
}

// db[N] = column sums of dy[M,N] (bias gradient), fp32 accumulation, deterministic.
Tensor colsum(Tensor dy) {
  check_gpu(dy, "dy");
  TORCH_CHECK(dy.dim() == 2, "colsum: [M, N] expected");
  const int64_t M = dy.size(0), N = dy.size(1);
  Tensor db = at::empty({N}, dy.options());
  Tensor ws = at::empty({kernels::colreduce_workspace(M, N)}, dy.options().dtype(at::kFloat));
  DevGuard guard(dy);
  kernels::launch_colsum(dy.data_ptr(), db.data_ptr(), ws.data_ptr<float>(), M, N, kdtype(dy), stream_of(dy));
  return db;
}

// General matrix product on the MFMA kernels (used by tests and benchmarks):
// C[M,N] = alpha * op(A) op(B)^T with the layout flags of kernels::GemmArgs.
// out: write C into this [M, N] tensor (e.g. a gradient-bucket slice) instead of allocating it.
Tensor gemm(Tensor A, Tensor B, bool a_kmajor, bool b_kmajor, c10::optional<Tensor> bias, int64_t act, c10::optional<Tensor> out) {
  check_gpu(A, "A");
  check_gpu(B, "B");
  TORCH_CHECK(A.dim() == 2 && B.dim() == 2 && A.scalar_type() == B.scalar_type(), "gemm: 2-D tensors of one dtype expected");
  TORCH_CHECK(A.get_device() == B.get_device(), "gemm: device mismatch");
  const int64_t M = a_kmajor ? A.size(0) : A.size(1);
  const int64_t K = a_kmajor ? A.size(1) : A.size(0);
  const int64_t N = b_kmajor ? B.size(0) : B.size(1);
  const int64_t Kb = b_kmajor ? B.size(1) : B.size(0);
  TORCH_CHECK(K == Kb, "gemm: contraction sizes differ (", K, " vs ", Kb, ")");
  Tensor C;
  if (out.has_value() && out->defined()) {
    check_gpu(*out, "out");
    TORCH_CHECK(out->dim() == 2 && out->size(0) == M && out->size(1) == N && out->scalar_type() == A.scalar_type() && out->is_contiguous() &&
                    out->get_device() == A.get_device(), "gemm: out must be a contiguous [M, N] tensor in the operand dtype");
    C = *out;
  } else {
    C = at::empty({M, N}, A.options());
  }
  kernels::GemmArgs a{};
  a.A = A.data_ptr(); a.B = B.data_ptr(); a.C = C.data_ptr();
  if (bias.has_value() && bias->defined()) {
    check_gpu(*bias, "bias");
    TORCH_CHECK(bias->numel() == N && bias->scalar_type() == A.scalar_type(), "gemm: bias must be [N] in the operand dtype");
    a.bias = bias->data_ptr();
  }
  a.M = M; a.N = N; a.K = K; a.lda = A.size(1); a.ldb = B.size(1); a.ldc = N;
  a.a_kmajor = a_kmajor; a.b_kmajor = b_kmajor; a.act = static_cast<int>(act); a.alpha = 1.f; a.beta = 0.f;
  const int dt = kdtype(A);
  TORCH_CHECK(gemm_path_ok(a, dt), "gemm: shape M=", M, " N=", N, " K=", K, " not supported for dtype ", A.scalar_type());
  DevGuard guard(A);
  run_gemm(a, dt, stream_of(A), A);
  return C;
}

// ------------------------------------------------------------------ 3x3 convolution (stride 1, padding 1) as an implicit GEMM
namespace {
struct HaloGeom {
  int64_t B, H, W, C, guard, Q, Kpad, rows_total;
};
HaloGeom halo_geom(int64_t B, int64_t H, int64_t W, int64_t C) {
  HaloGeom g{B, H, W, C, 0, 0, 0, 0};
  g.guard = W + 3;
  g.Q = B * (H + 2) * (W + 2);
  g.Kpad = (g.Q + 63) / 64 * 64;
  g.rows_total = g.guard + g.Kpad + g.guard;
  return g;
}
}  // namespace

// x: [B, H, W, C] contiguous bf16 (the channels-last image seen as rows).  Returns the zero-haloed copy
// [guard + Kpad + guard, C]; haloed pixel q is row guard + q (see kernels::launch_pad_halo).
Tensor conv3x3_pad(Tensor x) {
  check_gpu(x, "x");
  TORCH_CHECK(x.dim() == 4 && x.is_contiguous() && x.scalar_type() == at::kBFloat16, "conv3x3_pad: contiguous bf16 [B, H, W, C] expected");
  const HaloGeom g = halo_geom(x.size(0), x.size(1), x.size(2), x.size(3));
  TORCH_CHECK(g.C % 64 == 0, "conv3x3_pad: channels must be a multiple of 64");
  Tensor buf = at::empty({g.rows_total, g.C}, x.options());
  DevGuard guard(x);
  kernels::launch_pad_halo(x.data_ptr(), buf.data_ptr(), g.rows_total, g.guard, static_cast<int>(g.B), static_cast<int>(g.H),
                           static_cast<int>(g.W), static_cast<int>(g.C), stream_of(x));
  return buf;
}

// y[B*H*W, N] = sum over taps of X[pixel + tap] . w[N][tap][C]   (xpad from conv3x3_pad, w: [N, 9*C] contiguous).
Tensor conv3x3_forward(Tensor xpad, Tensor w, int64_t B, int64_t H, int64_t W) {
  check_gpu(xpad, "xpad");
  check_gpu(w, "weight");
  TORCH_CHECK(xpad.dim() == 2 && w.dim() == 2 && xpad.is_contiguous() && w.is_contiguous(), "conv3x3_forward: 2-D contiguous tensors expected");
  TORCH_CHECK(xpad.scalar_type() == at::kBFloat16 && w.scalar_type() == at::kBFloat16, "conv3x3_forward: bf16 expected");
  const HaloGeom g = halo_geom(B, H, W, xpad.size(1));
  TORCH_CHECK(xpad.size(0) == g.rows_total && w.size(1) == 9 * g.C, "conv3x3_forward: xpad / weight do not match the image geometry");
  const int64_t N = w.size(0);
  Tensor y = at::empty({B * H * W, N}, xpad.options());
  kernels::GemmArgs a{};
  a.A = static_cast<const char*>(xpad.data_ptr()) + g.guard * g.C * 2;
  a.B = w.data_ptr();
  a.C = y.data_ptr();
  a.M = B * H * W; a.N = N; a.K = 9 * g.C; a.lda = a.K; a.ldb = a.K; a.ldc = N;
  a.a_kmajor = 1; a.b_kmajor = 1; a.alpha = 1.f;
  a.conv_mode = 1; a.conv_B = static_cast<int>(B); a.conv_H = static_cast<int>(H); a.conv_W = static_cast<int>(W); a.conv_C = static_cast<int>(g.C);
  TORCH_CHECK(kernels::gemm_bf16_conv_supported(a), "conv3x3_forward: unsupported geometry B=", B, " H=", H, " W=", W, " C=", g.C, " N=", N);
  DevGuard guard(xpad);
  kernels::launch_gemm_bf16(a, stream_of(xpad));
  return y;
}

// dW[Cout, 9*Cin] = sum over all haloed pixels q of dY[q][Cout]^T X[q + tap][Cin]   (both from conv3x3_pad of the same geometry).
Tensor conv3x3_wgrad(Tensor dypad, Tensor xpad, int64_t B, int64_t H, int64_t W) {
  check_gpu(dypad, "dypad");
  check_gpu(xpad, "xpad");
  TORCH_CHECK(dypad.dim() == 2 && xpad.dim() == 2 && dypad.is_contiguous() && xpad.is_contiguous(), "conv3x3_wgrad: 2-D contiguous tensors expected");
  TORCH_CHECK(dypad.scalar_type() == at::kBFloat16 && xpad.scalar_type() == at::kBFloat16, "conv3x3_wgrad: bf16 expected");
  const HaloGeom g = halo_geom(B, H, W, xpad.size(1));
  const int64_t Cout = dypad.size(1);
  TORCH_CHECK(xpad.size(0) == g.rows_total && dypad.size(0) == g.rows_total, "conv3x3_wgrad: operands do not match the image geometry");
  Tensor dw = at::empty({Cout, 9 * g.C}, xpad.options());
  kernels::GemmArgs a{};
  a.A = static_cast<const char*>(dypad.data_ptr()) + g.guard * Cout * 2;
  a.B = static_cast<const char*>(xpad.data_ptr()) + g.guard * g.C * 2;
  a.C = dw.data_ptr();
  a.M = Cout; a.N = 9 * g.C; a.K = g.Kpad; a.lda = Cout; a.ldb = g.C; a.ldc = a.N;
  a.a_kmajor = 0; a.b_kmajor = 0; a.alpha = 1.f;
  a.conv_mode = 2; a.conv_B = static_cast<int>(B); a.conv_H = static_cast<int>(H); a.conv_W = static_cast<int>(W); a.conv_C = static_cast<int>(g.C);
  // one wave of workgroups over the chip: tiles * ranges ~ 256, at least 16 k-steps per range
  const int64_t tiles = ((a.M + 255) / 256) * ((a.N + 255) / 256), nk = a.K / 64;
  int64_t split = std::max<int64_t>(1, 256 / tiles);
  split = std::min<int64_t>(split, std::max<int64_t>(1, nk / 16));
  const int64_t iters = (nk + split - 1) / split;
  split = (nk + iters - 1) / iters;
  Tensor ws = at::empty({split * a.M * a.N}, xpad.options().dtype(at::kFloat));
  a.workspace = ws.data_ptr<float>();
  a.split_k = static_cast<int>(split);
  TORCH_CHECK(kernels::gemm_bf16_conv_supported(a), "conv3x3_wgrad: unsupported geometry B=", B, " H=", H, " W=", W, " Cin=", g.C, " Cout=", Cout);
  DevGuard guard(xpad);
  kernels::launch_gemm_bf16(a, stream_of(xpad));
  return dw;
}

// g = (dout @ w) * act'(aux) in one kernel, plus db = colsum(g) when wanted: the input-gradient
// GEMM of a layer fused with the activation backward of the layer in front of it.
//   dout [M, K] (gradient of the layer's output), w [K, N] (its weight, as stored: [out, in]),
//   aux [M, N]: pre-activation (GELU) or activation output (ReLU) of the layer in front.
std::tuple<Tensor, Tensor> dgrad_act_backward(Tensor dout, Tensor w, Tensor aux, int64_t act, bool want_colsum) {
  check_gpu(dout, "dout");
  check_gpu(w, "weight");
  check_gpu(aux, "aux");
  TORCH_CHECK(dout.dim() == 2 && w.dim() == 2 && aux.dim() == 2 && dout.size(1) == w.size(0) && aux.size(0) == dout.size(0) &&
                  aux.size(1) == w.size(1), "dgrad_act_backward: shapes must be dout [M,K], weight [K,N], aux [M,N]");
  TORCH_CHECK(dout.scalar_type() == at::kBFloat16 && w.scalar_type() == at::kBFloat16 && aux.scalar_type() == at::kBFloat16,
              "dgrad_act_backward: bf16 tensors expected");
  TORCH_CHECK(dout.get_device() == w.get_device() && dout.get_device() == aux.get_device(), "dgrad_act_backward: device mismatch");
  const int64_t M = dout.size(0), K = dout.size(1), N = w.size(1);
  Tensor g = at::empty({M, N}, dout.options());
  kernels::GemmArgs a{};
  a.A = dout.data_ptr(); a.B = w.data_ptr(); a.C = g.data_ptr();
  a.M = M; a.N = N; a.K = K;
  a.lda = K; a.ldb = N; a.ldc = N;
  a.a_kmajor = 1; a.b_kmajor = 0;
  a.act = kernels::kActNone; a.alpha = 1.f;
  a.aux = aux.data_ptr(); a.aux_mode = static_cast<int>(act);
  Tensor partial, db;
  if (want_colsum) {
    partial = at::empty({2 * (M / 256), N}, dout.options().dtype(at::kFloat));
    a.colsum_partial = partial.data_ptr<float>();
    db = at::empty({N}, dout.options());
  }
  TORCH_CHECK(kernels::gemm_bf16_act_backward_supported(a),
              "dgrad_act_backward: needs M and N multiples of 256, K a multiple of 64, 16-byte aligned tensors, act relu | gelu");
  DevGuard guard(dout);
  hipStream_t s = stream_of(dout);
  kernels::launch_gemm_bf16(a, s);
  if (want_colsum) kernels::launch_reduce_partials(partial.data_ptr<float>(), db.data_ptr(), N, static_cast<int>(2 * (M / 256)), kernels::kBF16, s);
  return {g, db};
}

bool dgrad_act_backward_supported(int64_t M, int64_t N, int64_t K) { return M % 256 == 0 && N % 256 == 0 && K % 64 == 0 && M > 0 && N > 0 && K > 0; }

// ------------------------------------------------------------------ layer norm
// y = LN(x) or, with `res`, s = x + res and y = LN(s).  Returns (y, mean, rstd, s); s is x itself without `res`.
std::tuple<Tensor, Tensor, Tensor, Tensor> layernorm_forward(Tensor x, Tensor gamma, Tensor beta, double eps, c10::optional<Tensor> res,
                                                             c10::optional<Tensor> res_bias) {
  check_gpu(x, "x");
  check_gpu(gamma, "weight");
  check_gpu(beta, "bias");
  TORCH_CHECK(x.dim() == 2, "layernorm_forward: x must be [rows, cols]");
  const int64_t rows = x.size(0), cols = x.size(1);
  TORCH_CHECK(gamma.numel() == cols && beta.numel() == cols, "layernorm_forward: affine parameters must be [cols]");
  TORCH_CHECK(gamma.scalar_type() == x.scalar_type() && beta.scalar_type() == x.scalar_type(), "layernorm_forward: dtype mismatch");
  TORCH_CHECK(kernels::layernorm_supported(cols), "layernorm_forward: at most 8192 columns");
  Tensor y = at::empty_like(x);
  Tensor sum = x;
  const void* res_ptr = nullptr;
  if (res.has_value()) {
    check_gpu(*res, "res");
    TORCH_CHECK(res->sizes() == x.sizes() && res->scalar_type() == x.scalar_type() && res->get_device() == x.get_device(),
                "layernorm_forward: res must match x");
    sum = at::empty_like(x);
    res_ptr = res->data_ptr();
  }
  const void* res_bias_ptr = nullptr;
  if (res_bias.has_value()) {
    TORCH_CHECK(res.has_value(), "layernorm_forward: res_bias needs res");
    check_gpu(*res_bias, "res_bias");
    TORCH_CHECK(res_bias->numel() == cols && res_bias->scalar_type() == x.scalar_type() && res_bias->get_device() == x.get_device(),
                "layernorm_forward: res_bias must be [cols] of x's dtype");
    res_bias_ptr = res_bias->data_ptr();
  }
  auto fopt = x.options().dtype(at::kFloat);
  Tensor mean = at::empty({rows}, fopt), rstd = at::empty({rows}, fopt);
  DevGuard guard(x);
  kernels::launch_layernorm_fwd(x.data_ptr(), res_ptr, res_bias_ptr, res_ptr ? sum.data_ptr() : nullptr, gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                mean.data_ptr<float>(), rstd.data_ptr<float>(), rows, cols, static_cast<float>(eps), kdtype(x), stream_of(x));
  return {y, mean, rstd, sum};
}

// dx = LN'(dy) + dres (dres optional).  want_dsum: the fourth result is the column sum of dx (else undefined).
std::tuple<Tensor, Tensor, Tensor, Tensor> layernorm_backward(Tensor dy, Tensor x, Tensor gamma, Tensor mean, Tensor rstd, c10::optional<Tensor> dres,
                                                              bool want_dsum) {
  check_gpu(dy, "dy");
  check_gpu(x, "x");
  check_gpu(gamma, "weight");
  check_gpu(mean, "mean");
  check_gpu(rstd, "rstd");
  TORCH_CHECK(x.dim() == 2 && dy.sizes() == x.sizes() && dy.scalar_type() == x.scalar_type(), "layernorm_backward: shape/dtype mismatch");
  const int64_t rows = x.size(0), cols = x.size(1);
  TORCH_CHECK(gamma.numel() == cols && mean.numel() == rows && rstd.numel() == rows, "layernorm_backward: size mismatch");
  TORCH_CHECK(mean.scalar_type() == at::kFloat && rstd.scalar_type() == at::kFloat, "layernorm_backward: statistics must be float32");
  TORCH_CHECK(kernels::layernorm_supported(cols), "layernorm_backward: at most 8192 columns");
  const void* dres_ptr = nullptr;
  if (dres.has_value()) {
    check_gpu(*dres, "dres");
    TORCH_CHECK(dres->sizes() == x.sizes() && dres->scalar_type() == x.scalar_type() && dres->get_device() == x.get_device(),
                "layernorm_backward: dres must match x");
    dres_ptr = dres->data_ptr();
  }
  Tensor dx = at::empty_like(x), dgamma = at::empty_like(gamma), dbeta = at::empty_like(gamma);
  Tensor dsum = want_dsum ? at::empty_like(gamma) : Tensor();
  Tensor ws = at::empty({kernels::layernorm_bwd_workspace(rows, cols)}, x.options().dtype(at::kFloat));
  DevGuard guard(x);
  kernels::launch_layernorm_bwd(dy.data_ptr(), x.data_ptr(), dres_ptr, gamma.data_ptr(), mean.data_ptr<float>(), rstd.data_ptr<float>(),
                                dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), want_dsum ? dsum.data_ptr() : nullptr, ws.data_ptr<float>(), rows,
                                cols, kdtype(x), stream_of(x));
  return {dx, dgamma, dbeta, dsum};
}

// ------------------------------------------------------------------ attention
// qkv: [B, T, 3*H*64] bf16 (q | k | v along the last dimension).  Returns (o [B,T,H*64], lse [B,H,T]).
// kernel: "auto", "classic" (32 queries per wave, two waves per SIMD) or "wide" (64 queries per wave; B*H must be even).
std::tuple<Tensor, Tensor> attention_forward(Tensor qkv, int64_t n_head, double softmax_scale, const std::string& kernel) {
  check_gpu(qkv, "qkv");
  TORCH_CHECK(qkv.dim() == 3 && qkv.scalar_type() == at::kBFloat16, "attention_forward: qkv must be bf16 [B, T, 3*C]");
  const int64_t B = qkv.size(0), T = qkv.size(1), C3 = qkv.size(2);
  TORCH_CHECK(C3 % 3 == 0 && (C3 / 3) % n_head == 0, "attention_forward: last dimension must be 3 * n_head * head_dim");
  const int64_t Cd = C3 / 3, hd = Cd / n_head;
  TORCH_CHECK(kernels::attention_supported(T, hd, C3), "attention_forward: needs head_dim 64 and T a multiple of 128 (got head_dim ", hd, ", T ", T, ")");
  Tensor o = at::empty({B, T, Cd}, qkv.options());
  Tensor lse = at::empty({B, n_head, T}, qkv.options().dtype(at::kFloat));
  DevGuard guard(qkv);
  const char* base = static_cast<const char*>(qkv.data_ptr());
  TORCH_CHECK(kernel == "auto" || kernel == "classic" || kernel == "wide" || kernel == "tile32", "attention_forward: kernel must be auto, classic, wide or tile32");
  TORCH_CHECK(kernel != "wide" || (B * n_head) % 2 == 0, "attention_forward: the wide kernel takes heads two at a time, B * n_head must be even");
  kernels::launch_attention_fwd(base, base + Cd * 2, base + 2 * Cd * 2, o.data_ptr(), lse.data_ptr<float>(), C3, static_cast<int>(B),
                                static_cast<int>(n_head), static_cast<int>(T), static_cast<float>(softmax_scale), stream_of(qkv),
                                kernel == "auto" ? -1 : kernel == "wide" ? 1 : kernel == "tile32" ? 2 : 0);
  return {o, lse};
}

// Returns (dqkv [B, T, 3*C], dbias [3*C] or undefined) given the forward's inputs / outputs and do [B, T, C].
// want_bias_grad: the kernels also emit the column sums of dqkv (gradient of the projection's bias).
// kernel: "auto", "classic" (16x16x32 matrix instructions) or "wide" (32x32x16).
std::tuple<Tensor, Tensor> attention_backward(Tensor d_o, Tensor qkv, Tensor o, Tensor lse, int64_t n_head, double softmax_scale, bool want_bias_grad,
                                              const std::string& kernel) {
  TORCH_CHECK(kernel == "auto" || kernel == "classic" || kernel == "wide", "attention_backward: kernel must be auto, classic or wide");
  check_gpu(d_o, "do");
  check_gpu(qkv, "qkv");
  check_gpu(o, "o");
  check_gpu(lse, "lse");
  TORCH_CHECK(qkv.dim() == 3 && qkv.scalar_type() == at::kBFloat16 && d_o.scalar_type() == at::kBFloat16 && o.scalar_type() == at::kBFloat16,
              "attention_backward: bf16 tensors expected");
  const int64_t B = qkv.size(0), T = qkv.size(1), C3 = qkv.size(2), Cd = C3 / 3, hd = Cd / n_head;
  TORCH_CHECK(d_o.sizes() == o.sizes() && o.size(0) == B && o.size(1) == T && o.size(2) == Cd, "attention_backward: shape mismatch");
  TORCH_CHECK(lse.scalar_type() == at::kFloat && lse.numel() == B * n_head * T, "attention_backward: lse must be float32 [B, H, T]");
  TORCH_CHECK(kernels::attention_supported(T, hd, C3), "attention_backward: unsupported shape");
  TORCH_CHECK(d_o.is_contiguous() && qkv.is_contiguous() && o.is_contiguous() && lse.is_contiguous(), "attention_backward: contiguous tensors expected");
  Tensor dqkv = at::empty_like(qkv);
  Tensor delta = at::empty({2, B, n_head, T}, lse.options());  // delta | lse*log2(e), filled by the dQ kernel
  Tensor partial, dbias;
  if (want_bias_grad) {
    partial = at::empty({B * T / 128, C3}, lse.options());
    dbias = at::empty({C3}, qkv.options());
  }
  DevGuard guard(qkv);
  hipStream_t s = stream_of(qkv);
  const char* base = static_cast<const char*>(qkv.data_ptr());
  char* gbase = static_cast<char*>(dqkv.data_ptr());
  kernels::launch_attention_bwd(base, base + Cd * 2, base + 2 * Cd * 2, o.data_ptr(), d_o.data_ptr(), lse.data_ptr<float>(), delta.data_ptr<float>(),
                                gbase, gbase + Cd * 2, gbase + 2 * Cd * 2, C3, C3, static_cast<int>(B), static_cast<int>(n_head),
                                static_cast<int>(T), static_cast<float>(softmax_scale), want_bias_grad ? partial.data_ptr<float>() : nullptr, s,
                                kernel == "auto" ? -1 : kernel == "wide" ? 1 : 0);
  if (want_bias_grad) kernels::launch_reduce_partials(partial.data_ptr<float>(), dbias.data_ptr(), C3, static_cast<int>(B * T / 128), kernels::kBF16, s);
  return {dqkv, dbias};
}

bool aligned16(const Tensor& t) { return reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 == 0; }

bool embedding_ok(const Tensor& idx, const Tensor& wte, const Tensor& wpe, int64_t T) {
  return idx.is_cuda() && wte.is_cuda() && wpe.is_cuda() && idx.scalar_type() == at::kLong && wte.scalar_type() == at::kBFloat16 &&
         wpe.scalar_type() == at::kBFloat16 && idx.is_contiguous() && wte.is_contiguous() && wpe.is_contiguous() && wte.dim() == 2 &&
         wpe.dim() == 2 && wte.size(1) == wpe.size(1) && aligned16(wte) && aligned16(wpe) &&
         kernels::embedding_supported(idx.numel(), T, wte.size(1), wte.size(0), wpe.size(0));
}

// idx: int64 [B, T] (or any shape whose last dimension is T) -> bf16 [idx.numel(), C]
Tensor embedding_forward(Tensor idx, Tensor wte, Tensor wpe, int64_t T) {
  TORCH_CHECK(embedding_ok(idx, wte, wpe, T), "embedding_forward: needs int64 ids, bf16 [V, C] / [P, C] tables with C % 8 == 0, T <= P");
  const int64_t rows = idx.numel(), C = wte.size(1);
  Tensor out = at::empty({rows, C}, wte.options());
  DevGuard guard(wte);
  kernels::launch_embedding_fwd(reinterpret_cast<const long long*>(idx.data_ptr<int64_t>()), wte.data_ptr(), wpe.data_ptr(), out.data_ptr(),
                                rows, static_cast<int>(T), static_cast<int>(C), wte.size(0), stream_of(wte));
  return out;
}

// grad: bf16 [rows, C]; tok_sorted / perm: the stable-sorted ids and the rows they came from.  Returns (dwte [V, C], dwpe [P, C]).
std::tuple<Tensor, Tensor> embedding_backward(Tensor grad, Tensor tok_sorted, Tensor perm, int64_t T, int64_t V, int64_t P) {
  check_gpu(grad, "grad");
  check_gpu(tok_sorted, "tok_sorted");
  check_gpu(perm, "perm");
  TORCH_CHECK(grad.dim() == 2 && grad.scalar_type() == at::kBFloat16 && aligned16(grad), "embedding_backward: bf16 [rows, C] gradient expected");
  const int64_t rows = grad.size(0), C = grad.size(1);
  TORCH_CHECK(tok_sorted.scalar_type() == at::kLong && perm.scalar_type() == at::kLong && tok_sorted.numel() == rows && perm.numel() == rows,
              "embedding_backward: sorted ids and their source rows must be int64 of one entry per gradient row");
  TORCH_CHECK(kernels::embedding_supported(rows, T, C, V, P), "embedding_backward: unsupported shape");
  Tensor dwte = at::zeros({V, C}, grad.options());
  Tensor dwpe = at::zeros({P, C}, grad.options());
  DevGuard guard(grad);
  const int64_t chunks = kernels::embedding_bwd_chunks(rows);
  Tensor parts = at::empty({2, chunks, C}, grad.options().dtype(at::kFloat));
  Tensor head_tok = at::empty({chunks}, grad.options().dtype(at::kLong));
  kernels::launch_embedding_bwd(grad.data_ptr(), reinterpret_cast<const long long*>(tok_sorted.data_ptr<int64_t>()),
                                reinterpret_cast<const long long*>(perm.data_ptr<int64_t>()), dwte.data_ptr(), dwpe.data_ptr(), rows,
                                static_cast<int>(rows / T), static_cast<int>(T), static_cast<int>(C), V, parts.data_ptr<float>(),
                                parts.data_ptr<float>() + chunks * C, reinterpret_cast<long long*>(head_tok.data_ptr<int64_t>()),
                                stream_of(grad));
  return {dwte, dwpe};
}

}  // namespace

void init_ops(py::module_& m) {
  m.def("embedding_supported", &embedding_ok, py::arg("idx"), py::arg("wte"), py::arg("wpe"), py::arg("T"));
  m.def("embedding_forward", &embedding_forward, py::arg("idx"), py::arg("wte"), py::arg("wpe"), py::arg("T"));
  m.def("embedding_backward", &embedding_backward, py::arg("grad"), py::arg("tok_sorted"), py::arg("perm"), py::arg("T"), py::arg("V"), py::arg("P"));
  m.def("attention_forward", &attention_forward, py::arg("qkv"), py::arg("n_head"), py::arg("softmax_scale"), py::arg("kernel") = "auto");
  m.def("attention_backward", &attention_backward, py::arg("do"), py::arg("qkv"), py::arg("o"), py::arg("lse"), py::arg("n_head"),
        py::arg("softmax_scale"), py::arg("want_bias_grad") = false, py::arg("kernel") = "auto");
  m.def("gemm_bf16_split_k", [](int64_t M, int64_t N, int64_t K, bool a_kmajor, bool b_kmajor) {
    kernels::GemmArgs a{};
    a.M = M; a.N = N; a.K = K;
    a.a_kmajor = a_kmajor; a.b_kmajor = b_kmajor;
    a.lda = a_kmajor ? K : M; a.ldb = b_kmajor ? K : N; a.ldc = N;
    return kernels::gemm_bf16_split_k(a);
  }, "contraction ranges the bf16 GEMM would use for an [M,K]x[K,N] product (1 = single pass)");
  m.def("xent_fused", &xent_fused, py::arg("logits"), py::arg("targets"), py::arg("n_classes"), py::arg("ignore_index"), py::arg("mean"));
  m.def("dgrad_act_backward", &dgrad_act_backward, py::arg("dout"), py::arg("weight"), py::arg("aux"), py::arg("act"), py::arg("want_colsum"));
  m.def("dgrad_act_backward_supported", &dgrad_act_backward_supported);
  m.def("attention_supported", [](int64_t T, int64_t hd, int64_t rs) { return kernels::attention_supported(T, hd, rs); });
  m.def("multi_copy_scale", &multi_copy_scale, py::arg("srcs"), py::arg("dsts"), py::arg("scale") = 1.0);
  m.def("scale_inplace", &scale_inplace);
  m.def("adamw_tick", &adamw_tick);
  m.def("adamw_step", &adamw_step, py::arg("params"), py::arg("grads"), py::arg("exp_avgs"), py::arg("exp_avg_sqs"), py::arg("masters"),
        py::arg("hparams"), py::arg("grad_scale") = py::none());
  m.def("grad_sumsq", &grad_sumsq);
  m.def("clip_coef", &clip_coef);
  m.def("xent_forward", &xent_forward, py::arg("logits"), py::arg("targets"), py::arg("n_classes") = -1, py::arg("ignore_index") = -100);
  m.def("xent_backward", &xent_backward, py::arg("logits"), py::arg("targets"), py::arg("row_lse"), py::arg("grad_out"), py::arg("metrics"),
        py::arg("mean") = true, py::arg("n_classes") = -1, py::arg("ignore_index") = -100, py::arg("inplace") = false);
  m.def("gemm_supported", [](int64_t M, int64_t N, int64_t K, bool a_kmajor, bool b_kmajor, const std::string& dtype) {
    return gemm_shape_ok(M, N, K, a_kmajor, b_kmajor, dtype == "float32" ? kernels::kF32 : (dtype == "bfloat16" ? kernels::kBF16 : kernels::kF16));
  });
  m.def("linear_route", [](int64_t M, int64_t N, int64_t K, const std::string& dtype) {
    const int dt = dtype == "float32" ? kernels::kF32 : (dtype == "bfloat16" ? kernels::kBF16 : kernels::kF16);
    return linear_route(M, N, K, dt);
  });
  m.def("linear_forward", &linear_forward, py::arg("x"), py::arg("weight"), py::arg("bias"), py::arg("act") = 0, py::arg("save_pre_act") = false);
  m.def("linear_backward", &linear_backward, py::arg("grad"), py::arg("x"), py::arg("weight"), py::arg("need_dx") = true, py::arg("need_db") = true,
        py::arg("dw_out") = py::none());
  m.def("rank1_mlp_forward", &rank1_mlp_forward, py::arg("x"), py::arg("w1"), py::arg("b1"), py::arg("W2"), py::arg("b2"), py::arg("act") = 0);
  m.def("rank1_mlp_backward", &rank1_mlp_backward, py::arg("dy"), py::arg("x"), py::arg("w1"), py::arg("b1"), py::arg("W2"), py::arg("act") = 0);
  m.def("act_backward", &act_backward, py::arg("dy"), py::arg("saved"), py::arg("act"), py::arg("need_db") = false);
  m.def("colsum", &colsum);
  m.def("conv3x3_pad", &conv3x3_pad, py::arg("x_nhwc"));
  m.def("conv3x3_forward", &conv3x3_forward, py::arg("xpad"), py::arg("weight"), py::arg("B"), py::arg("H"), py::arg("W"));
  m.def("conv3x3_wgrad", &conv3x3_wgrad, py::arg("dypad"), py::arg("xpad"), py::arg("B"), py::arg("H"), py::arg("W"));
  m.def("gemm", &gemm, py::arg("A"), py::arg("B"), py::arg("a_kmajor") = true, py::arg("b_kmajor") = true, py::arg("bias") = py::none(),
        py::arg("act") = 0, py::arg("out") = py::none());
  m.def("batchnorm_supported", [](int64_t R, int64_t C, const std::string& dtype) {
    return kernels::batchnorm_supported(R, C, dtype == "float32" ? kernels::kF32 : (dtype == "bfloat16" ? kernels::kBF16 : kernels::kF16));
  });
  m.def("batchnorm_forward", &batchnorm_forward, py::arg("x"), py::arg("weight"), py::arg("bias"), py::arg("running_mean"), py::arg("running_var"),
        py::arg("momentum"), py::arg("eps"), py::arg("residual") = py::none(), py::arg("relu") = false);
  m.def("batchnorm_apply", &batchnorm_apply, py::arg("x"), py::arg("scale"), py::arg("shift"), py::arg("residual") = py::none(),
        py::arg("relu") = false);
  m.def("batchnorm_backward", &batchnorm_backward, py::arg("dy"), py::arg("x"), py::arg("y"), py::arg("scale_shift"), py::arg("weight"),
        py::arg("mean"), py::arg("rstd"), py::arg("relu") = false, py::arg("want_dres") = false);
  m.def("layernorm_supported", &kernels::layernorm_supported, py::arg("cols"));
  m.def("layernorm_forward", &layernorm_forward, py::arg("x"), py::arg("weight"), py::arg("bias"), py::arg("eps") = 1e-5,
        py::arg("res") = py::none(), py::arg("res_bias") = py::none());
  m.def("layernorm_backward", &layernorm_backward, py::arg("dy"), py::arg("x"), py::arg("weight"), py::arg("mean"), py::arg("rstd"),
        py::arg("dres") = py::none(), py::arg("want_dsum") = false);
  m.attr("ACT_NONE") = static_cast<int>(kernels::kActNone);
  m.attr("ACT_RELU") = static_cast<int>(kernels::kActRelu);
  m.attr("ACT_GELU") = static_cast<int>(kernels::kActGelu);
}
