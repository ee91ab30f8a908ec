#include "comm_stream.h"

#include <chrono>
#include <map>
#include <mutex>
#include <sstream>
#include <stdexcept>

#include "kernels/kernels.h"

namespace dpa {

namespace {

#define DPA_CS_CHECK(expr)                                                                              \
  do {                                                                                                  \
    hipError_t _e = (expr);                                                                             \
    if (_e != hipSuccess) {                                                                             \
      std::ostringstream _os;                                                                           \
      _os << "HIP error " << hipGetErrorString(_e) << " at " << __FILE__ << ":" << __LINE__ << " in " #expr; \
      throw std::runtime_error(_os.str());                                                              \
    }                                                                                                   \
  } while (0)

constexpr double kHoldMs = 3.0;
constexpr int kCandidatesPerPriority = 6;  // more than the four pipes / the hardware-queue pool of a priority

std::mutex g_mu;
std::map<int, CommStreamChoice> g_chosen;  // per device, for the life of the process
std::map<int, float*> g_scratch;

float* scratch_for(int device) {
  auto it = g_scratch.find(device);
  if (it != g_scratch.end()) return it->second;
  float* p = nullptr;
  DPA_CS_CHECK(hipMalloc(&p, 256 * sizeof(float)));
  DPA_CS_CHECK(hipMemset(p, 0, 256 * sizeof(float)));
  g_scratch[device] = p;
  return p;
}

}  // namespace

std::string CommStreamChoice::describe() const {
  std::ostringstream os;
  os << (high_priority ? "high" : "normal") << "-priority stream, " << (independent ? "independent of" : "NOT independent of")
     << " the compute stream, " << candidates_tried << " candidate(s) probed [";
  for (size_t i = 0; i < probe_ms.size(); ++i) os << (i ? " " : "") << probe_ms[i];
  os << " ms]" << (reused ? ", reused" : "");
  return os.str();
}

double probe_stream_independence(int device, hipStream_t candidate, hipStream_t compute, double hold_ms) {
  float* scratch = scratch_for(device);
  hipEvent_t ev;
  DPA_CS_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  // Nothing may be pending on either stream: the test is about these two kernels only.
  DPA_CS_CHECK(hipStreamSynchronize(compute));
  DPA_CS_CHECK(hipStreamSynchronize(candidate));
  kernels::launch_spin(scratch, hold_ms, candidate);
  const auto t0 = std::chrono::steady_clock::now();
  kernels::launch_touch(scratch + 64, compute);
  DPA_CS_CHECK(hipEventRecord(ev, compute));
  DPA_CS_CHECK(hipEventSynchronize(ev));
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  DPA_CS_CHECK(hipStreamSynchronize(candidate));
  (void)hipEventDestroy(ev);
  return ms;
}

CommStreamChoice acquire_comm_stream(int device, hipStream_t compute, bool want_high) {
  std::lock_guard<std::mutex> g(g_mu);
  auto it = g_chosen.find(device);
  if (it != g_chosen.end()) {
    CommStreamChoice c = it->second;
    c.reused = true;
    return c;
  }
  int prev = -1;
  DPA_CS_CHECK(hipGetDevice(&prev));
  if (prev != device) DPA_CS_CHECK(hipSetDevice(device));
  int least = 0, greatest = 0;
  DPA_CS_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  CommStreamChoice out;
  std::vector<hipStream_t> rejected;
  hipStream_t first_normal = nullptr;
  for (int pass = want_high ? 0 : 1; pass < 2 && !out.stream; ++pass) {
    const bool high = pass == 0;
    for (int k = 0; k < kCandidatesPerPriority; ++k) {
      hipStream_t s = nullptr;
      DPA_CS_CHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high ? greatest : 0));
      const double ms = probe_stream_independence(device, s, compute, kHoldMs);
      out.probe_ms.push_back(ms);
      ++out.candidates_tried;
      if (ms < 0.5 * kHoldMs) {
        out.stream = s;
        out.high_priority = high;
        out.independent = true;
        break;
      }
      // Rejected candidates stay alive until the choice is made, so the next one gets another hardware queue.
      if (!high && !first_normal) first_normal = s;
      else rejected.push_back(s);
    }
  }
  if (!out.stream) {
    // Every candidate shared the compute stream's queue or pipe (a process at its hardware-queue limit):
    // a normal-priority stream is still correct, it just cannot overlap.
    out.stream = first_normal;
    first_normal = nullptr;
    out.high_priority = false;
    out.independent = false;
  }
  if (first_normal) rejected.push_back(first_normal);
  for (hipStream_t s : rejected) (void)hipStreamDestroy(s);
  if (prev != device && prev >= 0) (void)hipSetDevice(prev);
  g_chosen[device] = out;
  return out;
}

}  // namespace dpa
