// The communication stream of a process: one long-lived HIP stream per device, shared by every communicator
// the process creates, chosen so that it can make progress independently of the compute stream.
//
// Why this needs care (measured on MI355X with tools/kernel_harness/queue_pipe_probe.hip, HIP runtime only, and
// with the framework's own probe tools/diag_late_init.py; results under profiles/stall_*):
//   * Every HIP stream is mapped onto a hardware queue, and hardware queues are served by the command
//     processor's pipes in the order the process created them, four pipes, round robin: the 1st, 5th, 9th ...
//     hardware queue of a process share a pipe.
//   * Queues of DIFFERENT priority on one pipe are served strictly by priority: while the higher-priority
//     queue has a packet outstanding -- a running kernel or a barrier packet waiting for a signal -- the
//     lower-priority queue on that pipe is not served at all.  Queues of equal priority take turns.
//   * So a high-priority stream whose queue landed on the compute stream's pipe (a) serialises with it
//     instead of overlapping, and (b) deadlocks the first time it waits for an event the compute stream has
//     not reached yet: the waiting barrier packet keeps the pipe, the compute queue behind it can never
//     signal.  Whether that happens depends only on how many hardware queues existed before the stream was
//     created, which is why it looked like a dependence on "GPU work before the communicator".
// A normal-priority stream cannot get into (b).  It can still share the compute stream's hardware queue when
// the process is at its hardware-queue limit (GPU_MAX_HW_QUEUES), which is correct but not concurrent.
// Both cases show up in one bounded test that involves no cross-stream wait at all: a spin kernel runs on
// the candidate while the compute stream runs a one-lane kernel; if that small kernel finishes only when the
// spin does, the candidate is rejected and another stream (which gets the next hardware queue) is tried.
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

namespace dpa {

struct CommStreamChoice {
  hipStream_t stream = nullptr;
  bool high_priority = false;      // priority of the stream in use
  bool independent = false;        // the probe saw it run concurrently with the compute stream
  int candidates_tried = 0;        // streams probed until one passed (over both priorities)
  bool reused = false;             // an earlier communicator of this process already chose it
  std::vector<double> probe_ms;    // per candidate: time the compute stream's small kernel took
  std::string describe() const;
};

// want_high: ask for a high-priority stream first (falls back to normal priority when no high-priority
// candidate is independent of `compute`).  The first call per device probes and decides; later calls return
// the same stream.  The stream lives until the process ends.
CommStreamChoice acquire_comm_stream(int device, hipStream_t compute, bool want_high);

// The bounded independence probe by itself (exposed for tests): milliseconds the compute stream's small
// kernel needed while a `hold_ms` spin ran on `candidate`.
double probe_stream_independence(int device, hipStream_t candidate, hipStream_t compute, double hold_ms);

}  // namespace dpa
