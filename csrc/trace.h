// Profiler ranges (roctx) for rocprofv3 --marker-trace, resolved at run time.
//
// The marker library is looked up with dlopen the first time tracing is switched on
// (rocprofiler-sdk's roctx first, the roctracer one as fallback), so the extension neither links
// against a profiler library nor pays anything when tracing is off: push/pop are one predictable
// branch.  Switched on by DIST_TRACE=1 in the environment or trace::enable(true).
#pragma once

#include <dlfcn.h>

#include <atomic>
#include <cstdlib>
#include <string>

namespace dpa {
namespace trace {

struct Api {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  void (*mark)(const char*) = nullptr;
  bool loaded = false;
};

inline Api& api() {
  static Api a;
  return a;
}

inline std::atomic<int>& state() {  // -1 = not decided yet, 0 = off, 1 = on
  static std::atomic<int> s{-1};
  return s;
}

inline bool load() {
  Api& a = api();
  if (a.loaded) return a.push != nullptr;
  a.loaded = true;
  for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
    void* h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
    if (!h) continue;
    a.push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    a.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    a.mark = reinterpret_cast<void (*)(const char*)>(dlsym(h, "roctxMarkA"));
    if (a.push && a.pop) return true;
    a.push = nullptr;
    a.pop = nullptr;
  }
  return false;
}

inline bool enable(bool on) {
  if (on && !load()) {
    state().store(0);
    return false;
  }
  state().store(on ? 1 : 0);
  return on;
}

inline bool enabled() {
  int s = state().load(std::memory_order_relaxed);
  if (s < 0) {
    const char* e = std::getenv("DIST_TRACE");
    enable(e && e[0] == '1');
    s = state().load();
  }
  return s == 1;
}

inline void push(const char* name) {
  if (enabled()) api().push(name);
}
inline void push(const std::string& name) { push(name.c_str()); }
inline void pop() {
  if (enabled()) api().pop();
}

struct Range {
  bool on;
  explicit Range(const std::string& name) : on(enabled()) {
    if (on) api().push(name.c_str());
  }
  ~Range() {
    if (on) api().pop();
  }
};

}  // namespace trace
}  // namespace dpa
