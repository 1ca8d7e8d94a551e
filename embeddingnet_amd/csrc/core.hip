// Error reporting and version for libembnet_hip.so.
#include "common.h"
#include <atomic>
#include <mutex>
#include <vector>
#include "../../include/embnet.h"

namespace embnet {

char* last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

// ---- kernel trace ---------------------------------------------------------------------------------------------
struct TraceRec { char name[128]; int unit; double work, bytes; hipEvent_t e0, e1; };
static std::atomic<int> g_trace_on{0};
static std::mutex g_trace_mu;
static std::vector<TraceRec> g_trace;
static std::vector<hipEvent_t> g_event_pool;

bool trace_on() { return g_trace_on.load(std::memory_order_relaxed) != 0; }

static hipEvent_t take_event() {
  if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
  hipEvent_t e; (void)hipEventCreate(&e); return e;
}

TraceScope::TraceScope(const char* name, int unit, double work, void* stream, double bytes) : idx(-1), s((hipStream_t)stream) {
  if (!trace_on()) return;
  std::lock_guard<std::mutex> lk(g_trace_mu);
  TraceRec r;
  snprintf(r.name, sizeof r.name, "%s", name);
  r.unit = unit; r.work = work; r.bytes = bytes >= 0 ? bytes : (unit == TRACE_BYTES ? work : 0.0);
  r.e0 = take_event(); r.e1 = take_event();
  (void)hipEventRecord(r.e0, s);
  idx = (int)g_trace.size();
  g_trace.push_back(r);
}

TraceScope::~TraceScope() {
  if (idx < 0) return;
  std::lock_guard<std::mutex> lk(g_trace_mu);
  if (idx < (int)g_trace.size()) (void)hipEventRecord(g_trace[idx].e1, s);
}

}  // namespace embnet

extern "C" int embnet_trace_enable(int on) { return embnet::g_trace_on.exchange(on ? 1 : 0); }

extern "C" int embnet_trace_reset(void) {
  std::lock_guard<std::mutex> lk(embnet::g_trace_mu);
  for (auto& r : embnet::g_trace) { embnet::g_event_pool.push_back(r.e0); embnet::g_event_pool.push_back(r.e1); }
  embnet::g_trace.clear();
  return 0;
}

extern "C" int embnet_trace_count(void) {
  std::lock_guard<std::mutex> lk(embnet::g_trace_mu);
  return (int)embnet::g_trace.size();
}

extern "C" int embnet_trace_get(int i, char* name, int name_cap, float* ms, double* work, int* unit, double* bytes) {
  std::lock_guard<std::mutex> lk(embnet::g_trace_mu);
  if (i < 0 || i >= (int)embnet::g_trace.size() || !name || name_cap <= 0 || !ms || !work || !unit || !bytes)
    return embnet::fail(embnet::EMBNET_EINVAL, "trace_get: bad index or null pointer");
  const embnet::TraceRec& r = embnet::g_trace[i];
  snprintf(name, (size_t)name_cap, "%s", r.name);
  *work = r.work; *unit = r.unit; *bytes = r.bytes; *ms = 0.f;
  if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(ms, r.e0, r.e1) != hipSuccess)
    return embnet::fail(embnet::EMBNET_ELAUNCH, "trace_get: event not recorded");
  return 0;
}

extern "C" int embnet_abi_version(void) { return EMBNET_ABI_VERSION; }
extern "C" const char* embnet_last_error(void) { return embnet::last_error_buf(); }
