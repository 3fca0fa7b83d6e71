// Error reporting + event-bracketed measurement for libsemanticlens_hip.so.
#include <mutex>
#include <vector>

#include "common.hpp"

namespace sl {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return SL_E_HIP;
}

struct ProfRec {
  hipEvent_t start, stop;
  int fam;
  double work;
};

static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_recs;       // live records
static std::vector<ProfRec> g_free_recs;  // events to reuse

ProfScope::ProfScope(int family, hipStream_t s, double work) : start(nullptr), stop(nullptr) {
  (void)s;
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r;
  if (!g_free_recs.empty()) {
    r = g_free_recs.back();
    g_free_recs.pop_back();
  } else {
    if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
  }
  r.fam = family;
  r.work = work;
  g_recs.push_back(r);
  start = r.start;
  stop = r.stop;
}

}  // namespace sl

using namespace sl;

SL_API const char* sl_last_error(void) { return g_err; }
SL_API int sl_abi_version(void) { return SL_ABI_VERSION; }

SL_API int sl_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return hip_fail(e, "hipGetDeviceCount");
  return n;
}

SL_API int sl_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
  return 0;
}

SL_API int sl_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_recs) {
    (void)hipEventSynchronize(r.stop);
    g_free_recs.push_back(r);
  }
  g_recs.clear();
  return 0;
}

SL_API int sl_prof_read(int family, double* total_ms, int64_t* launches, double* work) {
  SL_REQUIRE(family >= 0 && family < SL_PROF_NFAM, "sl_prof_read: bad family %d", family);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double ms = 0.0, w = 0.0;
  int64_t n = 0;
  for (auto& r : g_recs) {
    if (r.fam != family) continue;
    SL_CHECK_HIP(hipEventSynchronize(r.stop));
    float t = 0.f;
    SL_CHECK_HIP(hipEventElapsedTime(&t, r.start, r.stop));
    ms += t;
    w += r.work;
    ++n;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  if (work) *work = w;
  return 0;
}
