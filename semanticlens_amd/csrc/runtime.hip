// Error reporting + event-bracketed measurement for libsemanticlens_hip.so.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
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

// ---- variant switches ---------------------------------------------------------------------------------------------------
static const char* const kOptNames[OPT_COUNT] = {"g3_tile", "f32_tile", "g3_strip_off", "colreduce_nw"};
static std::atomic<int64_t> g_opts[OPT_COUNT];
static std::once_flag g_opts_once;

static int opt_id(const char* name) {
  for (int i = 0; i < OPT_COUNT; ++i)
    if (name && strcmp(name, kOptNames[i]) == 0) return i;
  return -1;
}

static void opts_from_env() {  // SL_OPTIONS="g3_tile=128,colreduce_nw=8": the same switches for a whole process (subprocess tests)
  const char* e = getenv("SL_OPTIONS");
  if (!e) return;
  std::string all(e);
  size_t pos = 0;
  while (pos < all.size()) {
    size_t end = all.find(',', pos);
    if (end == std::string::npos) end = all.size();
    const std::string item = all.substr(pos, end - pos);
    const size_t eq = item.find('=');
    if (eq != std::string::npos) {
      const int id = opt_id(item.substr(0, eq).c_str());
      if (id >= 0) g_opts[id].store(atoll(item.c_str() + eq + 1));
    }
    pos = end + 1;
  }
}

int64_t option(int id) {
  std::call_once(g_opts_once, opts_from_env);
  return g_opts[id].load(std::memory_order_relaxed);
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

SL_API int sl_set_option(const char* name, int64_t value) {
  const int id = opt_id(name);
  SL_REQUIRE(id >= 0, "sl_set_option: unknown option '%s' (g3_tile, f32_tile, g3_strip_off, colreduce_nw)", name ? name : "(null)");
  (void)option(id);  // the environment is read first, once: an explicit call wins over it
  g_opts[id].store(value);
  return 0;
}

SL_API int64_t sl_get_option(const char* name) {
  const int id = opt_id(name);
  return id < 0 ? -1 : option(id);
}

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
