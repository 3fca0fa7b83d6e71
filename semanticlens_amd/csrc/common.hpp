// Shared host/device helpers for libsemanticlens_hip.so (gfx950 only).
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/semanticlens_amd.h"

#define SL_API extern "C" __attribute__((visibility("default")))

namespace sl {

constexpr int kWave = 64;  // gfx950 wavefront

// ---- error plumbing -------------------------------------------------------------------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define SL_CHECK_HIP(expr)                                   \
  do {                                                       \
    hipError_t _e = (expr);                                  \
    if (_e != hipSuccess) return ::sl::hip_fail(_e, #expr);  \
  } while (0)

#define SL_REQUIRE(cond, ...)        \
  do {                               \
    if (!(cond)) {                   \
      ::sl::set_error(__VA_ARGS__);  \
      return SL_E_INVALID;           \
    }                                \
  } while (0)

// ---- per-dispatch timing (measurement, include/semanticlens_amd.h sl_prof_*) -------------
// When profiling is on, a ProfScope owns one (start, stop) event pair and SL_LAUNCH hands it to
// hipExtLaunchKernelGGL, which stamps the events with the dispatch's own begin/end times (the same
// timestamps rocprofv3's kernel trace reports) instead of bracketing the launch with stream events.
struct ProfScope {
  ProfScope(int family, hipStream_t s, double work);
  hipEvent_t start, stop;  // nullptr when profiling is off
};

#define SL_LAUNCH(prof, kernel, grid, block, shmem, stream, ...)                                             \
  do {                                                                                                       \
    if ((prof).start)                                                                                        \
      hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, (prof).start, (prof).stop, 0, __VA_ARGS__);  \
    else                                                                                                     \
      hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                   \
  } while (0)

// ---- variant switches (include/semanticlens_amd.h sl_set_option): which of several bit-identical kernel variants a dispatcher
// picks.  0 = the dispatcher's own rule.  Read on every dispatch (no latching): a test can force one variant after another in-process.
enum { OPT_G3_TILE = 0, OPT_F32_TILE, OPT_G3_STRIP_OFF, OPT_COLREDUCE_NW, OPT_COUNT };
int64_t option(int id);

inline int num_cus() {
  static int n = [] {
    int dev = 0, v = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
    return v > 0 ? v : 256;
  }();
  return n;
}

// ---- bf16 / ordering helpers (device + host) ----------------------------------------------
__host__ __device__ inline uint32_t f32_bits(float f) {
  union {
    float f;
    uint32_t u;
  } c;
  c.f = f;
  return c.u;
}
__host__ __device__ inline float bits_f32(uint32_t u) {
  union {
    float f;
    uint32_t u;
  } c;
  c.u = u;
  return c.f;
}

// fp32 -> bf16 round-to-nearest-even, NaN -> 0x7FC0: `acts.T.to(torch.bfloat16)`
// (activation_caching.py:133; c10::BFloat16 round_to_nearest_even).
__host__ __device__ inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u = f32_bits(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC0;
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__host__ __device__ inline float bf16_to_f32(uint16_t h) { return bits_f32((uint32_t)h << 16); }

// fp32 -> fp16 RNE and back (for activations that arrive in half precision: the reference
// aggregates in the tensor's dtype before the bf16 cast).
__device__ inline float round_through_f16(float f) { return (float)(_Float16)f; }

// Monotone 16-bit key of a bf16 bit pattern for "value descending" compares that match
// ATen's comparator `(isnan(a) && !isnan(b)) || a > b` (TopKImpl.h): NaN is the largest,
// -0.0 == +0.0, larger key == larger value.
__host__ __device__ inline uint32_t bf16_order_key(uint16_t h) {
  uint32_t mag = h & 0x7FFFu;
  if (mag > 0x7F80u) return 0xFFFFu;  // NaN
  return (h & 0x8000u) ? (0x8000u - mag) : (0x8000u + mag);
}

// total order: key desc, then id asc
__host__ __device__ inline bool better(uint32_t ka, int64_t ia, uint32_t kb, int64_t ib) {
  return ka > kb || (ka == kb && ia < ib);
}

}  // namespace sl
