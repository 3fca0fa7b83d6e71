// K12 — image preprocessing on the device (second half).
//
// The reference preprocesses every sample on the host, one PIL image at a time, inside the DataLoader
// (`foundation_models/clip.py:157-163` -> the open_clip transform: Resize(S, BICUBIC) -> CenterCrop(S) -> ToTensor ->
// Normalize).  Here a ragged batch of raw RGB bytes is uploaded once and three kernels produce the (B, 3, S, S)
// fp32 encoder input:
//   coeff_kernel     per image and axis: Pillow's antialiased filter taps (Resample.c precompute_coeffs +
//                    normalize_coeffs_8bpc) for the S output columns/rows that survive the centre crop, in fp64 with
//                    contraction off, so the 22-bit fixed-point taps are the ones Pillow computes
//   horizontal_kernel  rows of the source -> (h, S, 3) 8-bit intermediate (Pillow's first pass, clip8 included)
//   vertical_kernel    second pass over the intermediate, then (u / 255 - mean) / std as three IEEE fp32 operations
// Results are bit-identical to PIL + torchvision (tests/golden/preprocess.npz).  Integer/byte work, bound by the
// read of the source pixels.
#include "common.hpp"

namespace sl {
namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;
constexpr int kPlanStride = SL_PP_PLAN_STRIDE;
enum { P_OFF = 0, P_H, P_W, P_OH, P_OW, P_TOP, P_LEFT, P_KH, P_KV, P_COEF_H, P_COEF_V, P_TMP };

struct Norm {
  float mean[3], stdv[3];
};

__host__ __device__ inline double filter_support(int interp) { return interp == SL_PP_BICUBIC ? 2.0 : 1.0;}

// ksize of Resample.c for an axis (host + device, plain IEEE double arithmetic)
__host__ __device__ inline int axis_ksize(int in_size, int out_size, int interp) {
#pragma clang fp contract(off)
  double scale = (double)(float)in_size / out_size;
  if (scale < 1.0) scale = 1.0;
  return (int)ceil(filter_support(interp) * scale) * 2 + 1;
}

__device__ inline double tap_weight(double x, int interp) {
#pragma clang fp contract(off)
  if (x < 0.0) x = -x;
  if (interp == SL_PP_BICUBIC) {
    const double a = -0.5;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
  }
  if (x < 1.0) return 1.0 - x;
  return 0.0;
}

// grid (2, B): blockIdx.x = axis (0 horizontal, 1 vertical).  Per kept output index j in [0, S): bounds (first
// source index, tap count) and ksize int32 taps, written to the image's coefficient block:
//   [bounds: S x 2 int32][taps: S x ksize int32]
__global__ __launch_bounds__(64) void coeff_kernel(const int64_t* __restrict__ plan, int S, int interp,
                                                   int32_t* __restrict__ ws_coef) {
#pragma clang fp contract(off)
  const int64_t* p = plan + (int64_t)blockIdx.y * kPlanStride;
  const int axis = blockIdx.x;
  const int in_size = (int)(axis == 0 ? p[P_W] : p[P_H]);
  const int out_size = (int)(axis == 0 ? p[P_OW] : p[P_OH]);
  const int crop0 = (int)(axis == 0 ? p[P_LEFT] : p[P_TOP]);
  const int ksize = (int)(axis == 0 ? p[P_KH] : p[P_KV]);
  int32_t* bounds = ws_coef + (axis == 0 ? p[P_COEF_H] : p[P_COEF_V]);
  int32_t* taps = bounds + 2 * S;
  const float in0 = 0.f, in1 = (float)in_size;
  double scale = (double)(in1 - in0) / out_size, filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = filter_support(interp) * filterscale;
  const double ss = 1.0 / filterscale;
  for (int j = threadIdx.x; j < S; j += blockDim.x) {
    const int xx = j + crop0;
    const double center = in0 + (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += tap_weight((x + xmin - center + 0.5) * ss, interp);
    int32_t* k = taps + (int64_t)j * ksize;
    for (int x = 0; x < xmax; ++x) {
      double v = tap_weight((x + xmin - center + 0.5) * ss, interp);
      if (ww != 0.0) v /= ww;
      k[x] = v < 0 ? (int)(-0.5 + v * (1 << kPrecisionBits)) : (int)(0.5 + v * (1 << kPrecisionBits));
    }
    for (int x = xmax < 0 ? 0 : xmax; x < ksize; ++x) k[x] = 0;
    bounds[2 * j] = xmin;
    bounds[2 * j + 1] = xmax;
  }
}

__device__ inline uint32_t clip8(int v) {
  v >>= kPrecisionBits;
  return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// grid (ceil(max_h * S / 256), B): thread = (source row y, kept output column j)
__global__ __launch_bounds__(256) void horizontal_kernel(const uint8_t* __restrict__ pixels, const int64_t* __restrict__ plan,
                                                         int S, const int32_t* __restrict__ ws_coef,
                                                         uint8_t* __restrict__ ws_tmp) {
  const int64_t* p = plan + (int64_t)blockIdx.y * kPlanStride;
  const int h = (int)p[P_H], w = (int)p[P_W];
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (int64_t)h * S) return;
  const int y = (int)(t / S), j = (int)(t % S);
  const int32_t* bounds = ws_coef + p[P_COEF_H];
  const int ksize = (int)p[P_KH];
  const int32_t* k = bounds + 2 * S + (int64_t)j * ksize;
  const int xmin = bounds[2 * j], xmax = bounds[2 * j + 1];
  const uint8_t* src = pixels + p[P_OFF] + ((int64_t)y * w + xmin) * 3;
  int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < xmax; ++x) {
    const int kv = k[x];
    s0 += (int)src[3 * x] * kv;
    s1 += (int)src[3 * x + 1] * kv;
    s2 += (int)src[3 * x + 2] * kv;
  }
  uint8_t* dst = ws_tmp + p[P_TMP] + t * 3;
  dst[0] = (uint8_t)clip8(s0);
  dst[1] = (uint8_t)clip8(s1);
  dst[2] = (uint8_t)clip8(s2);
}

// grid (ceil(S * S / 256), B): thread = (kept output row i, kept output column j)
__global__ __launch_bounds__(256) void vertical_kernel(const int64_t* __restrict__ plan, int S,
                                                       const int32_t* __restrict__ ws_coef,
                                                       const uint8_t* __restrict__ ws_tmp, Norm nm, float* __restrict__ out,
                                                       uint8_t* __restrict__ out_u8) {
  const int64_t b = blockIdx.y;
  const int64_t* p = plan + b * kPlanStride;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= S * S) return;
  const int i = t / S, j = t % S;
  const int32_t* bounds = ws_coef + p[P_COEF_V];
  const int ksize = (int)p[P_KV];
  const int32_t* k = bounds + 2 * S + (int64_t)i * ksize;
  const int ymin = bounds[2 * i], ymax = bounds[2 * i + 1];
  const uint8_t* src = ws_tmp + p[P_TMP] + ((int64_t)ymin * S + j) * 3;
  int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
  for (int y = 0; y < ymax; ++y) {
    const int kv = k[y];
    const uint8_t* q = src + (int64_t)y * S * 3;
    s0 += (int)q[0] * kv;
    s1 += (int)q[1] * kv;
    s2 += (int)q[2] * kv;
  }
  const uint32_t u[3] = {clip8(s0), clip8(s1), clip8(s2)};
  if (out_u8) {
    uint8_t* d = out_u8 + ((b * S + i) * S + j) * 3;
    d[0] = (uint8_t)u[0];
    d[1] = (uint8_t)u[1];
    d[2] = (uint8_t)u[2];
  }
  if (out) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      // ToTensor (u / 255) then Normalize ((t - mean) / std): three correctly rounded fp32 operations
      const float tt = __fdiv_rn((float)u[c], 255.0f);
      const float dd = __fsub_rn(tt, nm.mean[c]);
      out[((b * 3 + c) * S + i) * S + j] = __fdiv_rn(dd, nm.stdv[c]);
    }
  }
}

inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

}  // namespace
}  // namespace sl

using namespace sl;

SL_API int sl_preprocess_plan(const int32_t* h_hw, const int64_t* h_pixel_offsets, int64_t B, int S, int resize_mode,
                              int interp, int64_t* h_plan, int64_t* h_info) {
  SL_REQUIRE(B >= 0 && S >= 1, "sl_preprocess_plan: bad shape");
  SL_REQUIRE(resize_mode == SL_PP_SHORTEST || resize_mode == SL_PP_SQUASH, "sl_preprocess_plan: unknown resize mode");
  SL_REQUIRE(interp == SL_PP_BICUBIC || interp == SL_PP_BILINEAR, "sl_preprocess_plan: unknown interpolation");
  SL_REQUIRE(h_info && (B == 0 || (h_hw && h_plan)), "sl_preprocess_plan: null pointer");
  int64_t coef = 0, pix = 0, mh = 0;
  size_t tmp = 0;
  for (int64_t b = 0; b < B; ++b) {
    const int64_t h = h_hw[2 * b], w = h_hw[2 * b + 1];
    SL_REQUIRE(h >= 1 && w >= 1 && h < (1 << 24) && w < (1 << 24), "sl_preprocess_plan: image %lld has size %lld x %lld",
               (long long)b, (long long)h, (long long)w);
    int64_t oh = S, ow = S;
    if (resize_mode == SL_PP_SHORTEST) {  // torchvision Resize(int): shorter edge -> S, other edge int(S * long / short)
      const int64_t shrt = w <= h ? w : h, lng = w <= h ? h : w;
      const int64_t new_long = (int64_t)((double)(S * lng) / (double)shrt);
      if (w <= h) oh = new_long;
      else ow = new_long;
    }
    auto crop = [](int64_t size, int64_t c) {  // int(round((size - c) / 2.0)), round half to even
      const int64_t d = size - c, fl = d >= 0 ? d / 2 : -((-d + 1) / 2);
      if ((d & 1) == 0) return fl;
      return (fl & 1) ? fl + 1 : fl;
    };
    int64_t* p = h_plan + b * kPlanStride;
    for (int i = 0; i < kPlanStride; ++i) p[i] = 0;
    p[P_OFF] = h_pixel_offsets ? h_pixel_offsets[b] : pix;
    p[P_H] = h;
    p[P_W] = w;
    p[P_OH] = oh;
    p[P_OW] = ow;
    SL_REQUIRE(oh >= S && ow >= S, "sl_preprocess_plan: resized image %lld smaller than the crop", (long long)b);
    p[P_TOP] = crop(oh, S);
    p[P_LEFT] = crop(ow, S);
    p[P_KH] = axis_ksize((int)w, (int)ow, interp);
    p[P_KV] = axis_ksize((int)h, (int)oh, interp);
    p[P_COEF_H] = coef;
    coef += (int64_t)S * (2 + p[P_KH]);
    p[P_COEF_V] = coef;
    coef += (int64_t)S * (2 + p[P_KV]);
    p[P_TMP] = (int64_t)tmp;
    tmp += align16((size_t)h * (size_t)S * 3);
    pix += h * w * 3;
    if (h > mh) mh = h;
  }
  // workspace: [coefficients int32][intermediate bytes]; P_TMP offsets are relative to the second region
  h_info[SL_PP_INFO_COEF_BYTES] = (int64_t)align16((size_t)coef * 4);
  h_info[SL_PP_INFO_WS_BYTES] = h_info[SL_PP_INFO_COEF_BYTES] + (int64_t)tmp;
  h_info[SL_PP_INFO_MAX_H] = mh;
  h_info[SL_PP_INFO_PIXEL_BYTES] = pix;
  return 0;
}

SL_API int sl_preprocess(const uint8_t* d_pixels, const int64_t* d_plan, int64_t B, int S, int interp, int64_t max_h,
                         int64_t coef_bytes, const float* h_mean, const float* h_std, float* d_out, uint8_t* d_out_u8,
                         void* d_ws, size_t ws_bytes, void* stream) {
  SL_REQUIRE(B >= 0 && S >= 1 && max_h >= 0, "sl_preprocess: bad shape");
  SL_REQUIRE(interp == SL_PP_BICUBIC || interp == SL_PP_BILINEAR, "sl_preprocess: unknown interpolation");
  if (B == 0) return 0;
  SL_REQUIRE(d_pixels && d_plan && d_ws && (d_out || d_out_u8), "sl_preprocess: null pointer");
  SL_REQUIRE(!d_out || (h_mean && h_std), "sl_preprocess: mean/std required for the float output");
  SL_REQUIRE(coef_bytes >= 0 && (size_t)coef_bytes <= ws_bytes && (coef_bytes & 15) == 0, "sl_preprocess: bad workspace split");
  SL_REQUIRE(B <= 65535, "sl_preprocess: at most 65535 images per call");
  hipStream_t st = (hipStream_t)stream;
  int32_t* coef = static_cast<int32_t*>(d_ws);
  uint8_t* tmp = static_cast<uint8_t*>(d_ws) + coef_bytes;
  Norm nm{};
  for (int c = 0; c < 3; ++c) {
    nm.mean[c] = h_mean ? h_mean[c] : 0.f;
    nm.stdv[c] = h_std ? h_std[c] : 1.f;
  }
  hipLaunchKernelGGL(coeff_kernel, dim3(2, (unsigned)B), dim3(64), 0, st, d_plan, S, interp, coef);
  SL_CHECK_HIP(hipGetLastError());
  const int64_t hb = (max_h * S + 255) / 256;
  SL_REQUIRE(hb < (1ll << 31), "sl_preprocess: image too tall");
  if (hb > 0) {
    hipLaunchKernelGGL(horizontal_kernel, dim3((unsigned)hb, (unsigned)B), dim3(256), 0, st, d_pixels, d_plan, S, coef, tmp);
    SL_CHECK_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(vertical_kernel, dim3((unsigned)(((int64_t)S * S + 255) / 256), (unsigned)B), dim3(256), 0, st, d_plan,
                     S, coef, tmp, nm, d_out, d_out_u8);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}
