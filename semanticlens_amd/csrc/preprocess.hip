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

__host__ __device__ inline int pad4(int v) { return (v + 3) & ~3; }
constexpr int kWideFlag = 1 << 30;  // in bounds[2 j + 1]: some tap of this row needs more than 24 signed bits

// grid (2, B): blockIdx.x = axis (0 horizontal, 1 vertical).  Per kept output index j in [0, S): bounds (first
// source index, tap count) and the taps, written to the image's coefficient block (16-byte aligned rows, zero padded):
//   [bounds: pad4(2 S) int32][taps: S x pad4(ksize) int32]
__global__ __launch_bounds__(256) void coeff_kernel(const int64_t* __restrict__ plan, int S, int interp,
                                                   int32_t* __restrict__ ws_coef) {
#pragma clang fp contract(off)
  const int64_t* p = plan + (int64_t)blockIdx.y * kPlanStride;
  const int axis = blockIdx.x;
  const int in_size = (int)(axis == 0 ? p[P_W] : p[P_H]);
  const int out_size = (int)(axis == 0 ? p[P_OW] : p[P_OH]);
  const int crop0 = (int)(axis == 0 ? p[P_LEFT] : p[P_TOP]);
  const int ksize = pad4((int)(axis == 0 ? p[P_KH] : p[P_KV]));
  int32_t* bounds = ws_coef + (axis == 0 ? p[P_COEF_H] : p[P_COEF_V]);
  int32_t* taps = bounds + pad4(2 * S);
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
    bool wide = false;  // a tap outside the signed 24-bit range: consumers take the 32-bit multiply path for this row
    for (int x = 0; x < xmax; ++x) {
      double v = tap_weight((x + xmin - center + 0.5) * ss, interp);
      if (ww != 0.0) v /= ww;
      const int kv = v < 0 ? (int)(-0.5 + v * (1 << kPrecisionBits)) : (int)(0.5 + v * (1 << kPrecisionBits));
      wide |= kv >= (1 << 23) || kv < -(1 << 23);
      k[x] = kv;
    }
    for (int x = xmax < 0 ? 0 : xmax; x < ksize; ++x) k[x] = 0;
    bounds[2 * j] = xmin;
    bounds[2 * j + 1] = (xmax < 0 ? 0 : xmax) | (wide ? kWideFlag : 0);
  }
}

__device__ inline uint32_t clip8(int v) {
  v >>= kPrecisionBits;
  return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

// 12 contiguous bytes (4 RGB pixels) at any byte alignment (gfx950 runs with unaligned global access enabled)
__device__ inline u32x3 load12(const uint8_t* p) {
  u32x3 v;
  __builtin_memcpy(&v, p, 12);
  return v;
}

// byte * tap: taps fit 24 signed bits except in flagged rows, so the full-rate 24-bit multiply is exact (the 32-bit
// integer multiply is a quarter-rate instruction and was the bottleneck of both passes)
template <bool WIDE>
__device__ inline int bmul(uint32_t byte, int k) {
  if (WIDE) return (int)byte * k;
  return __mul24((int)byte, k);
}

// acc[c] += sum over the 4 pixels in `px` of pixel[c] * k[tap]
template <bool WIDE>
__device__ inline void mac4(const u32x3 px, const int4 k, int& s0, int& s1, int& s2) {
  s0 += bmul<WIDE>(px.x & 0xff, k.x) + bmul<WIDE>(px.x >> 24, k.y) + bmul<WIDE>((px.y >> 16) & 0xff, k.z) + bmul<WIDE>((px.z >> 8) & 0xff, k.w);
  s1 += bmul<WIDE>((px.x >> 8) & 0xff, k.x) + bmul<WIDE>(px.y & 0xff, k.y) + bmul<WIDE>(px.y >> 24, k.z) + bmul<WIDE>((px.z >> 16) & 0xff, k.w);
  s2 += bmul<WIDE>((px.x >> 16) & 0xff, k.x) + bmul<WIDE>((px.y >> 8) & 0xff, k.y) + bmul<WIDE>(px.z & 0xff, k.z) + bmul<WIDE>(px.z >> 24, k.w);
}

// the four-taps-at-a-time body of the horizontal pass for R rows
template <bool WIDE, int R>
__device__ inline void hpass(const uint8_t* const (&rows)[R], const int32_t* __restrict__ k, int xmax, int (&acc)[R][3]) {
  int x = 0;
  for (; x + 4 <= xmax; x += 4) {
    const int4 kk = *reinterpret_cast<const int4*>(k + x);
#pragma unroll
    for (int r = 0; r < R; ++r) mac4<WIDE>(load12(rows[r] + 3 * x), kk, acc[r][0], acc[r][1], acc[r][2]);
  }
  if (x < xmax) {  // 1..3 taps left: window [xmax - 4, xmax), taps below x already done
    const int xs = xmax - 4, done = x - xs;
    int4 kk;
    kk.x = 0;  // done >= 1
    kk.y = done >= 2 ? 0 : k[xs + 1];
    kk.z = done >= 3 ? 0 : k[xs + 2];
    kk.w = k[xs + 3];
#pragma unroll
    for (int r = 0; r < R; ++r) mac4<WIDE>(load12(rows[r] + 3 * xs), kk, acc[r][0], acc[r][1], acc[r][2]);
  }
}

constexpr int kRowsPerThread = 4;

// grid (ceil(ceil(max_h / 4) * S / 256), B): thread = (four consecutive source rows, kept output column j); the
// launch is latency bound (one short dependent chain per thread), so a thread keeps the taps of its column in
// registers and applies them to four rows.  Taps are consumed four at a time (one 12-byte pixel load per row + one
// 16-byte coefficient load); a last partial group re-reads the final four taps with the already consumed ones
// masked, so nothing outside [xmin, xmin + xmax) is touched.
__global__ __launch_bounds__(256) void horizontal_kernel(const uint8_t* __restrict__ pixels, const int64_t* __restrict__ plan,
                                                         int S, const int32_t* __restrict__ ws_coef,
                                                         uint8_t* __restrict__ ws_tmp) {
  constexpr int R = kRowsPerThread;
  const int64_t* p = plan + (int64_t)blockIdx.y * kPlanStride;
  const int h = (int)p[P_H], w = (int)p[P_W];
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;  // 32-bit index math: the launch is bounded to 2^31 threads
  const uint32_t hb = (uint32_t)(h + R - 1) / R;
  if (t >= hb * (uint32_t)S) return;
  const int y0 = (int)(t / (uint32_t)S) * R, j = (int)(t % (uint32_t)S);
  const int nrow = h - y0 < R ? h - y0 : R;
  const int32_t* bounds = ws_coef + p[P_COEF_H];
  const int ksize = pad4((int)p[P_KH]);
  const int32_t* k = bounds + pad4(2 * S) + (int64_t)j * ksize;
  const int xmin = bounds[2 * j];
  const int xflag = bounds[2 * j + 1];
  const int xmax = xflag & ~kWideFlag;
  const int64_t row_bytes = (int64_t)w * 3;
  const uint8_t* src = pixels + p[P_OFF] + ((int64_t)y0 * w + xmin) * 3;
  int acc[R][3];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r][0] = acc[r][1] = acc[r][2] = 1 << (kPrecisionBits - 1);
  // rows past the image reuse the last valid row (never stored)
  const uint8_t* rows[R];
#pragma unroll
  for (int r = 0; r < R; ++r) rows[r] = src + (r < nrow ? r : nrow - 1) * row_bytes;
  if (xmax >= 4) {
    if (xflag & kWideFlag) hpass<true, R>(rows, k, xmax, acc);
    else hpass<false, R>(rows, k, xmax, acc);
  } else {
    for (int x = 0; x < xmax; ++x) {
      const int kv = k[x];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        acc[r][0] += (int)rows[r][3 * x] * kv;
        acc[r][1] += (int)rows[r][3 * x + 1] * kv;
        acc[r][2] += (int)rows[r][3 * x + 2] * kv;
      }
    }
  }
  uint8_t* dst = ws_tmp + p[P_TMP] + ((int64_t)y0 * S + j) * 3;
  if ((S & 3) == 0) {
    // four adjacent columns (lanes 4m..4m+3, same rows because S % 4 == 0) hold 12 contiguous bytes: lanes 0..2 of
    // the quad each assemble one aligned dword from their own 24 bits and the next lane's, lane 3 stores nothing
    const int q = threadIdx.x & 3;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t mine = clip8(acc[r][0]) | (clip8(acc[r][1]) << 8) | (clip8(acc[r][2]) << 16);
      const uint32_t next = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xF9 /* quad_perm [1,2,3,3] */, 0xF, 0xF, true);
      const uint32_t word = q == 0 ? (mine | (next << 24)) : q == 1 ? ((mine >> 8) | (next << 16)) : ((mine >> 16) | (next << 8));
      if (r < nrow && q != 3)
        *reinterpret_cast<uint32_t*>(dst + (int64_t)r * S * 3 - 3 * q + 4 * q) = word;
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (r < nrow) {
        dst[(int64_t)r * S * 3 + 0] = (uint8_t)clip8(acc[r][0]);
        dst[(int64_t)r * S * 3 + 1] = (uint8_t)clip8(acc[r][1]);
        dst[(int64_t)r * S * 3 + 2] = (uint8_t)clip8(acc[r][2]);
      }
    }
  }
}

// grid (ceil(S * ceil(S / 4) / 256), B): thread = (kept output row i, four adjacent kept output columns).  One 12-byte
// load per tap row feeds 12 accumulators; the normalised result leaves as one 16-byte store per channel.
__global__ __launch_bounds__(256) void vertical_kernel(const int64_t* __restrict__ plan, int S,
                                                       const int32_t* __restrict__ ws_coef,
                                                       const uint8_t* __restrict__ ws_tmp, Norm nm, float* __restrict__ out,
                                                       uint8_t* __restrict__ out_u8) {
  const int64_t b = blockIdx.y;
  const int64_t* p = plan + b * kPlanStride;
  const int S4 = (S + 3) >> 2;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= S * S4) return;
  const int i = t / S4, j = (t % S4) * 4;
  const int ncol = S - j < 4 ? S - j : 4;
  const int32_t* bounds = ws_coef + p[P_COEF_V];
  const int ksize = pad4((int)p[P_KV]);
  const int32_t* k = bounds + pad4(2 * S) + (int64_t)i * ksize;
  const int ymin = bounds[2 * i], yflag = bounds[2 * i + 1];
  const int ymax = yflag & ~kWideFlag;
  const bool wide = (yflag & kWideFlag) != 0;
  const uint8_t* src = ws_tmp + p[P_TMP] + ((int64_t)ymin * S + j) * 3;
  int acc[12];
#pragma unroll
  for (int q = 0; q < 12; ++q) acc[q] = 1 << (kPrecisionBits - 1);
  if (ncol == 4 && !wide) {
#pragma unroll 4
    for (int y = 0; y < ymax; ++y) {
      const int kv = k[y];
      const u32x3 v = load12(src + (int64_t)y * S * 3);
      const uint32_t wd[3] = {v.x, v.y, v.z};
#pragma unroll
      for (int q = 0; q < 12; ++q) acc[q] += bmul<false>((wd[q >> 2] >> (8 * (q & 3))) & 0xff, kv);
    }
  } else {
    for (int y = 0; y < ymax; ++y) {
      const int kv = k[y];
      const uint8_t* q8 = src + (int64_t)y * S * 3;
#pragma unroll
      for (int q = 0; q < 12; ++q)
        if (q < 3 * ncol) acc[q] += (int)q8[q] * kv;
    }
  }
  uint32_t u[12];
#pragma unroll
  for (int q = 0; q < 12; ++q) u[q] = clip8(acc[q]);  // q = 3 * column + channel
  if (out_u8) {
    uint8_t* d = out_u8 + ((b * S + i) * S + j) * 3;
    if (ncol == 4) {
      u32x3 pk;
      pk.x = u[0] | (u[1] << 8) | (u[2] << 16) | (u[3] << 24);
      pk.y = u[4] | (u[5] << 8) | (u[6] << 16) | (u[7] << 24);
      pk.z = u[8] | (u[9] << 8) | (u[10] << 16) | (u[11] << 24);
      __builtin_memcpy(d, &pk, 12);
    } else {
      for (int q = 0; q < 3 * ncol; ++q) d[q] = (uint8_t)u[q];
    }
  }
  if (out) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      // ToTensor (u / 255) then Normalize ((t - mean) / std): three correctly rounded fp32 operations
      float r[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const float tt = __fdiv_rn((float)u[3 * cc + c], 255.0f);
        r[cc] = __fdiv_rn(__fsub_rn(tt, nm.mean[c]), nm.stdv[c]);
      }
      float* o = out + ((b * 3 + c) * S + i) * S + j;
      if (ncol == 4 && (S & 3) == 0) {
        *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1], r[2], r[3]);
      } else {
        for (int cc = 0; cc < ncol; ++cc) o[cc] = r[cc];
      }
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
    coef += pad4(2 * S) + (int64_t)S * pad4((int)p[P_KH]);
    p[P_COEF_V] = coef;
    coef += pad4(2 * S) + (int64_t)S * pad4((int)p[P_KV]);
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
  hipLaunchKernelGGL(coeff_kernel, dim3(2, (unsigned)B), dim3(256), 0, st, d_plan, S, interp, coef);
  SL_CHECK_HIP(hipGetLastError());
  const int64_t hb = ((max_h + kRowsPerThread - 1) / kRowsPerThread * S + 255) / 256;
  SL_REQUIRE(hb < (1ll << 23), "sl_preprocess: image too tall");
  if (hb > 0) {
    hipLaunchKernelGGL(horizontal_kernel, dim3((unsigned)hb, (unsigned)B), dim3(256), 0, st, d_pixels, d_plan, S, coef, tmp);
    SL_CHECK_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(vertical_kernel, dim3((unsigned)(((int64_t)S * ((S + 3) / 4) + 255) / 256), (unsigned)B), dim3(256), 0, st, d_plan,
                     S, coef, tmp, nm, d_out, d_out_u8);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}
