// K1 / K2 — activation reduce kernels (HBM-bound).
//
// Replaces the reference's aggregators (component_visualization/aggregators.py:38-244:
// `tensor.clone().flatten(2).amax(-1)` etc. followed by `.cpu()`) and the bf16 cast of
// ActMax.update (activation_caching.py:133).  One pass over the activation, no clone, no
// host round trip; output is the (B,C) candidate matrix the top-k merge consumes.
//
// Roofline: HBM.  Algorithmic bytes per launch = B*C*S*sizeof(act) read (+ B*C*2 written).
//
// Three code paths, chosen on the host from the strides:
//   rowreduce<G>  — rows contiguous (NCHW): the tensor is a flat stream of R = B*C rows of
//                   S floats.  G lanes own one row and read it as 16-byte pieces from the
//                   16-byte-aligned window that covers it (rows such as 7x7 = 196 B are not
//                   16-B aligned, so head/tail lanes mask by element index).  64/G rows share
//                   one 1-KiB wave-load; reduction across the G lanes is DPP.
//   colreduce     — reduced axis strided, component axis contiguous (tokens (B,T,F), or
//                   channels_last conv): lanes along F with 16-byte loads, the 4 waves of a
//                   workgroup split T and combine through LDS.
//   generic       — any strides / fp16 / bf16: one lane per output element.
#include "common.hpp"

namespace sl {
namespace {

enum Op : int { OP_MAX = 0, OP_SUM = 1, OP_ABSMAX = 2, OP_ABSSUM = 3 };

// ---- cross-lane helpers ------------------------------------------------------------------
template <int CTRL>
__device__ inline int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
// float -> int whose signed order equals the float order with +NaN on top
__device__ inline int f32_sort_key(float f) {
  int b = (int)f32_bits(f);
  return b ^ ((b >> 31) & 0x7FFFFFFF);
}
__device__ inline float sort_key_f32(int k) { return bits_f32((uint32_t)(k ^ ((k >> 31) & 0x7FFFFFFF))); }

template <bool SUM>
__device__ inline float combine(float a, float b) {
  if constexpr (SUM) return a + b;
  return sort_key_f32(max(f32_sort_key(a), f32_sort_key(b)));  // NaN-propagating max
}

// all-reduce over aligned groups of G lanes (G = 1,2,4,...,64)
template <int G, bool SUM>
__device__ inline float group_allreduce(float v) {
  if constexpr (SUM) {
    if constexpr (G >= 2) v += bits_f32((uint32_t)dpp_i32<0xB1>((int)f32_bits(v)));   // quad_perm [1,0,3,2]
    if constexpr (G >= 4) v += bits_f32((uint32_t)dpp_i32<0x4E>((int)f32_bits(v)));   // quad_perm [2,3,0,1]
    if constexpr (G >= 8) v += bits_f32((uint32_t)dpp_i32<0x141>((int)f32_bits(v)));  // row_half_mirror
    if constexpr (G >= 16) v += bits_f32((uint32_t)dpp_i32<0x140>((int)f32_bits(v))); // row_mirror
    if constexpr (G >= 32) v += __shfl_xor(v, 16, 64);
    if constexpr (G >= 64) v += __shfl_xor(v, 32, 64);
    return v;
  } else {
    int k = f32_sort_key(v);
    if constexpr (G >= 2) k = max(k, dpp_i32<0xB1>(k));
    if constexpr (G >= 4) k = max(k, dpp_i32<0x4E>(k));
    if constexpr (G >= 8) k = max(k, dpp_i32<0x141>(k));
    if constexpr (G >= 16) k = max(k, dpp_i32<0x140>(k));
    if constexpr (G >= 32) k = max(k, __shfl_xor(k, 16, 64));
    if constexpr (G >= 64) k = max(k, __shfl_xor(k, 32, 64));
    return sort_key_f32(k);
  }
}

// per-lane accumulator: running max ignores NaN (v_max_f32) and remembers it separately
template <int OP>
struct Acc {
  float v;
  bool nan;
  __device__ inline void init() {
    v = (OP == OP_SUM || OP == OP_ABSSUM) ? 0.f : -__builtin_huge_valf();
    nan = false;
  }
  __device__ inline void add(float x, bool valid) {
    if constexpr (OP == OP_ABSMAX || OP == OP_ABSSUM) x = __builtin_fabsf(x);
    if constexpr (OP == OP_SUM || OP == OP_ABSSUM) {
      v += valid ? x : 0.f;
    } else {
      x = valid ? x : -__builtin_huge_valf();
      nan |= (x != x);
      v = __builtin_fmaxf(v, x);
    }
  }
  __device__ inline float lane_value() const {
    if constexpr (OP == OP_SUM || OP == OP_ABSSUM) return v;
    return nan ? bits_f32(0x7FC00000u) : v;
  }
};

template <int OP>
__device__ inline float finish(float v, float count) {
  if constexpr (OP == OP_SUM || OP == OP_ABSSUM) return v / count;  // torch: sum / n
  return v;
}

__device__ inline void store_outputs(float r, int64_t idx, uint16_t* cand, float* outf) {
  if (outf) outf[idx] = r;
  if (cand) cand[idx] = f32_to_bf16_rne(r);
}

// ---- rowreduce: contiguous rows -------------------------------------------------------------
// x: 16-byte aligned, R rows of S floats back to back.
template <int G, int U, int OP>
__global__ __launch_bounds__(256) void rowreduce_kernel(const float* __restrict__ x, int64_t R, int S,
                                                         uint16_t* __restrict__ cand, float* __restrict__ outf) {
  constexpr int RPT = kWave / G;  // rows per task (one wave-load covers RPT rows)
  const int lane = threadIdx.x & 63;
  const int li = lane % G;
  const int g = lane / G;
  const int64_t total = R * (int64_t)S;
  const int64_t ntasks = (R + RPT - 1) / RPT;
  const int64_t nbatch = (ntasks + U - 1) / U;
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int nsteps = ((S + 6) / 4 + G - 1) / G;  // 16-byte pieces per row window, per lane
  const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);

  for (int64_t tb = wave0; tb < nbatch; tb += nwaves) {
    Acc<OP> acc[U];
    int64_t e0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc[u].init();
      int64_t r = (tb * U + u) * RPT + g;
      e0[u] = r < R ? r * (int64_t)S : -1;
    }
    for (int step = 0; step < nsteps; ++step) {
      const int q = step * G + li;  // piece index inside the row window
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e0[u] >= 0) {
          const int64_t p4 = (e0[u] >> 2) + q;  // global 16-byte piece index
          const int64_t pe = p4 << 2;
          const int h = (int)(e0[u] & 3);
          if (q * 4 - h < S) {  // piece overlaps the row
            if (pe + 4 <= total) {
              v[u] = x4[p4];
            } else {  // last piece of the tensor: stay in bounds
              float t[4] = {0.f, 0.f, 0.f, 0.f};
              for (int i = 0; i < 4; ++i)
                if (pe + i < total) t[i] = x[pe + i];
              v[u] = make_float4(t[0], t[1], t[2], t[3]);
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int h = (int)(e0[u] & 3);
        const int pos = q * 4 - h;  // row-local index of v.x
        const bool row_ok = e0[u] >= 0;
        acc[u].add(v[u].x, row_ok && (unsigned)(pos + 0) < (unsigned)S);
        acc[u].add(v[u].y, row_ok && (unsigned)(pos + 1) < (unsigned)S);
        acc[u].add(v[u].z, row_ok && (unsigned)(pos + 2) < (unsigned)S);
        acc[u].add(v[u].w, row_ok && (unsigned)(pos + 3) < (unsigned)S);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      constexpr bool SUM = (OP == OP_SUM || OP == OP_ABSSUM);
      float r = group_allreduce<G, SUM>(acc[u].lane_value());
      if (li == 0 && e0[u] >= 0) {
        r = finish<OP>(r, (float)S);
        store_outputs(r, (tb * U + u) * RPT + g, cand, outf);
      }
    }
  }
}

// ---- colreduce: out[b][f] = op_t x[b][t][f], f contiguous ------------------------------------
// One workgroup (4 waves) per (b, 256-float chunk of F); waves split T; LDS combine.
template <int OP>
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ x, int64_t B, int T, int64_t F,
                                                         int64_t sb, int64_t st, int t_begin, int t_end,
                                                         float denom, uint16_t* __restrict__ cand,
                                                         float* __restrict__ outf) {
  __shared__ float s_part[4][256];
  constexpr bool SUM = (OP == OP_SUM || OP == OP_ABSSUM);
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int64_t nchunk = (F + 255) / 256;
  const int64_t ntask = B * nchunk;
  for (int64_t task = blockIdx.x; task < ntask; task += gridDim.x) {
    const int64_t b = task / nchunk;
    const int64_t f0 = (task % nchunk) * 256 + lane * 4;
    Acc<OP> a0, a1, a2, a3;
    a0.init(); a1.init(); a2.init(); a3.init();
    const bool in = f0 < F;  // F % 4 == 0 on this path
    const float* base = x + b * sb + f0;
    if (in) {
      int t = t_begin + w;
#pragma unroll 1
      for (; t + 28 < t_end; t += 32) {  // 8 loads in flight per lane
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(base + (int64_t)(t + 4 * j) * st);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          a0.add(v[j].x, true); a1.add(v[j].y, true); a2.add(v[j].z, true); a3.add(v[j].w, true);
        }
      }
      for (; t < t_end; t += 4) {
        float4 v = *reinterpret_cast<const float4*>(base + (int64_t)t * st);
        a0.add(v.x, true); a1.add(v.y, true); a2.add(v.z, true); a3.add(v.w, true);
      }
    }
    s_part[w][lane * 4 + 0] = a0.lane_value();
    s_part[w][lane * 4 + 1] = a1.lane_value();
    s_part[w][lane * 4 + 2] = a2.lane_value();
    s_part[w][lane * 4 + 3] = a3.lane_value();
    __syncthreads();
    {
      const int f = threadIdx.x;  // 256 threads -> 256 features of the chunk
      const int64_t fg = (task % nchunk) * 256 + f;
      if (fg < F) {
        float r = s_part[0][f];
        r = combine<SUM>(r, s_part[1][f]);
        r = combine<SUM>(r, s_part[2][f]);
        r = combine<SUM>(r, s_part[3][f]);
        r = finish<OP>(r, denom);
        store_outputs(r, b * F + fg, cand, outf);
      }
    }
    __syncthreads();
  }
}

// ---- generic: any strides, fp32 / fp16 / bf16 -------------------------------------------------
template <typename T>
__device__ inline float load_as_f32(const void* p, int64_t i);
template <>
__device__ inline float load_as_f32<float>(const void* p, int64_t i) { return ((const float*)p)[i]; }
template <>
__device__ inline float load_as_f32<_Float16>(const void* p, int64_t i) { return (float)((const _Float16*)p)[i]; }
template <>
__device__ inline float load_as_f32<uint16_t>(const void* p, int64_t i) { return bf16_to_f32(((const uint16_t*)p)[i]); }

// round to the activation dtype first (the reference aggregates in that dtype), then report
template <typename T>
__device__ inline float round_to_dtype(float v) { return v; }
template <>
__device__ inline float round_to_dtype<_Float16>(float v) { return (float)(_Float16)v; }
template <>
__device__ inline float round_to_dtype<uint16_t>(float v) { return bf16_to_f32(f32_to_bf16_rne(v)); }

template <typename T, int OP>
__global__ __launch_bounds__(256) void generic_reduce_kernel(const void* __restrict__ x, int64_t B, int64_t C,
                                                              int64_t S, int64_t sb, int64_t sc, int64_t ss,
                                                              int64_t s_begin, int64_t s_end, float denom,
                                                              uint16_t* __restrict__ cand, float* __restrict__ outf) {
  const int64_t n = B * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / C, c = i % C;
    const int64_t off = b * sb + c * sc;
    Acc<OP> a;
    a.init();
    for (int64_t s = s_begin; s < s_end; ++s) a.add(load_as_f32<T>(x, off + s * ss), true);
    float r = finish<OP>(a.lane_value(), denom);
    r = round_to_dtype<T>(r);
    store_outputs(r, i, cand, outf);
  }
}

template <int G, int U, int OP>
void launch_rowreduce(const float* x, int64_t R, int S, uint16_t* cand, float* outf, hipStream_t st) {
  constexpr int RPT = kWave / G;
  const int64_t ntasks = (R + RPT - 1) / RPT;
  const int64_t nbatch = (ntasks + U - 1) / U;
  int64_t blocks = (nbatch + 3) / 4;
  const int64_t cap = (int64_t)num_cus() * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((rowreduce_kernel<G, U, OP>), dim3((unsigned)blocks), dim3(256), 0, st, x, R, S, cand, outf);
}

template <int OP>
void dispatch_rowreduce(const float* x, int64_t R, int S, uint16_t* cand, float* outf, hipStream_t st) {
  // pieces needed for a row window: up to (S + 6) / 4
  const int need = (S + 6) / 4;
  if (need <= 4) launch_rowreduce<4, 8, OP>(x, R, S, cand, outf, st);
  else if (need <= 8) launch_rowreduce<8, 8, OP>(x, R, S, cand, outf, st);
  else if (need <= 16) launch_rowreduce<16, 8, OP>(x, R, S, cand, outf, st);
  else if (need <= 32) launch_rowreduce<32, 8, OP>(x, R, S, cand, outf, st);
  else if (need <= 64) launch_rowreduce<64, 8, OP>(x, R, S, cand, outf, st);
  else launch_rowreduce<64, 4, OP>(x, R, S, cand, outf, st);
}

template <int OP>
void launch_colreduce(const float* x, int64_t B, int T, int64_t F, int64_t sb, int64_t st_, int t0, int t1,
                      float denom, uint16_t* cand, float* outf, hipStream_t st) {
  int64_t blocks = B * ((F + 255) / 256);
  const int64_t cap = (int64_t)num_cus() * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((colreduce_kernel<OP>), dim3((unsigned)blocks), dim3(256), 0, st, x, B, T, F, sb, st_, t0, t1,
                     denom, cand, outf);
}

template <typename T, int OP>
void launch_generic(const void* x, int64_t B, int64_t C, int64_t S, int64_t sb, int64_t sc, int64_t ss, int64_t s0,
                    int64_t s1, float denom, uint16_t* cand, float* outf, hipStream_t st) {
  int64_t blocks = (B * C + 255) / 256;
  const int64_t cap = (int64_t)num_cus() * 16;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((generic_reduce_kernel<T, OP>), dim3((unsigned)blocks), dim3(256), 0, st, x, B, C, S, sb, sc, ss,
                     s0, s1, denom, cand, outf);
}

template <int OP>
void dispatch_generic(const void* x, int dtype, int64_t B, int64_t C, int64_t S, int64_t sb, int64_t sc, int64_t ss,
                      int64_t s0, int64_t s1, float denom, uint16_t* cand, float* outf, hipStream_t st) {
  if (dtype == SL_F32) launch_generic<float, OP>(x, B, C, S, sb, sc, ss, s0, s1, denom, cand, outf, st);
  else if (dtype == SL_F16) launch_generic<_Float16, OP>(x, B, C, S, sb, sc, ss, s0, s1, denom, cand, outf, st);
  else launch_generic<uint16_t, OP>(x, B, C, S, sb, sc, ss, s0, s1, denom, cand, outf, st);
}

// (B, C, S) with strides -> (B, C): reduce over s in [s0, s1).  Picks the fastest legal path.
template <int OP>
int reduce_dispatch(const void* x, int dtype, int64_t B, int64_t C, int64_t S, int64_t sb, int64_t sc, int64_t ss,
                    int64_t s0, int64_t s1, uint16_t* cand, float* outf, hipStream_t st) {
  const float denom = (float)(s1 - s0);
  const bool aligned = ((uintptr_t)x & 15) == 0;
  const bool full = (s0 == 0 && s1 == S);
  if (dtype == SL_F32 && aligned && full && ss == 1 && sc == S && sb == C * S && S < (1 << 28)) {
    dispatch_rowreduce<OP>((const float*)x, B * C, (int)S, cand, outf, st);
  } else if (dtype == SL_F32 && aligned && sc == 1 && (C % 4) == 0 && (ss % 4) == 0 && (sb % 4) == 0 &&
             S < (1 << 30)) {
    launch_colreduce<OP>((const float*)x, B, (int)S, C, sb, ss, (int)s0, (int)s1, denom, cand, outf, st);
  } else {
    dispatch_generic<OP>(x, dtype, B, C, S, sb, sc, ss, s0, s1, denom, cand, outf, st);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "reduce kernel launch");
  return 0;
}

int dtype_size(int dtype) { return dtype == SL_F32 ? 4 : 2; }

}  // namespace
}  // namespace sl

using namespace sl;

SL_API int sl_reduce_conv(const void* d_act, int dtype, int64_t B, int64_t C, int64_t S, int64_t sb, int64_t sc,
                          int64_t ss, int agg, uint16_t* d_cand_bf16, float* d_out_f32, void* stream) {
  SL_REQUIRE(d_act || B * C * S == 0, "sl_reduce_conv: null activation");
  SL_REQUIRE(dtype >= SL_F32 && dtype <= SL_BF16, "sl_reduce_conv: bad dtype %d", dtype);
  SL_REQUIRE(B >= 0 && C >= 0 && S >= 0, "sl_reduce_conv: negative shape");
  SL_REQUIRE(agg == SL_CONV_MAX || agg == SL_CONV_MEAN, "sl_reduce_conv: bad agg %d", agg);
  SL_REQUIRE(d_cand_bf16 || d_out_f32, "sl_reduce_conv: no output");
  if (B * C == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof(SL_PROF_REDUCE, st, (double)B * C * S * dtype_size(dtype));
  if (agg == SL_CONV_MAX) return reduce_dispatch<OP_MAX>(d_act, dtype, B, C, S, sb, sc, ss, 0, S, d_cand_bf16, d_out_f32, st);
  return reduce_dispatch<OP_SUM>(d_act, dtype, B, C, S, sb, sc, ss, 0, S, d_cand_bf16, d_out_f32, st);
}

SL_API int sl_reduce_tokens(const void* d_act, int dtype, int64_t B, int64_t T, int64_t F, int64_t sb, int64_t st_,
                            int64_t sf, int agg, int64_t pos, uint16_t* d_cand_bf16, float* d_out_f32,
                            void* stream) {
  SL_REQUIRE(d_act || B * T * F == 0, "sl_reduce_tokens: null activation");
  SL_REQUIRE(dtype >= SL_F32 && dtype <= SL_BF16, "sl_reduce_tokens: bad dtype %d", dtype);
  SL_REQUIRE(B >= 0 && T >= 0 && F >= 0, "sl_reduce_tokens: negative shape");
  SL_REQUIRE(agg >= SL_TOK_MEAN && agg <= SL_TOK_TOKEN, "sl_reduce_tokens: bad agg %d", agg);
  SL_REQUIRE(d_cand_bf16 || d_out_f32, "sl_reduce_tokens: no output");
  if (B * F == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  // (B,T,F) reduced over T == (B, C=F, S=T) with sc = sf, ss = st
  int64_t t0 = 0, t1 = T;
  if (agg == SL_TOK_TOKEN) {
    int64_t p = pos < 0 ? pos + T : pos;
    SL_REQUIRE(p >= 0 && p < T, "sl_reduce_tokens: token position %lld out of range for T=%lld", (long long)pos,
               (long long)T);
    t0 = p;
    t1 = p + 1;
  }
  ProfScope prof(SL_PROF_REDUCE, st, (double)B * (t1 - t0) * F * dtype_size(dtype));
  switch (agg) {
    case SL_TOK_MEAN:
      return reduce_dispatch<OP_SUM>(d_act, dtype, B, F, T, sb, sf, st_, t0, t1, d_cand_bf16, d_out_f32, st);
    case SL_TOK_ABSMEAN:
      return reduce_dispatch<OP_ABSSUM>(d_act, dtype, B, F, T, sb, sf, st_, t0, t1, d_cand_bf16, d_out_f32, st);
    case SL_TOK_ABSMAX:
      return reduce_dispatch<OP_ABSMAX>(d_act, dtype, B, F, T, sb, sf, st_, t0, t1, d_cand_bf16, d_out_f32, st);
    default:  // max, and the single-token pick (max over one element is the element itself)
      return reduce_dispatch<OP_MAX>(d_act, dtype, B, F, T, sb, sf, st_, t0, t1, d_cand_bf16, d_out_f32, st);
  }
}
