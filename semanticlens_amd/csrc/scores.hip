// K7 / K8 / K10 — score reductions of semanticlens/scores.py and the template mean of lens.py.
//
//   clarity_score     (scores.py:18-47)   V (C,n,D)  -> (C)     HBM-bound: C*n*D*4 bytes read once
//   redundancy_score  (scores.py:50-81)   V (Bt,C,D) -> (Bt)    K6's MFMA GEMM + row-max epilogue
//   template mean     (lens.py:196-199)   E (Q*T,D), E0 (T,D) -> (Q,D)
#include "common.hpp"

#include <cstdlib>
#include <cstring>

namespace sl {

int cosine_matrix_nt(const float* A, int64_t M, const float* B, int64_t N, int64_t K, float* ra, float* rb, float* out,
                     void* split, hipStream_t st);
size_t cosine_split_bytes(int64_t M, int64_t N, int64_t K);

namespace {

__device__ inline float block_sum_256(float v, float* s_red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) s_red[w] = v;
  __syncthreads();
  const float r = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  __syncthreads();
  return r;
}

// One workgroup per component.  Pass 1: each wave computes 1/max(||v_j||,1e-12) for its rows
// (F.normalize, scores.py:45).  Pass 2 re-reads the (L2-resident) rows: thread f accumulates
// mean_j v_j[f]/||v_j||, squares, block-sums.  clarity = ((sum m^2) - 1/n) / (n-1) * n  (:46).
__global__ __launch_bounds__(256) void clarity_kernel(const float* __restrict__ V, int64_t C, int n, int64_t D,
                                                       float* __restrict__ out) {
  extern __shared__ float s_rn[];  // n inverse norms + 4 reduction slots
  float* s_red = s_rn + n;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
    const float* Vc = V + c * (int64_t)n * D;
    for (int j = w; j < n; j += 4) {
      const float* p = Vc + (int64_t)j * D;
      float s = 0.f;
      for (int64_t i = lane; i < D; i += 64) s += p[i] * p[i];
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      if (lane == 0) s_rn[j] = 1.f / fmaxf(sqrtf(s), 1e-12f);
    }
    __syncthreads();
    float part = 0.f;
    for (int64_t f = threadIdx.x; f < D; f += 256) {
      float m = 0.f;
      for (int j = 0; j < n; ++j) m += Vc[(int64_t)j * D + f] * s_rn[j];
      m = m / (float)n;
      part += m * m;
    }
    const float tot = block_sum_256(part, s_red);
    if (threadIdx.x == 0) out[c] = (tot - 1.f / (float)n) / (float)(n - 1) * (float)n;
    __syncthreads();
  }
}

// ---- clarity, one pass (round 5) -------------------------------------------------------------------------------------------
// The kernel above reads every row twice (norms, then the weighted column sums; the second pass from L2) with 4-byte loads, one
// launch per layer: 0.15 of the HBM bound on configs[4]'s four 2.4-19 MB layers (round-4 driver line).  Here a component's
// (n, D) slab is read ONCE in 16-byte pieces: wave w owns rows w, w + 4, ...; all loads of a batch of rows are issued before
// any is used; a row's inverse norm needs only that row (one xor-shuffle tree), so the wave folds its rows into weighted column
// sums in registers and the four waves' partial sums meet in LDS.  Several layers (same n and D, any C) share one launch:
// `ClaritySources` carries up to 32 (input, output) pointer pairs, components are numbered across layers.
constexpr int kMaxClarityLayers = 32;
struct ClaritySources {
  const float* V[kMaxClarityLayers];
  float* out[kMaxClarityLayers];
  int64_t start[kMaxClarityLayers + 1];  // first global component of each layer
  int n_layers;
};

template <int PPL>  // 16-byte pieces per lane and row: ceil(D / 256)
__global__ __launch_bounds__(256) void clarity_multi_kernel(ClaritySources src, int n, int D) {
  constexpr int RB = (16 / PPL) < 1 ? 1 : ((16 / PPL) > 8 ? 8 : (16 / PPL));  // rows per batch: <= 16 loads in flight per lane
  __shared__ float s_m[4][PPL * 256];
  __shared__ float s_red[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int pieces = D / 4;
  const int64_t total = src.start[src.n_layers];
  int l = 0;
  for (int64_t c = blockIdx.x; c < total; c += gridDim.x) {
    while (l + 1 < src.n_layers && c >= src.start[l + 1]) ++l;  // components ascend within a workgroup
    const float4* Vc = reinterpret_cast<const float4*>(src.V[l] + (c - src.start[l]) * (int64_t)n * D);
    float4 acc[PPL];
#pragma unroll
    for (int p = 0; p < PPL; ++p) acc[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j0 = w; j0 < n; j0 += 4 * RB) {
      float4 v[RB][PPL];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int j = j0 + 4 * r;
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
          const int q = p * 64 + lane;
          v[r][p] = (j < n && q < pieces) ? Vc[(int64_t)j * pieces + q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < PPL; ++p) s += v[r][p].x * v[r][p].x + v[r][p].y * v[r][p].y + v[r][p].z * v[r][p].z + v[r][p].w * v[r][p].w;
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        const float rn = 1.f / fmaxf(sqrtf(s), 1e-12f);  // F.normalize, scores.py:45 (rows past n are all zero: they add nothing)
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
          acc[p].x += v[r][p].x * rn;
          acc[p].y += v[r][p].y * rn;
          acc[p].z += v[r][p].z * rn;
          acc[p].w += v[r][p].w * rn;
        }
      }
    }
#pragma unroll
    for (int p = 0; p < PPL; ++p) *reinterpret_cast<float4*>(&s_m[w][(p * 64 + lane) * 4]) = acc[p];
    __syncthreads();
    float part = 0.f;
    for (int f = threadIdx.x; f < PPL * 256; f += 256) {  // columns past D hold zeros
      const float m = (s_m[0][f] + s_m[1][f] + s_m[2][f] + s_m[3][f]) / (float)n;
      part += m * m;
    }
    const float tot = block_sum_256(part, s_red);  // ends with a barrier: s_m is free again
    if (threadIdx.x == 0) src.out[l][c - src.start[l]] = (tot - 1.f / (float)n) / (float)(n - 1) * (float)n;
  }
}

// row max of (cos - 2*I) — scores.py:79-80; one wave per row
__global__ __launch_bounds__(256) void rowmax_offdiag_kernel(const float* __restrict__ sims, int64_t C,
                                                              float* __restrict__ rowmax) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t r = wave; r < C; r += nw) {
    // torch's .max propagates NaN (scores.py:79), fmaxf drops it: remember it separately
    float m = -__builtin_huge_valf();
    bool nan = false;
    for (int64_t j = lane; j < C; j += 64) {
      float v = sims[r * C + j];
      if (j == r) v -= 2.f;
      nan |= (v != v);
      m = fmaxf(m, v);
    }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    const bool any_nan = __any(nan);
    if (lane == 0) rowmax[r] = any_nan ? __builtin_nanf("") : m;
  }
}

// mean of C values, single workgroup, fixed order -> deterministic
__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ v, int64_t C, float* __restrict__ out) {
  __shared__ float s_red[4];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < C; i += 256) s += v[i];
  const float tot = block_sum_256(s, s_red);
  if (threadIdx.x == 0) *out = tot / (float)C;
}

__global__ __launch_bounds__(256) void template_mean_kernel(const float* __restrict__ E, const float* __restrict__ E0,
                                                             int64_t Q, int T, int64_t D, float* __restrict__ out) {
  const int64_t n = Q * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t q = i / D, d = i % D;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += E[(q * T + t) * D + d] - E0[(int64_t)t * D + d];
    out[i] = s / (float)T;
  }
}

size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

}  // namespace
}  // namespace sl

using namespace sl;

static int clarity_two_pass(const float* d_V, int64_t C, int64_t n, int64_t D, float* d_out, hipStream_t st) {
  ProfScope prof(SL_PROF_SCORES, st, (double)C * n * D * 4);
  int64_t blocks = C;
  const int64_t cap = (int64_t)num_cus() * 8;
  if (blocks > cap) blocks = cap;
  SL_LAUNCH(prof, clarity_kernel, dim3((unsigned)blocks), dim3(256), (size_t)(n + 4) * 4, st, d_V, C, (int)n, D, d_out);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

SL_API int sl_clarity_multi(const float* const* h_d_Vs, const int64_t* h_Cs, int L, int64_t n, int64_t D, float* const* h_d_outs,
                            void* stream) {
  SL_REQUIRE(L >= 0 && n >= 0 && D >= 0, "sl_clarity_multi: negative shape");
  if (L == 0) return 0;
  SL_REQUIRE(h_d_Vs && h_Cs && h_d_outs, "sl_clarity_multi: null pointer array");
  SL_REQUIRE(n >= 1 && n <= 8192, "sl_clarity: n_samples=%lld not in [1, 8192]", (long long)n);
  hipStream_t st = (hipStream_t)stream;
  bool fast = D >= 4 && (D % 4) == 0 && D <= 2048;
  for (int i = 0; i < L; ++i) {
    SL_REQUIRE(h_Cs[i] >= 0, "sl_clarity_multi: negative component count");
    SL_REQUIRE(h_Cs[i] == 0 || (h_d_Vs[i] && h_d_outs[i]), "sl_clarity: null pointer");
    fast = fast && (((uintptr_t)h_d_Vs[i]) & 15) == 0;
  }
  if (!fast) {  // odd widths / unaligned slabs: the two-pass kernel, layer by layer
    for (int i = 0; i < L; ++i)
      if (h_Cs[i] > 0) {
        const int rc = clarity_two_pass(h_d_Vs[i], h_Cs[i], n, D, h_d_outs[i], st);
        if (rc) return rc;
      }
    return 0;
  }
  for (int i0 = 0; i0 < L; i0 += kMaxClarityLayers) {
    ClaritySources src;
    src.n_layers = 0;
    int64_t total = 0;
    for (int i = i0; i < L && i < i0 + kMaxClarityLayers; ++i) {
      if (h_Cs[i] == 0) continue;
      src.V[src.n_layers] = h_d_Vs[i];
      src.out[src.n_layers] = h_d_outs[i];
      src.start[src.n_layers] = total;
      total += h_Cs[i];
      ++src.n_layers;
    }
    if (total == 0) continue;
    src.start[src.n_layers] = total;
    ProfScope prof(SL_PROF_SCORES, st, (double)total * n * D * 4);
    // grid (tools/clarity_lab.py, round 6): up to two rounds of resident workgroups (8 per CU) the components are walked by a
    // resident grid (2 880 components x 40 KB: 25.5 us against 27.1 with one workgroup per component); beyond that one workgroup per
    // component and the dispatcher balances (9 216 x 92 KB: 139.6 us = 0.76 of 8 TB/s against 150.4).  Fetching a component ahead
    // of the one being reduced (two register buffers, two waves per SIMD) lost at both sizes: 31.9 / 148.8 us.
    int64_t blocks = total;
    const int64_t cap = (int64_t)num_cus() * 8;
    if (blocks > cap && blocks <= 2 * cap) blocks = cap;
    const int ppl = (int)((D / 4 + 63) / 64);
#define SL_CLARITY(P_)                                                                                                     \
  case P_:                                                                                                                  \
    SL_LAUNCH(prof, clarity_multi_kernel<P_>, dim3((unsigned)blocks), dim3(256), 0, st, src, (int)n, (int)D);              \
    break
    switch (ppl) {
      SL_CLARITY(1);
      SL_CLARITY(2);
      SL_CLARITY(3);
      SL_CLARITY(4);
      SL_CLARITY(5);
      SL_CLARITY(6);
      SL_CLARITY(7);
      default:
        SL_LAUNCH(prof, clarity_multi_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, src, (int)n, (int)D);
    }
#undef SL_CLARITY
    SL_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

SL_API int sl_clarity(const float* d_V, int64_t C, int64_t n, int64_t D, float* d_out, void* stream) {
  SL_REQUIRE(C >= 0 && n >= 0 && D >= 0, "sl_clarity: negative shape");
  if (C == 0) return 0;
  SL_REQUIRE(d_V && d_out, "sl_clarity: null pointer");
  return sl_clarity_multi(&d_V, &C, 1, n, D, &d_out, stream);
}

SL_API size_t sl_redundancy_ws_bytes(int64_t Bt, int64_t C, int64_t D) {
  (void)Bt;
  (void)D;
  return align256((size_t)C * 4) * 2 + align256((size_t)C * (size_t)C * 4) + cosine_split_bytes(C, C, D) + 256;
}

SL_API int sl_redundancy(const float* d_V, int64_t Bt, int64_t C, int64_t D, float* d_out, void* d_ws, size_t ws_bytes,
                         void* stream) {
  SL_REQUIRE(Bt >= 0 && C >= 0 && D >= 0, "sl_redundancy: negative shape");
  if (Bt == 0) return 0;
  SL_REQUIRE(C >= 1, "sl_redundancy: needs at least one component");  // torch: max over an empty dim raises
  SL_REQUIRE(d_V && d_out, "sl_redundancy: null pointer");
  SL_REQUIRE(d_ws && ws_bytes >= sl_redundancy_ws_bytes(Bt, C, D), "sl_redundancy: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  unsigned char* ws = (unsigned char*)(((uintptr_t)d_ws + 255) & ~(uintptr_t)255);
  float* rinv = (float*)ws;
  float* rowmax = (float*)(ws + align256((size_t)C * 4));
  float* sims = (float*)(ws + 2 * align256((size_t)C * 4));
  for (int64_t b = 0; b < Bt; ++b) {
    const float* X = d_V + b * C * D;
    void* split = ws + 2 * align256((size_t)C * 4) + align256((size_t)C * (size_t)C * 4);
    if (int rc = cosine_matrix_nt(X, C, X, C, D, rinv, rinv, sims, split, st)) return rc;
    int64_t blocks = (C + 3) / 4;
    const int64_t cap = (int64_t)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(rowmax_offdiag_kernel, dim3((unsigned)blocks), dim3(256), 0, st, sims, C, rowmax);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, st, rowmax, C, d_out + b);
  }
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

SL_API int sl_template_mean(const float* d_E, const float* d_E0, int64_t Q, int64_t T, int64_t D, float* d_out,
                            void* stream) {
  SL_REQUIRE(Q >= 0 && T >= 1 && D >= 0, "sl_template_mean: bad shape");
  if (Q * D == 0) return 0;
  SL_REQUIRE(d_E && d_E0 && d_out, "sl_template_mean: null pointer");
  int64_t blocks = (Q * D + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(template_mean_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_E, d_E0, Q,
                     (int)T, D, d_out);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}
