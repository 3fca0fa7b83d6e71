// K7 / K8 / K10 — score reductions of semanticlens/scores.py and the template mean of lens.py.
//
//   clarity_score     (scores.py:18-47)   V (C,n,D)  -> (C)     HBM-bound: C*n*D*4 bytes read once
//   redundancy_score  (scores.py:50-81)   V (Bt,C,D) -> (Bt)    K6's MFMA GEMM + row-max epilogue
//   template mean     (lens.py:196-199)   E (Q*T,D), E0 (T,D) -> (Q,D)
#include "common.hpp"

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

SL_API int sl_clarity(const float* d_V, int64_t C, int64_t n, int64_t D, float* d_out, void* stream) {
  SL_REQUIRE(C >= 0 && n >= 0 && D >= 0, "sl_clarity: negative shape");
  if (C == 0) return 0;
  SL_REQUIRE(d_V && d_out, "sl_clarity: null pointer");
  SL_REQUIRE(n >= 1 && n <= 8192, "sl_clarity: n_samples=%lld not in [1, 8192]", (long long)n);
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof(SL_PROF_SCORES, st, (double)C * n * D * 4);
  int64_t blocks = C;
  const int64_t cap = (int64_t)num_cus() * 8;
  if (blocks > cap) blocks = cap;
  SL_LAUNCH(prof, clarity_kernel, dim3((unsigned)blocks), dim3(256), (size_t)(n + 4) * 4, st, d_V, C, (int)n, D, d_out);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
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
