// K6 — similarity_score (semanticlens/scores.py:84-128), the text/image-probing contraction
// of lens.py:206-214:  normalize(x) @ normalize(y)^T.
//
// The only dense contraction on the hot path, and the only MFMA user.  Inputs are fp32 and the
// result must match the reference to 1e-4, so the GEMM runs on the fp32-input matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, 157.3 TFLOP/s peak on MI355X) rather than
// rounding operands to bf16.  L2 normalisation is folded into the epilogue:
//   out[q][c] = (x_q . y_c) * rinv_x[q] * rinv_y[c],   rinv = 1 / max(||row||_2, 1e-12)
// so normalised copies of x and y are never materialised.
//
// Roofline: MFMA (fp32 input).  Algorithmic flops per launch = 2*Q*C*D.
//
// Tiling: workgroup 256 threads = 4 waves (2x2), block tile 128x128x32, wave tile 64x64 =
// 2x2 MFMA tiles of 32x32 (64 accumulator registers).  Both operands are K-contiguous (an "NT"
// GEMM), staged global -> registers -> LDS with the next tile's loads in flight during the
// current tile's MFMAs.  LDS rows are padded to 36 floats so the 16-byte fragment reads of a
// 16-lane group land in 16 distinct 16-byte slots of the 256-byte bank row (conflict-free).
// In one MFMA the two lane halves (lane>>5) consume two different k of the tile; which two is
// free as long as A and B agree, so half h owns k in [16h, 16h+16) and reads them as float4.
#include "common.hpp"

namespace sl {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, LDS_LD = BK + 4;
#ifndef SL_GEMM_XCD_REMAP
#define SL_GEMM_XCD_REMAP 0  // A/B measured: no gain (the GEMM is MFMA-issue bound, not L2-miss bound)
#endif
constexpr bool XCD_REMAP = SL_GEMM_XCD_REMAP != 0;

// one wave per row: 1 / max(||row||, eps)
__global__ __launch_bounds__(256) void row_inv_norm_kernel(const float* __restrict__ x, int64_t rows, int64_t cols,
                                                            float eps, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t r = wave; r < rows; r += nw) {
    const float* p = x + r * cols;
    float s = 0.f;
    for (int64_t i = lane; i < cols; i += 64) s += p[i] * p[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) out[r] = 1.f / fmaxf(sqrtf(s), eps);
  }
}

// scale rows in place-free fashion: out[r][:] = x[r][:] * rinv[r]
__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ x, const float* __restrict__ rinv,
                                                          int64_t rows, int64_t cols, float* __restrict__ out) {
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = x[i] * rinv[i / cols];
}

// branch 0: F.cosine_similarity(x, y, dim=-1) with eps = 1e-8 on each norm (scores.py:127)
__global__ __launch_bounds__(256) void rowwise_cosine_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              int64_t rows, int64_t cols, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t r = wave; r < rows; r += nw) {
    const float* a = x + r * cols;
    const float* b = y + r * cols;
    float d = 0.f, sa = 0.f, sb = 0.f;
    for (int64_t i = lane; i < cols; i += 64) {
      d += a[i] * b[i];
      sa += a[i] * a[i];
      sb += b[i] * b[i];
    }
    for (int off = 32; off > 0; off >>= 1) {
      d += __shfl_xor(d, off, 64);
      sa += __shfl_xor(sa, off, 64);
      sb += __shfl_xor(sb, off, 64);
    }
    if (lane == 0) out[r] = d / (fmaxf(sqrtf(sa), 1e-8f) * fmaxf(sqrtf(sb), 1e-8f));
  }
}

// branch 1 (shape quirk, scores.py:122-123): out = xhat @ yhat with yhat (K,N) row-major, already
// normalised; x is scaled by rinv_x in the epilogue.  Rare; plain VALU kernel.
__global__ __launch_bounds__(256) void gemm_nn_kernel(const float* __restrict__ x, const float* __restrict__ rx,
                                                       const float* __restrict__ yhat, int64_t M, int64_t K, int64_t N,
                                                       float* __restrict__ out) {
  const int64_t total = M * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / N, n = i % N;
    float s = 0.f;
    for (int64_t t = 0; t < K; ++t) s = fmaf(x[m * K + t] * rx[m], yhat[t * N + n], s);
    out[i] = s;
  }
}

// ---- the MFMA kernel: out[M][N] = (A[M][K] . B[N][K]^T) * ra[m] * rb[n] ------------------------
template <bool VEC>
__device__ inline void load_tile_regs(const float* __restrict__ g, int64_t rows, int64_t K, int64_t row0, int64_t k0,
                                      int tid, float4 (&r)[4]) {
  // tile = 128 rows x 32 floats = 1024 float4; thread t takes pieces t, t+256, t+512, t+768
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = tid + i * 256;
    const int row = piece >> 3, c4 = piece & 7;
    const int64_t gr = row0 + row, gk = k0 + c4 * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gr < rows) {
      const float* p = g + gr * K + gk;
      if constexpr (VEC) {
        if (gk + 4 <= K) v = *reinterpret_cast<const float4*>(p);
        // K % 4 == 0 on this path, so a piece is either fully inside or fully outside
      } else {
        if (gk + 0 < K) v.x = p[0];
        if (gk + 1 < K) v.y = p[1];
        if (gk + 2 < K) v.z = p[2];
        if (gk + 3 < K) v.w = p[3];
      }
    }
    r[i] = v;
  }
}

__device__ inline void store_tile_lds(float* __restrict__ s, int tid, const float4 (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = tid + i * 256;
    const int row = piece >> 3, c4 = piece & 7;
    *reinterpret_cast<float4*>(s + row * LDS_LD + c4 * 4) = r[i];
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void cosine_gemm_nt_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                              const float* __restrict__ ra,
                                                              const float* __restrict__ rb, int64_t M, int64_t N,
                                                              int64_t K, float* __restrict__ out, int tiles_n) {
  __shared__ __align__(16) float sA[2][BM * LDS_LD];
  __shared__ __align__(16) float sB[2][BN * LDS_LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;  // wave position in the 2x2 grid
  const int li = lane & 31, lh = lane >> 5;
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch, speed only), and each XCD has
  // its own L2.  Give every XCD a contiguous run of tiles so the tiles_n tiles that share one 128-row
  // panel of A hit it in the same L2 (bijective for any grid size).
  int tile = blockIdx.x;
  if (XCD_REMAP) {
    const int nblk = gridDim.x, xcd = tile & 7, j = tile >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int64_t m0 = (int64_t)(tile / tiles_n) * BM;
  const int64_t n0 = (int64_t)(tile % tiles_n) * BN;

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // Per-thread source pointers of its 4 + 4 pieces of a tile, computed once: rows past the matrix edge are
  // clamped onto the last row (their products land in output rows/cols that are never stored), so a full
  // K tile needs no bounds checks and no exec-masked branches between the MFMA blocks.
  const float* pa[4];
  const float* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = tid + i * 256;
    const int row = piece >> 3, c4 = piece & 7;
    const int64_t ar = m0 + row < M ? m0 + row : M - 1;
    const int64_t br = n0 + row < N ? n0 + row : N - 1;
    pa[i] = A + ar * K + c4 * 4;
    pb[i] = B + br * K + c4 * 4;
  }
  float4 ra_[4], rb_[4];
  // full tile: unconditional 16-byte loads (VEC) — nothing between the MFMA blocks but these 8 loads
  auto load_full = [&](int64_t k0) {
    if constexpr (VEC) {
#pragma unroll
      for (int i = 0; i < 4; ++i) ra_[i] = *reinterpret_cast<const float4*>(pa[i] + k0);
#pragma unroll
      for (int i = 0; i < 4; ++i) rb_[i] = *reinterpret_cast<const float4*>(pb[i] + k0);
    } else {
      load_tile_regs<false>(A, M, K, m0, k0, tid, ra_);
      load_tile_regs<false>(B, N, K, n0, k0, tid, rb_);
    }
  };
  // last, partial tile of K: element-wise and zero filled
  auto load_tail = [&](int64_t k0) {
    load_tile_regs<false>(A, M, K, m0, k0, tid, ra_);
    load_tile_regs<false>(B, N, K, n0, k0, tid, rb_);
  };
  auto stage = [&](int buf) {
    store_tile_lds(sA[buf], tid, ra_);
    store_tile_lds(sB[buf], tid, rb_);
  };
  auto compute = [&](int cur) {
    const float* a_base = sA[cur] + (wm * 64 + li) * LDS_LD + lh * 16;
    const float* b_base = sB[cur] + (wn * 64 + li) * LDS_LD + lh * 16;
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // 4 x float4 = this half's 16 k of the tile
      const float4 a0 = *reinterpret_cast<const float4*>(a_base + u * 4);
      const float4 a1 = *reinterpret_cast<const float4*>(a_base + 32 * LDS_LD + u * 4);
      const float4 b0 = *reinterpret_cast<const float4*>(b_base + u * 4);
      const float4 b1 = *reinterpret_cast<const float4*>(b_base + 32 * LDS_LD + u * 4);
      const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
      const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv0[e], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv1[e], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv0[e], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv1[e], acc[1][1], 0, 0, 0);
      }
    }
  };

  const int nfull = (int)(K / BK);
  const int ntiles = nfull + ((K % BK) ? 1 : 0);
  if (ntiles > 0) {
    if (nfull > 0) load_full(0);
    else load_tail(0);
    stage(0);
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nfull; ++kt) {  // steady state: tile kt+1 (full) flies in while tile kt is multiplied
      load_full((int64_t)(kt + 1) * BK);
      // hipcc otherwise sinks these loads to the end of the MFMA block (shorter live ranges), right in front of
      // the LDS stores that need them, and the whole HBM/L2 latency is exposed once per tile
      __builtin_amdgcn_sched_barrier(0);
      compute(kt & 1);
      __builtin_amdgcn_sched_barrier(0);
      stage((kt + 1) & 1);
      __syncthreads();
    }
    if (kt + 1 < ntiles) {  // the partial K tile follows
      load_tail((int64_t)(kt + 1) * BK);
      compute(kt & 1);
      stage((kt + 1) & 1);
      __syncthreads();
      ++kt;
    }
    compute(kt & 1);
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t col = n0 + wn * 64 + j * 32 + li;
      const float sc = col < N ? rb[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < M && col < N) out[row * N + col] = acc[i][j][r] * ra[row] * sc;
      }
    }
  }
}

int launch_inv_norm(const float* x, int64_t rows, int64_t cols, float eps, float* out, hipStream_t st) {
  int64_t blocks = (rows + 3) / 4;
  const int64_t cap = (int64_t)num_cus() * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(row_inv_norm_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, rows, cols, eps, out);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// out[M][N] = cos(A rows, B rows); ra/rb: scratch for the inverse norms.  Used by K6 and K8.
int cosine_matrix_nt(const float* A, int64_t M, const float* B, int64_t N, int64_t K, float* ra, float* rb, float* out,
                     hipStream_t st) {
  if (int rc = launch_inv_norm(A, M, K, 1e-12f, ra, st)) return rc;
  if (B == A && N == M) {
    rb = ra;
  } else if (int rc = launch_inv_norm(B, N, K, 1e-12f, rb, st)) {
    return rc;
  }
  const int64_t tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
  SL_REQUIRE(tm * tn < (1ll << 31), "cosine GEMM: too many tiles");
  ProfScope prof(SL_PROF_GEMM, st, 2.0 * (double)M * (double)N * (double)K);
  const bool vec = (K % 4 == 0) && (((uintptr_t)A | (uintptr_t)B) & 15) == 0;
  if (vec)
    SL_LAUNCH(prof, cosine_gemm_nt_kernel<true>, dim3((unsigned)(tm * tn)), dim3(256), 0, st, A, B, ra, rb, M, N, K, out,
              (int)tn);
  else
    SL_LAUNCH(prof, cosine_gemm_nt_kernel<false>, dim3((unsigned)(tm * tn)), dim3(256), 0, st, A, B, ra, rb, M, N, K, out,
              (int)tn);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace sl

using namespace sl;

static size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

SL_API size_t sl_similarity_ws_bytes(int64_t xr, int64_t xc, int64_t yr, int64_t yc) {
  size_t b = align256((size_t)xr * 4) + align256((size_t)yr * 4);
  if (!(xr == yr && xc == yc) && xc == yr) b += align256((size_t)yr * (size_t)yc * 4);
  return b + 256;
}

SL_API int sl_similarity(const float* d_x, int64_t xr, int64_t xc, const float* d_y, int64_t yr, int64_t yc,
                         float* d_out, void* d_ws, size_t ws_bytes, void* stream) {
  SL_REQUIRE(xr >= 0 && xc >= 0 && yr >= 0 && yc >= 0, "sl_similarity: negative shape");
  hipStream_t st = (hipStream_t)stream;
  const bool same = (xr == yr && xc == yc);
  // scores.py:119-126 — the branch is a function of the shapes alone
  if (!same && xc != yr && xc != yc) {
    set_error("x and y must have the same shape");  // the reference's ValueError text (scores.py:126)
    return SL_E_INVALID;
  }
  if (xr * xc == 0 && same) return 0;
  SL_REQUIRE(d_x && d_y && d_out, "sl_similarity: null pointer");
  if (same) {
    int64_t blocks = (xr + 3) / 4;
    const int64_t cap = (int64_t)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(rowwise_cosine_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_x, d_y, xr, xc, d_out);
    SL_CHECK_HIP(hipGetLastError());
    return 0;
  }
  SL_REQUIRE(d_ws && ws_bytes >= sl_similarity_ws_bytes(xr, xc, yr, yc), "sl_similarity: workspace too small");
  unsigned char* ws = (unsigned char*)(((uintptr_t)d_ws + 255) & ~(uintptr_t)255);
  float* rx = (float*)ws;
  float* ry = (float*)(ws + align256((size_t)xr * 4));
  if (xc == yr) {  // branch 1: normalize(x) @ normalize(y), no transpose
    float* yhat = (float*)(ws + align256((size_t)xr * 4) + align256((size_t)yr * 4));
    if (xr * yc == 0) return 1;
    if (int rc = launch_inv_norm(d_x, xr, xc, 1e-12f, rx, st)) return rc;
    if (int rc = launch_inv_norm(d_y, yr, yc, 1e-12f, ry, st)) return rc;
    int64_t n = yr * yc, blocks = (n + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_y, ry, yr, yc, yhat);
    n = xr * yc;
    blocks = (n + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(gemm_nn_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_x, rx, yhat, xr, xc, yc, d_out);
    SL_CHECK_HIP(hipGetLastError());
    return 1;
  }
  // branch 2: normalize(x) @ normalize(y)^T — the probing GEMM
  if (xr * yr == 0) return 2;
  if (int rc = cosine_matrix_nt(d_x, xr, d_y, yr, xc, rx, ry, d_out, st)) return rc;
  return 2;
}
