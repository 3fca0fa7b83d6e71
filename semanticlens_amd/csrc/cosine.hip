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
#include <cstdlib>
#include <cstring>

#include "gemm_bf16x3.hpp"
#include "gemm_f32.hpp"

namespace sl {
namespace {

// The similarity matrices leave with streaming (nt) stores: 369 MB per text_probing call that nothing on the GPU re-reads
// soon; with default stores the dirty lines are written back at the kernel boundary (the call 563 -> 547 us, the GEMM's own
// HIP-event time 495 -> 498 us: tools/probing_ab.py, three alternating runs).  The encoder's epilogues keep default stores
// (their consumer is the next kernel; nt there: 6.76 -> 6.79 ms per encode, tools/encoder_ab.py).
// out[q][c] = acc * rinv_x[q] * rinv_y[c]
struct CosineEpi {
  const float* ra;
  const float* rb;
  float* out;
  int64_t N;
  __device__ inline float column(int64_t col) const { return rb[col]; }
  __device__ inline void store(int64_t row, int64_t col, float acc, float cv) const { __builtin_nontemporal_store(acc * ra[row] * cv, &out[row * N + col]); }
  // the same epilogue for a GEMM over columns [c0, ...) of the output (gemm_bf16x3.hpp: column-strip split)
  CosineEpi shifted(int64_t c0) const { return CosineEpi{ra, rb + c0, out + c0, N}; }
};
// operands were normalised before the bf16 split: the accumulator is the cosine
struct PlainEpi {
  float* out;
  int64_t N;
  __device__ inline float column(int64_t) const { return 0.f; }
  __device__ inline void store(int64_t row, int64_t col, float acc, float) const { __builtin_nontemporal_store(acc, &out[row * N + col]); }
  PlainEpi shifted(int64_t c0) const { return PlainEpi{out + c0, N}; }
};

size_t align256_(size_t n) { return (n + 255) & ~(size_t)255; }

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

// Epilogue of the fused multi-layer probe: the B operand is the row-wise concatenation of L layers' concept
// matrices; column `col` belongs to layer l with start[l] <= col < start[l + 1] and goes to outs[l] (Q x C_l).
// column() runs once per output column and hands the layer index to store() through its float slot.
constexpr int kMaxFusedLayers = 32;
struct MultiEpi {
  float* out[kMaxFusedLayers];
  int64_t start[kMaxFusedLayers + 1];
  int n;
  int64_t c0;  // column-strip split: this launch's column 0 is column c0 of the concatenation
  __device__ float column(int64_t col) const {
    col += c0;
    int l = 0;
    while (l + 1 < n && col >= start[l + 1]) ++l;
    return __int_as_float(l);
  }
  __device__ void store(int64_t row, int64_t col, float acc, float cv) const {
    const int l = __float_as_int(cv);
    __builtin_nontemporal_store(acc, &out[l][row * (start[l + 1] - start[l]) + (col + c0 - start[l])]);
  }
  MultiEpi shifted(int64_t by) const {
    MultiEpi e = *this;
    e.c0 += by;
    return e;
  }
};

// The same routing for the fp32-MFMA mode, whose operands are not pre-normalised: the per-column value carries the layer
// and the column's inverse norm, the epilogue scales like CosineEpi (acc * rinv_x[row] * rinv_y[col], same order: the
// fused probe is bit-identical to layer-by-layer sl_similarity calls).
struct LayerCol {
  int layer;
  float rinv;
};
struct MultiCosineEpi {
  float* out[kMaxFusedLayers];
  int64_t start[kMaxFusedLayers + 1];
  int n;
  const float* ra;
  const float* rb;
  int64_t c0;
  __device__ LayerCol column(int64_t col) const {
    col += c0;
    int l = 0;
    while (l + 1 < n && col >= start[l + 1]) ++l;
    return LayerCol{l, rb[col]};
  }
  __device__ void store(int64_t row, int64_t col, float acc, LayerCol cv) const {
    const int l = cv.layer;
    __builtin_nontemporal_store(acc * ra[row] * cv.rinv, &out[l][row * (start[l + 1] - start[l]) + (col + c0 - start[l])]);
  }
  MultiCosineEpi shifted(int64_t by) const {
    MultiCosineEpi e = *this;
    e.c0 += by;
    return e;
  }
};

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

// SL_GEMM_MODE=f32    : fp32-input MFMA (exact fp32 products)
// SL_GEMM_MODE=bf16x3 : split-bf16 on the bf16 matrix cores (gemm_bf16x3.hpp), |error| ~1e-6 on cosines
static int g_gemm_mode = -1;  // -1: not set (environment decides), 0: f32, 1: bf16x3
bool use_bf16x3() {
  if (g_gemm_mode >= 0) return g_gemm_mode == 1;
  static const bool v = [] {
    const char* e = getenv("SL_GEMM_MODE");
    return !(e && strcmp(e, "f32") == 0);
  }();
  return v;
}
void set_gemm_mode(int m) { g_gemm_mode = m; }

// bytes of split-bf16 scratch for an (R x K) operand: hi + lo
size_t split_bytes(int64_t R, int64_t K) { return align256_(gemm3::split_elems(R, K) * 2); }

// rows of up to kMaxFusedLayers matrices -> their rows of ONE split operand, L2-normalised on the way: the inverse norm
// (the arithmetic of row_inv_norm_kernel: lane-strided squares, xor-shuffle tree) and the scaled hi / lo halves in one
// pass over HBM — the row is read a second time out of L2.  Replaces one row_inv_norm + one split launch per matrix
// (26 launches and 0.23 ms of a 0.73 ms text_probing call at 12 layers) by one launch per side; results are bit-identical.
struct RowSources {
  const float* ptr[kMaxFusedLayers];
  int64_t start[kMaxFusedLayers + 1];
  int n;
};
__global__ __launch_bounds__(256) void norm_split_rows_kernel(RowSources src, int64_t K, float eps, uint16_t* __restrict__ sp) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  const int64_t Kp = gemm3::split_kp(K);
  const int64_t rows = src.start[src.n];
  int l = 0;
  for (int64_t r = wave; r < rows; r += nw) {
    while (l + 1 < src.n && r >= src.start[l + 1]) ++l;  // rows ascend within a wave
    const float* p = src.ptr[l] + (r - src.start[l]) * K;
    if ((K & 3) == 0 && K <= 2048 && (((uintptr_t)p) & 15) == 0) {
      // rows of up to 2048 floats: read ONCE as 16-byte pieces into registers (two rows of the wave in flight would not fit),
      // sum of squares in the scalar kernel's per-lane order (element i belongs to lane i % 64 there: here lane = (i / 4) % 64,
      // so the sum is a different fp32 order — the inverse norm may differ in the last bit from row_inv_norm_kernel's)
      float4 v[8];
      float s4 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t c = ((int64_t)j * 64 + lane) * 4;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < K) v[j] = *reinterpret_cast<const float4*>(p + c);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s4 += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
      for (int off = 32; off > 0; off >>= 1) s4 += __shfl_xor(s4, off, 64);
      const float rinv4 = 1.f / fmaxf(sqrtf(s4), eps);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t c = ((int64_t)j * 64 + lane) * 4;
        if (c < Kp) gemm3::store_split4(make_float4(v[j].x * rinv4, v[j].y * rinv4, v[j].z * rinv4, v[j].w * rinv4), r, c, Kp, sp);
      }
      continue;
    }
    float s = 0.f;
    for (int64_t i = lane; i < K; i += 64) s += p[i] * p[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float rinv = 1.f / fmaxf(sqrtf(s), eps);
    if ((K & 3) == 0 && (((uintptr_t)p) & 15) == 0) {
      for (int64_t c = (int64_t)lane * 4; c < Kp; c += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < K) {
          v = *reinterpret_cast<const float4*>(p + c);
          v.x *= rinv; v.y *= rinv; v.z *= rinv; v.w *= rinv;
        }
        gemm3::store_split4(v, r, c, Kp, sp);
      }
    } else {
      for (int64_t c = lane; c < Kp; c += 64) gemm3::store_split(c < K ? p[c] * rinv : 0.f, r, c, Kp, sp);
    }
  }
}
// fp32-MFMA mode: the same walk, writing the inverse norms and (GATHER) the rows themselves into one fp32 operand
template <bool GATHER>
__global__ __launch_bounds__(256) void norm_rows_kernel(RowSources src, int64_t K, float eps, float* __restrict__ rinv_out,
                                                         float* __restrict__ rows_out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  const int64_t rows = src.start[src.n];
  int l = 0;
  for (int64_t r = wave; r < rows; r += nw) {
    while (l + 1 < src.n && r >= src.start[l + 1]) ++l;
    const float* p = src.ptr[l] + (r - src.start[l]) * K;
    float s = 0.f;
    for (int64_t i = lane; i < K; i += 64) {
      const float v = p[i];
      s += v * v;
      if constexpr (GATHER) rows_out[r * K + i] = v;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) rinv_out[r] = 1.f / fmaxf(sqrtf(s), eps);
  }
}
int launch_norm_gather(const RowSources& src, int64_t K, float eps, float* rinv, float* rows_out, hipStream_t st) {
  const int64_t rows = src.start[src.n];
  if (rows == 0) return 0;
  int64_t blocks = (rows + 3) / 4;
  const int64_t cap = (int64_t)num_cus() * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(norm_rows_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, src, K, eps, rinv, rows_out);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}
int launch_norm_split(const RowSources& src, int64_t K, float eps, uint16_t* sp, hipStream_t st) {
  const int64_t rows = src.start[src.n];
  if (rows == 0) return 0;
  int64_t blocks = (rows + 3) / 4;
  const int64_t cap = (int64_t)num_cus() * 8;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(norm_split_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, K, eps, sp);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

// out[M][N] = cos(A rows, B rows).  ra/rb: scratch for the inverse norms; split: scratch of
// split_bytes(M,K) + split_bytes(N,K) bytes (may be NULL -> fp32 path).  Used by K6 and K8.
int cosine_matrix_nt(const float* A, int64_t M, const float* B, int64_t N, int64_t K, float* ra, float* rb, float* out,
                     void* split, hipStream_t st) {
  const bool same = (B == A && N == M);
  // below K = 64 too few products average the ~2^-16 split error down; those GEMMs are tiny anyway
  if (split && use_bf16x3() && K >= 64) {
    unsigned char* p = (unsigned char*)split;
    uint16_t* as = (uint16_t*)p;
    uint16_t* bs = as;
    RowSources src{};
    src.ptr[0] = A;
    src.start[1] = M;
    src.n = 1;
    if (int rc = launch_norm_split(src, K, 1e-12f, as, st)) return rc;  // x_hat = x * rinv, then the split matrix
    if (!same) {
      bs = (uint16_t*)(p + split_bytes(M, K));
      src.ptr[0] = B;
      src.start[1] = N;
      if (int rc = launch_norm_split(src, K, 1e-12f, bs, st)) return rc;
    }
    ProfScope prof(SL_PROF_GEMM, st, 2.0 * (double)M * (double)N * (double)K);
    return gemm3::launch_gemm3_nt(prof, as, M, bs, N, K, PlainEpi{out, N}, st);
  }
  if (int rc = launch_inv_norm(A, M, K, 1e-12f, ra, st)) return rc;
  if (same) {
    rb = ra;
  } else if (int rc = launch_inv_norm(B, N, K, 1e-12f, rb, st)) {
    return rc;
  }
  ProfScope prof(SL_PROF_GEMM, st, 2.0 * (double)M * (double)N * (double)K);
  return gemm::launch_gemm_nt(prof, A, M, B, N, K, CosineEpi{ra, rb, out, N}, st);
}

size_t cosine_split_bytes(int64_t M, int64_t N, int64_t K) { return split_bytes(M, K) + split_bytes(N, K); }

// One query matrix against L concept matrices (the per-layer loop of lens.py:206-214): the query is
// normalised and split once, each layer then costs its own split + one GEMM.
int cosine_matrix_multi(const float* X, int64_t Q, int64_t K, const float* const* Ys, const int64_t* Cs, int L,
                        float* const* outs, unsigned char* ws, hipStream_t st) {
  int64_t cmax = 0, csum = 0;
  for (int l = 0; l < L; ++l) {
    cmax = Cs[l] > cmax ? Cs[l] : cmax;
    csum += Cs[l];
  }
  const bool fast = use_bf16x3() && K >= 64;
  const bool fused = fast && L <= kMaxFusedLayers;
  // fp32-MFMA mode: the layers are gathered into ONE (sum C, K) fp32 operand when that lets the 256 x 256 8-phase kernel
  // run (rows of whole 128-byte lines, enough tiles to fill the chip): 12 launches of 120 tiles become one of 1440
  const bool fused_f32 = !fast && L > 1 && L <= kMaxFusedLayers && K % 32 == 0 && K > 0 && ((uintptr_t)X & 15) == 0 &&
                         gemm8::fits(Q, csum, K * 4) && gemm8::worth_it(Q, csum) && !gemm8::worth_it(Q, cmax);
  const int64_t yrows = (fused || fused_f32) ? csum : cmax;  // rows of the y scratch (all layers, or one at a time)
  float* rx = (float*)ws;
  float* ry = (float*)(ws + align256_((size_t)Q * 4));
  unsigned char* sp = ws + align256_((size_t)Q * 4) + align256_((size_t)yrows * 4);
  uint16_t* xs = (uint16_t*)sp;
  uint16_t* ys = (uint16_t*)(sp + split_bytes(Q, K));
  if (fused) {
    // every layer is normalised + split into its rows of ONE (sum C, K) operand — one launch for x, one for all layers —
    // and a single GEMM launch then fills the chip (12 x 768 columns: 1440 tiles instead of 12 launches of 120)
    RowSources xsrc{};
    xsrc.ptr[0] = X;
    xsrc.start[0] = 0;
    xsrc.start[1] = Q;
    xsrc.n = 1;
    MultiEpi epi{};
    RowSources ysrc{};
    epi.n = 0;
    int64_t off = 0;
    for (int l = 0; l < L; ++l) {
      if (Cs[l] == 0) continue;
      ysrc.ptr[epi.n] = Ys[l];
      ysrc.start[epi.n] = off;
      epi.out[epi.n] = outs[l];
      epi.start[epi.n] = off;
      ++epi.n;
      off += Cs[l];
    }
    epi.start[epi.n] = off;
    ysrc.start[epi.n] = off;
    ysrc.n = epi.n;
    if (Q * off == 0) return 0;
    // Round 6: when the y operand starts right behind the x operand's last row (no alignment gap) the query and every layer are
    // rows [0, Q) and [Q, Q + sum C) of ONE split buffer: one normalise + split launch instead of two (the same per-row arithmetic)
    const size_t x_bytes = gemm3::split_elems(Q, K) * 2;
    if (epi.n + 1 <= kMaxFusedLayers && split_bytes(Q, K) == x_bytes) {
      RowSources all{};
      all.ptr[0] = X;
      all.start[0] = 0;
      for (int i = 0; i < epi.n; ++i) {
        all.ptr[i + 1] = ysrc.ptr[i];
        all.start[i + 1] = Q + ysrc.start[i];
      }
      all.start[epi.n + 1] = Q + off;
      all.n = epi.n + 1;
      if (int rc = launch_norm_split(all, K, 1e-12f, xs, st)) return rc;
    } else {
      if (int rc = launch_norm_split(xsrc, K, 1e-12f, xs, st)) return rc;
      if (int rc = launch_norm_split(ysrc, K, 1e-12f, ys, st)) return rc;
    }
    ProfScope prof(SL_PROF_GEMM, st, 2.0 * (double)Q * (double)off * (double)K);
    return gemm3::launch_gemm3_nt(prof, xs, Q, ys, off, K, epi, st);
  }
  if (int rc = launch_inv_norm(X, Q, K, 1e-12f, rx, st)) return rc;
  if (fast)
    if (int rc = gemm3::launch_split(X, rx, Q, K, xs, st)) return rc;
  if (fused_f32) {
    MultiCosineEpi epi{};
    epi.n = 0;
    epi.ra = rx;
    epi.rb = ry;
    float* yall = (float*)sp;  // the split scratch is at least (sum C) x K floats
    RowSources ysrc{};
    int64_t off = 0;
    for (int l = 0; l < L; ++l) {
      if (Cs[l] == 0) continue;
      ysrc.ptr[epi.n] = Ys[l];
      ysrc.start[epi.n] = off;
      epi.out[epi.n] = outs[l];
      epi.start[epi.n] = off;
      ++epi.n;
      off += Cs[l];
    }
    ysrc.start[epi.n] = off;
    ysrc.n = epi.n;
    if (int rc = launch_norm_gather(ysrc, K, 1e-12f, ry, yall, st)) return rc;  // one launch: inverse norms + the gathered rows
    epi.start[epi.n] = off;
    ProfScope prof(SL_PROF_GEMM, st, 2.0 * (double)Q * (double)off * (double)K);
    return gemm::launch_gemm_nt(prof, X, Q, yall, off, K, epi, st);
  }
  for (int l = 0; l < L; ++l) {
    if (Q * Cs[l] == 0) continue;
    if (int rc = launch_inv_norm(Ys[l], Cs[l], K, 1e-12f, ry, st)) return rc;
    ProfScope prof(SL_PROF_GEMM, st, 2.0 * (double)Q * (double)Cs[l] * (double)K);
    int rc;
    if (fast) {
      if ((rc = gemm3::launch_split(Ys[l], ry, Cs[l], K, ys, st))) return rc;
      rc = gemm3::launch_gemm3_nt(prof, xs, Q, ys, Cs[l], K, PlainEpi{outs[l], Cs[l]}, st);
    } else {
      rc = gemm::launch_gemm_nt(prof, X, Q, Ys[l], Cs[l], K, CosineEpi{rx, ry, outs[l], Cs[l]}, st);
    }
    if (rc) return rc;
  }
  return 0;
}

}  // namespace sl

using namespace sl;

static size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

SL_API size_t sl_similarity_ws_bytes(int64_t xr, int64_t xc, int64_t yr, int64_t yc) {
  size_t b = align256((size_t)xr * 4) + align256((size_t)yr * 4);
  if (!(xr == yr && xc == yc) && xc == yr) b += align256((size_t)yr * (size_t)yc * 4);
  else if (xc == yc) b += cosine_split_bytes(xr, yr, xc);
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
  void* split = ws + align256((size_t)xr * 4) + align256((size_t)yr * 4);
  if (int rc = cosine_matrix_nt(d_x, xr, d_y, yr, xc, rx, ry, d_out, split, st)) return rc;
  return 2;
}

SL_API size_t sl_similarity_multi_ws_bytes(int64_t Q, int64_t K, const int64_t* h_Cs, int L) {
  int64_t csum = 0;  // the fused path keeps every layer's normalised + split rows at once
  for (int l = 0; l < L; ++l) csum += h_Cs[l] > 0 ? h_Cs[l] : 0;
  return align256((size_t)Q * 4) + align256((size_t)csum * 4) + cosine_split_bytes(Q, csum, K) + 512;
}

SL_API int sl_similarity_multi(const float* d_x, int64_t Q, int64_t K, const float* const* h_d_ys, const int64_t* h_Cs,
                               int L, float* const* h_d_outs, void* d_ws, size_t ws_bytes, void* stream) {
  SL_REQUIRE(Q >= 0 && K >= 0 && L >= 0, "sl_similarity_multi: negative shape");
  if (L == 0 || Q == 0) return 0;
  SL_REQUIRE(d_x && h_d_ys && h_Cs && h_d_outs, "sl_similarity_multi: null pointer");
  for (int l = 0; l < L; ++l) {
    SL_REQUIRE(h_Cs[l] >= 0 && (h_Cs[l] == 0 || (h_d_ys[l] && h_d_outs[l])), "sl_similarity_multi: bad layer %d", l);
    // the shape quirks of scores.py:119-123 must not apply to any layer (the caller falls back to sl_similarity)
    SL_REQUIRE(!(h_Cs[l] == Q) && !(K == h_Cs[l]), "sl_similarity_multi: layer %d hits a shape-quirk branch of similarity_score", l);
  }
  SL_REQUIRE(d_ws && ws_bytes >= sl_similarity_multi_ws_bytes(Q, K, h_Cs, L), "sl_similarity_multi: workspace too small");
  unsigned char* ws = (unsigned char*)(((uintptr_t)d_ws + 255) & ~(uintptr_t)255);
  return cosine_matrix_multi(d_x, Q, K, h_d_ys, h_Cs, L, h_d_outs, ws, (hipStream_t)stream);
}

SL_API int sl_set_gemm_mode(int mode) {
  SL_REQUIRE(mode >= -1 && mode <= 1, "sl_set_gemm_mode: mode %d not in {-1 (environment), 0 (f32), 1 (bf16x3)}", mode);
  set_gemm_mode(mode);
  return 0;
}
