// fp32-input MFMA "NT" GEMM core shared by K6 (cosine similarity) and the native encoder's linear layers:
//   acc[m][n] = sum_k A[m][k] * B[n][k]      (both operands K-contiguous), then  epi(row, col, acc).
//
// v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, 157.3 TFLOP/s peak on MI355X).  Workgroup 256 threads = 4 waves
// (2x2), block tile 128x128x32, wave tile 64x64 = 2x2 MFMA tiles of 32x32 (64 accumulator registers).  Tiles are
// staged global -> registers -> LDS, double buffered; the next tile's 8 loads are issued (and pinned with
// sched_barrier) at the top of the tile, the 64 MFMAs follow, the 8 LDS stores close it.  LDS rows are padded to
// 36 floats so the 16-byte fragment reads of a 16-lane group hit 16 distinct 16-byte slots (0 bank conflicts, PMC).
// In one MFMA the two lane halves consume two different k; half h owns k in [16h, 16h+16) of the tile.
// The K tail is peeled out of the steady-state loop; rows past the matrix edge are clamped onto the last row
// (their products land in outputs the epilogue never stores).
#pragma once
#include <cstdlib>

#include "common.hpp"
#include "gemm_8phase.hpp"

namespace sl {
namespace gemm {

typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BN = 128, BK = 32, LDS_LD = BK + 4;

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

// Epi: struct with  T column(int64_t col) const  (per-column value — a float or a small default-constructible struct —
//                   loaded once per 16 rows) and  void store(int64_t row, int64_t col, float acc, T colval) const
template <bool VEC, class Epi>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const float* __restrict__ A, const float* __restrict__ B, int64_t M,
                                                       int64_t N, int64_t K, int tiles_n, Epi epi) {
  __shared__ __align__(16) float sA[2][BM * LDS_LD];
  __shared__ __align__(16) float sB[2][BN * LDS_LD];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;  // wave position in the 2x2 grid
  const int li = lane & 31, lh = lane >> 5;
  // (an XCD-aware tile order was A/B tested and is neutral: the kernel is MFMA-issue bound, not L2-miss bound)
  const int tile = blockIdx.x;
#ifdef SL_GEMM_CLOCKPROBE  // tools/native/clock_probe.hip
  const unsigned long long probe_c0 = __builtin_amdgcn_s_memtime(), probe_r0 = __builtin_amdgcn_s_memrealtime();
#endif
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
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      store_mfma_tile<true>(epi, m0 + wm * 64 + i * 32 + 4 * lh, n0 + wn * 64 + j * 32 + li, acc[i][j], M, N);
#ifdef SL_GEMM_CLOCKPROBE
  if (tid == 0) {
    epi.probe[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - probe_c0;
    epi.probe[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - probe_r0;
  }
#endif
}

template <class Epi>
int launch_gemm_nt(ProfScope& prof, const float* A, int64_t M, const float* B, int64_t N, int64_t K, const Epi& epi,
                   hipStream_t st) {
  const int64_t tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
  SL_REQUIRE(tm * tn < (1ll << 31), "GEMM: too many tiles");
  if (tm * tn == 0) return 0;
  const bool vec = (K % 4 == 0) && (((uintptr_t)A | (uintptr_t)B) & 15) == 0;
  // Large grids whose rows are whole 128-byte lines: the 256 x 256 8-phase kernel (gemm_8phase.hpp), same bits.
  // option "f32_tile" = 128 / 8 forces one of the two (tests).
  const int forced = (int)option(OPT_F32_TILE);
  if (vec && K % 32 == 0 && K > 0 && gemm8::fits(M, N, K * 4) && (forced ? forced == 8 : gemm8::worth_it_f32(M, N)))
    return gemm8::launch<gemm8::MODE_F32>(prof, A, M, B, N, K * 4, K / 32, epi, st);
  if (vec)
    SL_LAUNCH(prof, (gemm_nt_kernel<true, Epi>), dim3((unsigned)(tm * tn)), dim3(256), 0, st, A, B, M, N, K, (int)tn, epi);
  else
    SL_LAUNCH(prof, (gemm_nt_kernel<false, Epi>), dim3((unsigned)(tm * tn)), dim3(256), 0, st, A, B, M, N, K, (int)tn, epi);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace gemm
}  // namespace sl
