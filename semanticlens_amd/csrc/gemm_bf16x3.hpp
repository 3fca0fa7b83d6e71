// fp32-accurate "NT" GEMM on the bf16 matrix cores (split-bf16, three products):
//   a = a_hi + a_lo,  b = b_hi + b_lo  with  x_hi = bf16(x),  x_lo = bf16(x - x_hi)
//   a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi        (dropped a_lo*b_lo and the residual of x_lo: ~2^-16 relative)
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Per 32x32x16 block that is 3 MFMAs of 32 cycles against 8
// fp32-input MFMAs of 64 cycles: 5.3x the fp32-MFMA rate at peak (2.5 PFLOP/s bf16 dense / 3 = 833 TFLOP/s of
// fp32-equivalent work; always quoted as ALGORITHMIC flops 2*M*N*K, i.e. one third of the MFMA flops issued).
//
// Operands arrive pre-split (split_bf16_kernel: one pass that also applies the row scale, e.g. the L2
// normalisation of K6), so this kernel only moves bf16: tile 128x128x32, 4 waves (2x2), wave tile 64x64 = 2x2
// MFMA tiles; four LDS images per stage (A_hi, A_lo, B_hi, B_lo), rows of 32 bf16 padded to 80 bytes so the
// 16-byte fragment reads of a 16-lane group land in 16 distinct 16-byte slots; single LDS stage with the next
// tile prefetched into registers (2 pieces per image per thread), which keeps LDS at 40 KB and lets 3 workgroups
// share a CU.  A and B fragments are read with the same (lane>>5)*8 + j k-pattern, so the result does not depend
// on how the hardware orders k inside a fragment.
#pragma once
#include "common.hpp"
#include <cstdlib>

namespace sl {
namespace gemm3 {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, BK = 32;
// tools/native/clock_probe.hip -DSL_G3_LAYOUT_EXPERIMENT: timing-only emulation of an operand layout in which the hi and
// lo halves of a 32-wide k-tile share one 128-byte line (row stride 2 K, k-tile stride 64 elements, lo = hi + 32)
#ifdef SL_G3_LAYOUT_EXPERIMENT
#define SL_G3_LD(K) (2 * (K))
constexpr int KSTEP = 64;
#else
#define SL_G3_LD(K) (K)
constexpr int KSTEP = BK;
#endif
constexpr int ROW_BYTES = 80;                 // 64 data + 16 pad
constexpr int IMG_BYTES = BM * ROW_BYTES;     // one 128-row image

// fp32 (R x K) -> hi, lo bf16 (R x K each); optional per-row scale applied first (x * scale[r])
static __global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          int64_t R, int64_t K, uint16_t* __restrict__ hi,
                                                          uint16_t* __restrict__ lo) {
  const int64_t n = R * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = x[i];
    if (scale) v *= scale[i / K];
    const uint16_t h = f32_to_bf16_rne(v);
    hi[i] = h;
    lo[i] = f32_to_bf16_rne(v - bf16_to_f32(h));
  }
}

// Epi: same contract as gemm_f32.hpp (column(col), store(row, col, acc, colval))
#ifndef SL_G3_WAVES
#define SL_G3_WAVES 2
#endif
template <class Epi>
__global__ __launch_bounds__(256, SL_G3_WAVES) void gemm3_nt_kernel(const uint16_t* __restrict__ Ah, const uint16_t* __restrict__ Al,
                                                        const uint16_t* __restrict__ Bh, const uint16_t* __restrict__ Bl,
                                                        int64_t M, int64_t N, int64_t K, int tiles_n, Epi epi) {
  __shared__ __align__(16) unsigned char smem[4 * IMG_BYTES];  // A_hi | A_lo | B_hi | B_lo
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int tile = blockIdx.x;
  const int64_t m0 = (int64_t)(tile / tiles_n) * BM;
  const int64_t n0 = (int64_t)(tile % tiles_n) * BN;

#ifdef SL_GEMM_CLOCKPROBE  // tools/native/clock_probe.hip: shader cycles vs the 100 MHz constant clock, per workgroup
  const unsigned long long probe_c0 = __builtin_amdgcn_s_memtime(), probe_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // staging: an image is 128 rows x 64 bytes = 512 pieces of 16 bytes; thread t owns pieces t and t + 256
  // (row = piece / 4, quarter = piece % 4).  Rows past the edge are clamped (never stored); K % 8 == 0.
  const uint16_t* src[4][2];
  int lds_off[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int piece = tid + p * 256;
    const int row = piece >> 2, qt = piece & 3;
    const int64_t ar = m0 + row < M ? m0 + row : M - 1;
    const int64_t br = n0 + row < N ? n0 + row : N - 1;
    src[0][p] = Ah + ar * SL_G3_LD(K) + qt * 8;
    src[1][p] = Al + ar * SL_G3_LD(K) + qt * 8;
    src[2][p] = Bh + br * SL_G3_LD(K) + qt * 8;
    src[3][p] = Bl + br * SL_G3_LD(K) + qt * 8;
    lds_off[p] = row * ROW_BYTES + qt * 16;
  }
  uint4 stg[4][2];
  auto load_tile = [&](int64_t k0) {
    // the last K tile may be partial: K % 8 == 0, so a 16-byte piece is inside or outside as a whole
#pragma unroll
    for (int im = 0; im < 4; ++im)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int qt = (tid + p * 256) & 3;
        if (k0 + qt * 8 < K) stg[im][p] = *reinterpret_cast<const uint4*>(src[im][p] + k0);
        else stg[im][p] = make_uint4(0, 0, 0, 0);
      }
  };
  auto load_tile_full = [&](int64_t k0) {
#pragma unroll
    for (int im = 0; im < 4; ++im)
#pragma unroll
      for (int p = 0; p < 2; ++p) stg[im][p] = *reinterpret_cast<const uint4*>(src[im][p] + k0);
  };
  auto stage = [&]() {
#pragma unroll
    for (int im = 0; im < 4; ++im)
#pragma unroll
      for (int p = 0; p < 2; ++p) *reinterpret_cast<uint4*>(smem + im * IMG_BYTES + lds_off[p]) = stg[im][p];
  };
  auto compute = [&]() {
    const unsigned char* a_base = smem + (wm * 64 + li) * ROW_BYTES + lh * 16;
    const unsigned char* b_base = smem + 2 * IMG_BYTES + (wn * 64 + li) * ROW_BYTES + lh * 16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {  // two k-steps of 16 per tile
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        ah[t] = *reinterpret_cast<const bf16x8*>(a_base + t * 32 * ROW_BYTES + ks * 32);
        al[t] = *reinterpret_cast<const bf16x8*>(a_base + IMG_BYTES + t * 32 * ROW_BYTES + ks * 32);
        bh[t] = *reinterpret_cast<const bf16x8*>(b_base + t * 32 * ROW_BYTES + ks * 32);
        bl[t] = *reinterpret_cast<const bf16x8*>(b_base + IMG_BYTES + t * 32 * ROW_BYTES + ks * 32);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // small terms first, the dominant hi*hi last
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  };

  const int nfull = (int)(K / BK);
  const int ntiles = nfull + ((K % BK) ? 1 : 0);
  if (ntiles > 0) {
    if (nfull > 0) load_tile_full(0);
    else load_tile(0);
    stage();
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nfull; ++kt) {
      load_tile_full((int64_t)(kt + 1) * KSTEP);  // flies during the MFMAs
      __builtin_amdgcn_sched_barrier(0);
      compute();
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();  // everyone done reading this stage
      stage();
      __syncthreads();
    }
    if (kt + 1 < ntiles) {
      load_tile((int64_t)(kt + 1) * BK);
      compute();
      __syncthreads();
      stage();
      __syncthreads();
      ++kt;
    }
    compute();
  }

#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t col = n0 + wn * 64 + j * 32 + li;
      const float cv = col < N ? epi.column(col) : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < M && col < N) epi.store(row, col, acc[i][j][r], cv);
      }
    }
  }
#ifdef SL_GEMM_CLOCKPROBE
  if (tid == 0) {
    epi.probe[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - probe_c0;
    epi.probe[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - probe_r0;
  }
#endif
}

// ---- variant 2: 256 x 128 block tile, wave tile 128 x 64, tiles staged by LDS-DMA, single buffer -------------------
// The register-staged kernel above spends as many LDS cycles on its ds_write_b128 staging stores as on fragment
// reads (PMC: LDS pipe ~87 % of the MFMA time) and 32 VGPRs on the in-flight tile.  Here each wave issues twelve
// 1-KiB LDS-DMA loads (global_load_lds_dwordx4) per tile: no VGPR staging, no ds_write.  LDS-DMA writes lane L's 16
// bytes at base + 16 L, so an image is unpadded rows of 64 bytes; fragment reads stay conflict-free through an XOR
// swizzle of the 16-byte slot, slot = chunk ^ ((row >> 2) & 3), applied on the SOURCE address of the DMA and on the
// ds_read address (for every ds_read_b128 lane group the 16 (row % 4, slot) pairs are distinct).  The wave tile is
// 128 x 64: twice the MFMA work per barrier pair and per fragment read (12 ds_read_b128 feed 24 MFMAs per k-step
// instead of 8 feeding 12); two workgroups per CU alternate between their DMA/wait phase and their 48-MFMA phase.
// Accumulation order per output element is the same as in the kernel above: results are bit-identical
// (tools/g3test.py).  Measured at 10000 x 9216 x 1152: 250 -> 272 TFLOP/s algorithmic including the normalise/split
// passes; a 128 x 128 DMA variant with double buffering and one barrier per k-step was no faster than the
// register-staged kernel (246 vs 239) and is not kept; nor are two prefetching forms of this kernel, both slower than
// letting two co-resident workgroups cover each other: double-buffered 32-wide stages (96 KB, one workgroup per CU:
// 245 vs 270) and double-buffered 16-wide stages (48 KB, two per CU, twice the barriers: 230).  A 512-thread
// "ping-pong" kernel (256 x 256 tile, two wave groups alternating between a fragment-load phase and an MFMA phase,
// bit-identical results) was also built and measured: 284 TFLOP/s with 32-wide stages, 237 with 16-wide stages and
// counted vmcnt waits; with its LDS-DMA switched off the same schedule runs at 907 cycles per 768-cycle MFMA phase
// (583 TFLOP/s-equivalent), with it 1697 — the global -> LDS feed, not the MFMA / LDS schedule, is what holds all of
// these kernels near 50 % matrix-pipe duty.  Storing hi and lo of a k-tile in one 128-byte line (whole-line DMA,
// tools/native/fullline_probe.hip) recovers 9 % of the cycles.  Used for grids of >= 8 tiles per CU; smaller problems
// fill the chip better with 128 x 128 tiles.
__device__ __attribute__((aligned(16))) const uint32_t g_zero16[4] = {0, 0, 0, 0};

constexpr int BM3 = 256;
constexpr int IMG3A_BYTES = BM3 * 64, IMG3B_BYTES = BN * 64;
constexpr int BUF3_BYTES = 2 * IMG3A_BYTES + 2 * IMG3B_BYTES;  // A_hi | A_lo | B_hi | B_lo = 48 KB

template <class Epi>
__global__ __launch_bounds__(256, 2) void gemm3_nt_dma256_kernel(const uint16_t* __restrict__ Ah, const uint16_t* __restrict__ Al,
                                                                 const uint16_t* __restrict__ Bh, const uint16_t* __restrict__ Bl,
                                                                 int64_t M, int64_t N, int64_t K, int tiles_n, Epi epi) {
  __shared__ __align__(1024) unsigned char smem[BUF3_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int tile = blockIdx.x;
  const int64_t m0 = (int64_t)(tile / tiles_n) * BM3;
  const int64_t n0 = (int64_t)(tile % tiles_n) * BN;
#ifdef SL_GEMM_CLOCKPROBE  // tools/native/clock_probe.hip
  const unsigned long long probe_c0 = __builtin_amdgcn_s_memtime(), probe_r0 = __builtin_amdgcn_s_memrealtime();
#endif

  floatx16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // DMA: wave w stages A rows [64 w, 64 w + 64) (4 loads per image) and B rows [32 w, 32 w + 32) (2 loads per image)
  const int lrow = lane >> 2;
  int64_t a_src[4], b_src[2];  // element offsets of this lane's chunk inside the hi/lo matrices
  int a_chunk[4], b_chunk[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = w * 64 + i * 16 + lrow;
    a_chunk[i] = (lane & 3) ^ ((row >> 2) & 3);
    const int64_t ar = m0 + row < M ? m0 + row : M - 1;
    a_src[i] = ar * SL_G3_LD(K) + a_chunk[i] * 8;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = w * 32 + i * 16 + lrow;
    b_chunk[i] = (lane & 3) ^ ((row >> 2) & 3);
    const int64_t br = n0 + row < N ? n0 + row : N - 1;
    b_src[i] = br * SL_G3_LD(K) + b_chunk[i] * 8;
  }
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  auto dma_tile = [&](int64_t k0, bool partial) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool zero = partial && k0 + a_chunk[i] * 8 >= K;
      const void* gh = zero ? (const void*)g_zero16 : (const void*)(Ah + a_src[i] + k0);
      const void* gl = zero ? (const void*)g_zero16 : (const void*)(Al + a_src[i] + k0);
      unsigned char* l = smem + (w * 64 + i * 16) * 64;
      __builtin_amdgcn_global_load_lds((glb_void*)gh, (lds_void*)l, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void*)gl, (lds_void*)(l + IMG3A_BYTES), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool zero = partial && k0 + b_chunk[i] * 8 >= K;
      const void* gh = zero ? (const void*)g_zero16 : (const void*)(Bh + b_src[i] + k0);
      const void* gl = zero ? (const void*)g_zero16 : (const void*)(Bl + b_src[i] + k0);
      unsigned char* l = smem + 2 * IMG3A_BYTES + (w * 32 + i * 16) * 64;
      __builtin_amdgcn_global_load_lds((glb_void*)gh, (lds_void*)l, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void*)gl, (lds_void*)(l + IMG3B_BYTES), 16, 0, 0);
    }
  };
  int a_off[4], a_sw[4], b_off[2], b_sw[2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ar = wm * 128 + t * 32 + li;
    a_off[t] = ar * 64;
    a_sw[t] = (ar >> 2) & 3;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int br = wn * 64 + t * 32 + li;
    b_off[t] = 2 * IMG3A_BYTES + br * 64;
    b_sw[t] = (br >> 2) & 3;
  }
  auto compute = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ks * 2 + lh;
      bf16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int bo = b_off[t] + ((c ^ b_sw[t]) << 4);
        bh[t] = *reinterpret_cast<const bf16x8*>(smem + bo);
        bl[t] = *reinterpret_cast<const bf16x8*>(smem + IMG3B_BYTES + bo);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ao = a_off[t] + ((c ^ a_sw[t]) << 4);
        ah[t] = *reinterpret_cast<const bf16x8*>(smem + ao);
        al[t] = *reinterpret_cast<const bf16x8*>(smem + IMG3A_BYTES + ao);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  };

  const int ntiles = (int)((K + BK - 1) / BK);
  const bool tail = (K % BK) != 0;
  for (int kt = 0; kt < ntiles; ++kt) {
    dma_tile((int64_t)kt * KSTEP, tail && kt + 1 == ntiles);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // the tile has landed for every wave
    compute();
    __syncthreads();  // every wave is done reading before the next DMA overwrites the buffer
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t col = n0 + wn * 64 + j * 32 + li;
      const float cv = col < N ? epi.column(col) : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < M && col < N) epi.store(row, col, acc[i][j][r], cv);
      }
    }
  }
#ifdef SL_GEMM_CLOCKPROBE
  if (tid == 0) {
    epi.probe[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - probe_c0;
    epi.probe[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - probe_r0;
  }
#endif
}

inline int launch_split(const float* x, const float* scale, int64_t R, int64_t K, uint16_t* hi, uint16_t* lo, hipStream_t st) {
  int64_t blocks = (R * K + 255) / 256;
  const int64_t cap = (int64_t)num_cus() * 16;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, scale, R, K, hi, lo);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

template <class Epi>
int launch_gemm3_nt(ProfScope& prof, const uint16_t* Ah, const uint16_t* Al, int64_t M, const uint16_t* Bh,
                    const uint16_t* Bl, int64_t N, int64_t K, const Epi& epi, hipStream_t st) {
  const int64_t tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
  SL_REQUIRE(tm * tn < (1ll << 31), "GEMM: too many tiles");
  SL_REQUIRE(K % 8 == 0, "bf16x3 GEMM: K must be a multiple of 8");
  if (tm * tn == 0) return 0;
  static const int forced = [] {
    const char* e = getenv("SL_G3_TILE");  // 128: register-staged 128 x 128 tiles, 256: LDS-DMA staged 256 x 128 tiles
    return e ? atoi(e) : 0;
  }();
  const int64_t tm3 = (M + BM3 - 1) / BM3;
  const bool big = forced ? forced == 256 : tm3 * tn >= (int64_t)8 * num_cus();
  if (big)
    SL_LAUNCH(prof, (gemm3_nt_dma256_kernel<Epi>), dim3((unsigned)(tm3 * tn)), dim3(256), 0, st, Ah, Al, Bh, Bl, M, N, K, (int)tn, epi);
  else
    SL_LAUNCH(prof, (gemm3_nt_kernel<Epi>), dim3((unsigned)(tm * tn)), dim3(256), 0, st, Ah, Al, Bh, Bl, M, N, K, (int)tn, epi);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace gemm3
}  // namespace sl
