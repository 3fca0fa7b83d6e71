// fp32-accurate "NT" GEMM on the bf16 matrix cores (split-bf16, three products):
//   a = a_hi + a_lo,  b = b_hi + b_lo  with  x_hi = bf16(x),  x_lo = bf16(x - x_hi)
//   a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi        (dropped a_lo*b_lo and the residual of x_lo: ~2^-16 relative)
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Per 32x32x16 block that is 3 MFMAs of 32 cycles against 8
// fp32-input MFMAs of 64 cycles: 5.3x the fp32-MFMA rate at peak (2.5 PFLOP/s bf16 dense / 3 = 833 TFLOP/s of
// fp32-equivalent work; always quoted as ALGORITHMIC flops 2*M*N*K, i.e. one third of the MFMA flops issued).
//
// Operand format ("split matrix", produced by split_bf16_kernel and by the split-output epilogues of the encoder
// kernels): an (R x K) fp32 matrix becomes R rows of 2 Kp bf16 values, Kp = K rounded up to 32, zero padded; each
// 32-wide k-tile of a row is one 128-byte line [hi(32) | lo(32)].  A k-tile of a row is therefore fetched as a
// whole cache line (with separate hi / lo matrices every line was requested twice, by consecutive k-tiles: measured
// -9 % cycles for the LDS-DMA kernel), and no kernel needs a partial-tile path.
//
// Kernels, all with the same accumulation order per output element (bit-identical results, tests/test_gpu_parity.py);
// the default for grids of at least half a 256 x 256 tile per CU is the 8-phase kernel of gemm_8phase.hpp:
//   gemm3_nt_kernel         128 x 128 x 32 tile, 4 waves (2 x 2), wave tile 64 x 64; tiles staged through registers
//                           into four LDS images (A_hi, A_lo, B_hi, B_lo) with 80-byte rows (conflict-free fragment
//                           reads); single LDS stage + register prefetch, 40 KB, three workgroups per CU
//   gemm3_nt_dma256_kernel  256 x 128 x 32 tile, wave tile 128 x 64, tiles staged by LDS-DMA (see below); used for
//                           grids of >= 8 tiles per CU
// A and B fragments are read with the same (lane>>5)*8 + j k-pattern, so the result does not depend on how the
// hardware orders k inside a fragment.
#pragma once
#include "common.hpp"
#include "gemm_8phase.hpp"
#include "gemm_w4.hpp"
#include "gemm_skinny.hpp"
#include <cstdlib>

namespace sl {
namespace gemm3 {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int ROW_BYTES = 80;                 // 64 data + 16 pad
constexpr int IMG_BYTES = BM * ROW_BYTES;     // one 128-row image

// ---- split-matrix layout ---------------------------------------------------------------------------------------
__host__ __device__ inline int64_t split_kp(int64_t K) { return (K + 31) & ~(int64_t)31; }
__host__ __device__ inline size_t split_elems(int64_t R, int64_t K) { return (size_t)R * 2 * (size_t)split_kp(K); }
// position of the hi half of element (row, col); its lo half sits 32 elements further
__host__ __device__ inline int64_t split_pos(int64_t row, int64_t col, int64_t Kp) {
  return row * 2 * Kp + (col >> 5) * 64 + (col & 31);
}
// hi / lo halves of a split operand through gfx950's v_cvt_pk_bf16_f32 (round to nearest even, one instruction for two
// values; the bit-twiddled f32_to_bf16_rne of common.hpp costs six each and stays where NaN bit patterns are part of the
// contract: the top-k candidates).  Finite values convert identically.
__device__ inline uint16_t bf16_bits_hw(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }
__device__ inline void store_split(float v, int64_t row, int64_t col, int64_t Kp, uint16_t* __restrict__ sp) {
  const __bf16 h = (__bf16)v;
  const int64_t p = split_pos(row, col, Kp);
  sp[p] = __builtin_bit_cast(uint16_t, h);
  sp[p + 32] = bf16_bits_hw(v - (float)h);
}
// four consecutive columns (col % 4 == 0): two 8-byte stores
__device__ inline void store_split4(const float4 y, int64_t row, int64_t col, int64_t Kp, uint16_t* __restrict__ sp) {
  const __bf16 b0 = (__bf16)y.x, b1 = (__bf16)y.y, b2 = (__bf16)y.z, b3 = (__bf16)y.w;
  const uint16_t h0 = __builtin_bit_cast(uint16_t, b0), h1 = __builtin_bit_cast(uint16_t, b1);
  const uint16_t h2 = __builtin_bit_cast(uint16_t, b2), h3 = __builtin_bit_cast(uint16_t, b3);
  const uint16_t l0 = bf16_bits_hw(y.x - (float)b0), l1 = bf16_bits_hw(y.y - (float)b1);
  const uint16_t l2 = bf16_bits_hw(y.z - (float)b2), l3 = bf16_bits_hw(y.w - (float)b3);
  const int64_t p = split_pos(row, col, Kp);
  *reinterpret_cast<uint2*>(sp + p) = make_uint2((uint32_t)h0 | ((uint32_t)h1 << 16), (uint32_t)h2 | ((uint32_t)h3 << 16));
  *reinterpret_cast<uint2*>(sp + p + 32) = make_uint2((uint32_t)l0 | ((uint32_t)l1 << 16), (uint32_t)l2 | ((uint32_t)l3 << 16));
}

// fp32 (R x K, row stride K) -> split matrix; optional per-row scale applied first (x * scale[r]); zero padding
static __global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                int64_t R, int64_t K, uint16_t* __restrict__ sp) {
  const int64_t Kp = split_kp(K);
  const int64_t n = R * Kp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / Kp, c = i % Kp;
    float v = 0.f;
    if (c < K) {
      v = x[r * K + c];
      if (scale) v *= scale[r];
    }
    store_split(v, r, c, Kp, sp);
  }
}

// ---- kernel 1: 128 x 128 tiles staged through registers ----------------------------------------------------------
// Epi: same contract as gemm_f32.hpp (column(col), store(row, col, acc, colval))
#ifndef SL_G3_WAVES
#define SL_G3_WAVES 2
#endif
template <class Epi>
__global__ __launch_bounds__(256, SL_G3_WAVES) void gemm3_nt_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                                                    int64_t M, int64_t N, int64_t Kp, int tiles_n, Epi epi) {
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
  // (row = piece / 4, quarter = piece % 4) of every image.  Rows past the edge are clamped (never stored).
  const uint16_t* src[4][2];  // [A_hi | A_lo | B_hi | B_lo][piece] in the first k-tile (the lo half of a line is 32 elements in)
  int lds_off[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int piece = tid + p * 256;
    const int row = piece >> 2, qt = piece & 3;
    const int64_t ar = m0 + row < M ? m0 + row : M - 1;
    const int64_t br = n0 + row < N ? n0 + row : N - 1;
    src[0][p] = A + ar * 2 * Kp + qt * 8;
    src[1][p] = src[0][p] + 32;
    src[2][p] = B + br * 2 * Kp + qt * 8;
    src[3][p] = src[2][p] + 32;
    lds_off[p] = row * ROW_BYTES + qt * 16;
  }
  // the in-flight tile: eight named 16-byte registers (as an array hipcc leaves it in scratch memory in this loop shape)
  uint4 s00, s01, s10, s11, s20, s21, s30, s31;
#define SL_G3_LOAD_TILE(kt)                                                  \
  do {                                                                       \
    s00 = *reinterpret_cast<const uint4*>(src[0][0] + (int64_t)(kt) * 64);   \
    s01 = *reinterpret_cast<const uint4*>(src[0][1] + (int64_t)(kt) * 64);   \
    s10 = *reinterpret_cast<const uint4*>(src[1][0] + (int64_t)(kt) * 64);   \
    s11 = *reinterpret_cast<const uint4*>(src[1][1] + (int64_t)(kt) * 64);   \
    s20 = *reinterpret_cast<const uint4*>(src[2][0] + (int64_t)(kt) * 64);   \
    s21 = *reinterpret_cast<const uint4*>(src[2][1] + (int64_t)(kt) * 64);   \
    s30 = *reinterpret_cast<const uint4*>(src[3][0] + (int64_t)(kt) * 64);   \
    s31 = *reinterpret_cast<const uint4*>(src[3][1] + (int64_t)(kt) * 64);   \
  } while (0)
#define SL_G3_STAGE()                                                          \
  do {                                                                         \
    *reinterpret_cast<uint4*>(smem + 0 * IMG_BYTES + lds_off[0]) = s00;        \
    *reinterpret_cast<uint4*>(smem + 0 * IMG_BYTES + lds_off[1]) = s01;        \
    *reinterpret_cast<uint4*>(smem + 1 * IMG_BYTES + lds_off[0]) = s10;        \
    *reinterpret_cast<uint4*>(smem + 1 * IMG_BYTES + lds_off[1]) = s11;        \
    *reinterpret_cast<uint4*>(smem + 2 * IMG_BYTES + lds_off[0]) = s20;        \
    *reinterpret_cast<uint4*>(smem + 2 * IMG_BYTES + lds_off[1]) = s21;        \
    *reinterpret_cast<uint4*>(smem + 3 * IMG_BYTES + lds_off[0]) = s30;        \
    *reinterpret_cast<uint4*>(smem + 3 * IMG_BYTES + lds_off[1]) = s31;        \
  } while (0)
  auto compute = [&]() __attribute__((always_inline)) {
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

  const int ntiles = (int)(Kp / BK);
  if (ntiles > 0) {
    SL_G3_LOAD_TILE(0);
    SL_G3_STAGE();
    __syncthreads();
    for (int kt = 1; kt < ntiles; ++kt) {
      SL_G3_LOAD_TILE(kt);  // flies during the MFMAs
      __builtin_amdgcn_sched_barrier(0);
      compute();
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();  // everyone done reading this stage
      SL_G3_STAGE();
      __syncthreads();
    }
    compute();
  }
#undef SL_G3_LOAD_TILE
#undef SL_G3_STAGE

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

// ---- kernel 2: 256 x 128 block tile, wave tile 128 x 64, tiles staged by LDS-DMA, single buffer ---------------------
// The register-staged kernel above spends as many LDS cycles on its ds_write_b128 staging stores as on fragment
// reads (PMC: LDS pipe ~87 % of the MFMA time) and 32 VGPRs on the in-flight tile.  Here each wave issues twelve
// 1-KiB LDS-DMA loads (global_load_lds_dwordx4) per tile: no VGPR staging, no ds_write.  LDS-DMA writes lane L's 16
// bytes at base + 16 L; one instruction moves 8 rows x 128 bytes, i.e. eight whole [hi | lo] lines, so an LDS image is
// unpadded rows of 128 bytes.  Fragment reads stay conflict-free through an XOR swizzle of the 16-byte slot,
// slot = chunk ^ ((row >> 1) & 7) (chunks 0-3 = hi, 4-7 = lo), applied on the SOURCE address of the DMA and on the
// ds_read address: a 256-byte bank window holds two rows, and for every ds_read_b128 lane group the 8 even and the 8
// odd rows each get 8 distinct slots.  The wave tile is 128 x 64: twice the MFMA work per barrier pair and per fragment
// read (12 ds_read_b128 feed 24 MFMAs per k-step instead of 8 feeding 12); two workgroups per CU alternate between
// their DMA/wait phase and their 48-MFMA phase.
// Measured alternatives that were dropped (all bit-identical): 128 x 128 DMA tiles with double buffering (no faster than
// kernel 1), double-buffered 32- and 16-wide stages of this kernel (slower than two co-resident workgroups covering each
// other), a 512-thread ping-pong kernel (two wave groups alternating fragment-load and MFMA phases, raw s_barrier and
// counted vmcnt): with its LDS-DMA switched off the schedule runs at 907 cycles per 768-cycle MFMA phase, with it
// 1697 — the global -> LDS feed, not the MFMA / LDS schedule, holds every variant near 50 % matrix-pipe duty.
constexpr int BM3 = 256;
constexpr int IMG3A_BYTES = BM3 * 128, IMG3B_BYTES = BN * 128;
constexpr int BUF3_BYTES = IMG3A_BYTES + IMG3B_BYTES;  // A [hi | lo] rows, then B rows: 48 KB

template <class Epi>
__global__ __launch_bounds__(256, 2) void gemm3_nt_dma256_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                                                 int64_t M, int64_t N, int64_t Kp, int tiles_n, Epi epi) {
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

  // DMA: wave w stages A rows [64 w, 64 w + 64) (8 loads of 8 rows) and B rows [32 w, 32 w + 32) (4 loads).  Lane L
  // lands in row (L >> 3) of the load, slot L & 7, so it fetches chunk (L & 7) ^ ((row >> 1) & 7) of that row's line.
  const int lrow = lane >> 3;
  int64_t a_src[8], b_src[4];  // element offset of this lane's chunk in the first k-tile
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = w * 64 + i * 8 + lrow;
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    a_src[i] = (m0 + row < M ? m0 + row : M - 1) * 2 * Kp + chunk * 8;  // rows past the edge are clamped (never stored)
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = w * 32 + i * 8 + lrow;
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    b_src[i] = (n0 + row < N ? n0 + row : N - 1) * 2 * Kp + chunk * 8;
  }
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  auto dma_tile = [&](int64_t kt) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((glb_void*)(A + a_src[i] + kt * 64), (lds_void*)(smem + (w * 64 + i * 8) * 128), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((glb_void*)(B + b_src[i] + kt * 64),
                                       (lds_void*)(smem + IMG3A_BYTES + (w * 32 + i * 8) * 128), 16, 0, 0);
  };
  int a_off[4], a_sw[4], b_off[2], b_sw[2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ar = wm * 128 + t * 32 + li;
    a_off[t] = ar * 128;
    a_sw[t] = (ar >> 1) & 7;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int br = wn * 64 + t * 32 + li;
    b_off[t] = IMG3A_BYTES + br * 128;
    b_sw[t] = (br >> 1) & 7;
  }
  auto compute = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ks * 2 + lh;  // hi chunk of this lane's k-group; its lo chunk is 4 + c
      bf16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bh[t] = *reinterpret_cast<const bf16x8*>(smem + b_off[t] + ((c ^ b_sw[t]) << 4));
        bl[t] = *reinterpret_cast<const bf16x8*>(smem + b_off[t] + (((4 + c) ^ b_sw[t]) << 4));
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        ah[t] = *reinterpret_cast<const bf16x8*>(smem + a_off[t] + ((c ^ a_sw[t]) << 4));
        al[t] = *reinterpret_cast<const bf16x8*>(smem + a_off[t] + (((4 + c) ^ a_sw[t]) << 4));
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

  const int ntiles = (int)(Kp / BK);
  for (int kt = 0; kt < ntiles; ++kt) {
    dma_tile(kt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // the tile has landed for every wave
    compute();
    __syncthreads();  // every wave is done reading before the next DMA overwrites the buffer
  }

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      store_mfma_tile<true>(epi, m0 + wm * 128 + i * 32 + 4 * lh, n0 + wn * 64 + j * 32 + li, acc[i][j], M, N);
#ifdef SL_GEMM_CLOCKPROBE
  if (tid == 0) {
    epi.probe[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - probe_c0;
    epi.probe[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - probe_r0;
  }
#endif
}

// (kernel 3, the 256 x 256 ping-pong kernel of round 2, was superseded by gemm_8phase.hpp and lives in
// tools/native/gemm3_pingpong_lab.hpp for the lab harness; no dispatch of the library reaches it.)

inline int launch_split(const float* x, const float* scale, int64_t R, int64_t K, uint16_t* sp, hipStream_t st) {
  int64_t blocks = (R * split_kp(K) + 255) / 256;
  const int64_t cap = (int64_t)num_cus() * 16;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, scale, R, K, sp);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- column-strip split (round 5) --------------------------------------------------------------------------------------------
// The 256 x 256 kernel runs whole rounds of tiles; a trailing partial column tile (N = 1152 = 4.5 tiles: SigLIP-so400m's o-proj and
// fc2) costs a whole extra column of tiles, and with 64 row tiles (a 64-image call) 320 tiles are 1.25 rounds that take 1.77.  When
// the model below says so, the GEMM is cut at the last full column tile: columns [0, 256 n) keep the big kernel (256 tiles = ONE
// round) and the remainder strip — the same A, B rows from 256 n on, the epilogue shifted by 256 n columns — goes to whatever the
// grid-size rules pick for a strip that narrow (128 x 128 tiles).  Every output element keeps its accumulation order (all tile
// variants are bit-identical), so results do not change.  Measured (`tools/gemm_strip_lab.py`, profiles/r05_gemm_strip_lab.txt):
// M = 16 384: o-proj 173 -> 98 + 31 us, fc2 514 -> 310 + 96 us; M = 65 536: 495 -> 384 + 80, 1 523 -> 1 194 + 257.
// The cut may also fall one or two FULL tiles earlier when that lands the big kernel on whole rounds.
// Cost model, in rounds of the big kernel: t tiles cost floor(t / CUs) + (0.7 + 0.3 f) for a partial round filling a fraction f of
// the CUs; a strip of s 128 x 128 tiles costs 0.12 + 0.0014 s.  Option g3_strip_off = 1 switches the split off.
inline double g8_rounds_model(int64_t tiles, int64_t cus) {
  const int64_t full = tiles / cus, rem = tiles % cus;
  return (double)full + (rem ? 0.7 + 0.3 * (double)rem / (double)cus : 0.0);
}
inline int64_t strip_split_columns(int64_t M, int64_t N) {
  const bool on = option(OPT_G3_STRIP_OFF) == 0;
  const int64_t cus = num_cus();
  const int64_t tm = (M + 255) / 256, tn = (N + 255) / 256;
  if (!on || tn < 2) return 0;
  const double whole = g8_rounds_model(tm * tn, cus);
  // cut after m full column tiles, the strip up to three tiles wide: so400m's QKV at 64 images (N = 3456 = 13.5 tiles, 896 tiles =
  // 3.5 rounds) runs 12 column tiles in exactly three rounds and a 384-column strip
  int64_t best_m = 0;
  double best = 0.97 * whole;
  for (int64_t m = tn - 1; m >= 1 && m >= tn - 3; --m) {
    const int64_t rest = N - m * 256;
    const int64_t strip_tiles = ((M + 127) / 128) * ((rest + 127) / 128);
    const double cut = g8_rounds_model(tm * m, cus) + 0.12 + 0.0014 * (double)strip_tiles;
    if (cut < best) best = cut, best_m = m;
  }
  return best_m * 256;
}

// A (M rows), B (N rows): split matrices of K columns (layout above)
template <class Epi>
int launch_gemm3_nt(ProfScope& prof, const uint16_t* A, int64_t M, const uint16_t* B, int64_t N, int64_t K, const Epi& epi,
                    hipStream_t st, int may_split = 1) {  // 1: may cut a strip off, 0: the main part of a cut, 2: the strip itself
  const int64_t tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
  SL_REQUIRE(tm * tn < (1ll << 31), "GEMM: too many tiles");
  if (tm * tn == 0) return 0;
  const int64_t Kp = split_kp(K);
  // option "g3_tile" (sl_set_option; tests): 128: register-staged 128 x 128 tiles, 256: LDS-DMA staged 256 x 128 tiles, 8: 8-phase
  // 256 x 256, 160: 160 x 256 four-wave three-slot (gemm_w4.hpp), 64 / 1280: 64 x 64 eight-slot / 128 x 128 four-slot ring for small
  // grids (gemm_skinny.hpp); 0: by grid size
  const int forced = (int)option(OPT_G3_TILE);
  if (may_split == 2 && !forced && gemm8::fits(M, N, 4 * Kp) && Kp / 32 >= 8 && !gemmsk::prefer(M, N, Kp / 32))
    return gemmsk::launch<2>(prof, A, M, B, N, 4 * Kp, Kp / 32, epi, st);  // strips are priced (and run) as 128 x 128 tiles
  if (may_split == 1 && !forced && gemm8::fits(M, N, 4 * Kp) && gemm8::worth_it(M, N) && !gemmw4::prefer(M, N) &&
      !gemmsk::prefer(M, N, Kp / 32)) {
    const int64_t c0 = strip_split_columns(M, N);
    if (c0 > 0) {
      if (int rc = launch_gemm3_nt(prof, A, M, B, c0, K, epi, st, 0)) return rc;
      ProfScope strip(SL_PROF_GEMM, st, 0.0);  // its time counts, its flops are in `prof`'s work already
      return launch_gemm3_nt(strip, A, M, B + c0 * 2 * Kp, N - c0, K, epi.shifted(c0), st, 2);
    }
  }
  const int64_t tm3 = (M + BM3 - 1) / BM3;
  // gemm_skinny.hpp: 64 x 64 tiles behind an eight-stage LDS-DMA ring for small grids with long k loops
  if (gemm8::fits(M, N, 4 * Kp) && (forced ? forced == 64 : gemmsk::prefer(M, N, Kp / 32)))
    return gemmsk::launch<1>(prof, A, M, B, N, 4 * Kp, Kp / 32, epi, st);
  if (forced == 1280 && gemm8::fits(M, N, 4 * Kp)) return gemmsk::launch<2>(prof, A, M, B, N, 4 * Kp, Kp / 32, epi, st);
  // gemm_w4.hpp: 160 x 256 tiles where they shorten the makespan (150-tile GEMMs of the encoder: 240 items in one round)
  if (gemm8::fits(M, N, 4 * Kp) && (forced ? forced == 160 : gemmw4::prefer(M, N)))
    return gemmw4::launch<5>(prof, A, M, B, N, 4 * Kp, Kp / 32, epi, st);
  const bool k4 = gemm8::fits(M, N, 4 * Kp) && (forced ? forced == 8 : gemm8::worth_it(M, N));
  const bool k2 = forced ? forced == 256 : tm3 * tn >= (int64_t)8 * num_cus();
  if (k4)  // gemm_8phase.hpp; a row of a split matrix is 2 Kp bf16 = 4 Kp bytes, one 128-byte line per k-tile
    return gemm8::launch<gemm8::MODE_BF16X3>(prof, A, M, B, N, 4 * Kp, Kp / 32, epi, st);
  else if (k2)
    SL_LAUNCH(prof, (gemm3_nt_dma256_kernel<Epi>), dim3((unsigned)(tm3 * tn)), dim3(256), 0, st, A, B, M, N, Kp, (int)tn, epi);
  else if (!forced && gemm8::fits(M, N, 4 * Kp) && Kp / 32 >= 8)
    // mid-size grids: the same 128 x 128 tiles behind the four-stage LDS-DMA ring of gemm_skinny.hpp (5-30 % under the
    // register-staged kernel from 450 to 2 400 tiles, equal at 4 800; tools/enc_gemm_lab.py <M> with SL_G3_TILE = 128 / 1280)
    return gemmsk::launch<2>(prof, A, M, B, N, 4 * Kp, Kp / 32, epi, st);
  else
    SL_LAUNCH(prof, (gemm3_nt_kernel<Epi>), dim3((unsigned)(tm * tn)), dim3(256), 0, st, A, B, M, N, Kp, (int)tn, epi);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace gemm3
}  // namespace sl
