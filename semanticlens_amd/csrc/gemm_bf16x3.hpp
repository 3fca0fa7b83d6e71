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
// Two kernels, same accumulation order per output element (bit-identical results, tests/test_gpu_parity.py):
//   gemm3_nt_kernel         128 x 128 x 32 tile, 4 waves (2 x 2), wave tile 64 x 64; tiles staged through registers
//                           into four LDS images (A_hi, A_lo, B_hi, B_lo) with 80-byte rows (conflict-free fragment
//                           reads); single LDS stage + register prefetch, 40 KB, three workgroups per CU
//   gemm3_nt_dma256_kernel  256 x 128 x 32 tile, wave tile 128 x 64, tiles staged by LDS-DMA (see below); used for
//                           grids of >= 8 tiles per CU
// A and B fragments are read with the same (lane>>5)*8 + j k-pattern, so the result does not depend on how the
// hardware orders k inside a fragment.
#pragma once
#include "common.hpp"
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
__device__ inline void store_split(float v, int64_t row, int64_t col, int64_t Kp, uint16_t* __restrict__ sp) {
  const uint16_t h = f32_to_bf16_rne(v);
  const int64_t p = split_pos(row, col, Kp);
  sp[p] = h;
  sp[p + 32] = f32_to_bf16_rne(v - bf16_to_f32(h));
}
// four consecutive columns (col % 4 == 0): two 8-byte stores
__device__ inline void store_split4(const float4 y, int64_t row, int64_t col, int64_t Kp, uint16_t* __restrict__ sp) {
  const uint16_t h0 = f32_to_bf16_rne(y.x), h1 = f32_to_bf16_rne(y.y), h2 = f32_to_bf16_rne(y.z), h3 = f32_to_bf16_rne(y.w);
  const uint16_t l0 = f32_to_bf16_rne(y.x - bf16_to_f32(h0)), l1 = f32_to_bf16_rne(y.y - bf16_to_f32(h1));
  const uint16_t l2 = f32_to_bf16_rne(y.z - bf16_to_f32(h2)), l3 = f32_to_bf16_rne(y.w - bf16_to_f32(h3));
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

// ---- kernel 3: 256 x 256 block tile, two wave groups in ping-pong, four-stage LDS-DMA ring --------------------------
// A tile's LDS-DMA round trip is ~3000-5000 cycles under load while the L2 -> LDS path sustains > 20 TB/s
// (tools/native/feed_probe.hip): by Little's law kernel 2 would need ~124 KB of tile data in flight per CU to keep the
// matrix pipe busy, and its two 48 KB single buffers, in flight half of the time, allow about half of that.  This kernel
// lowers the demand (a 256 x 256 tile needs a third less data per flop) and keeps three of four 32 KB stages in flight:
//   * one 512-thread workgroup per CU = two groups of four waves (one wave of each group per SIMD); group g owns the
//     256 x 128 half of the output at columns 128 g; both groups share the A operand in LDS;
//   * a stage is ONE k-step (16 wide) of all 512 rows: per row 64 bytes [hi16 | lo16] (the two 32-byte halves of the
//     k-step inside the row's 128-byte line), staged by 1-KiB LDS-DMA loads of 16 rows; 16-byte slots XOR-swizzled by
//     (row >> 2) & 3 for conflict-free ds_read_b128;
//   * barrier-separated phases with opposite roles: in phase 2 s group 0 loads the 12 fragments of k-step s into
//     registers while group 1 runs the 24 MFMAs of k-step s - 1; in phase 2 s + 1 group 0 runs its 24 MFMAs while group 1
//     loads — every phase one wave per SIMD feeds the matrix pipe and the other uses the LDS;
//   * k-step s + 3 is issued at the start of phase 2 s + 1 (its stage was last read in phase 2 s - 1) and awaited with a
//     counted s_waitcnt vmcnt(8) at the end of phase 2 s + 3; raw s_barrier (__syncthreads would add vmcnt(0) and drain
//     the ring every phase).
// Same accumulation order per output element as kernels 1 and 2: bit-identical results.
// Measured (10000 x 9216 x 1152, tools/native/clock_probe.hip): 166 K cycles per 256 x 256 workgroup = 67 % matrix-pipe
// duty against 60 % for kernel 2, 457 vs 426 TFLOP/s on all-zero operands — but 355 vs 349 TFLOP/s on random operands,
// because the shader clock drops from 2.0 to 1.85 GHz: at ~1.05 PFLOP/s of issued bf16 MFMA work the part is at its power
// limit on real data.  Selected with SL_G3_TILE=512 only; kernel 2 stays the default for large grids.
constexpr int BM4 = 256, BN4 = 256;
constexpr int IMG4_BYTES = 256 * 64;          // 256 rows x [hi16 | lo16]
constexpr int STAGE4_BYTES = 2 * IMG4_BYTES;  // A rows, then B rows: 32 KB
constexpr int NSTAGE4 = 4;

template <class Epi>
__global__ __launch_bounds__(512, 2) void gemm3_nt_pingpong_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                                                   int64_t M, int64_t N, int64_t Kp, int tiles_n, Epi epi) {
  __shared__ __align__(1024) unsigned char smem[NSTAGE4 * STAGE4_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7
  const int grp = w >> 2, w4 = w & 3;
  const int wm = w4 >> 1, wn = w4 & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int tile = blockIdx.x;
  const int64_t m0 = (int64_t)(tile / tiles_n) * BM4;
  const int64_t n0 = (int64_t)(tile % tiles_n) * BN4;
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

  // DMA: one 1-KiB load covers 16 rows x 64 bytes; wave w stages rows [32 w, 32 w + 32) of A and of B (two loads each).
  // Lane L lands in row L >> 2 of the load, slot L & 3; slot s of a row holds chunk s ^ ((row >> 2) & 3), where chunks
  // 0, 1 are the two 16-byte halves of hi16 and 2, 3 those of lo16.  In the global line ([hi32 | lo32], 128 bytes) the
  // k-step's hi16 starts at byte 32 (s & 1) and its lo16 at 64 + 32 (s & 1).
  int64_t a_src[2], b_src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = w * 32 + i * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    const int eoff = (chunk >> 1) * 32 + (chunk & 1) * 8;  // element offset inside the line for an even k-step
    a_src[i] = (m0 + row < M ? m0 + row : M - 1) * 2 * Kp + eoff;  // rows past the edge are clamped (never stored)
    b_src[i] = (n0 + row < N ? n0 + row : N - 1) * 2 * Kp + eoff;
  }
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  auto dma_step = [&](int s) __attribute__((always_inline)) {  // k-step s -> stage s & 3
    const int64_t koff = (int64_t)(s >> 1) * 64 + (s & 1) * 16;
    unsigned char* l = smem + (s & (NSTAGE4 - 1)) * STAGE4_BYTES + w * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((glb_void*)(A + a_src[i] + koff), (lds_void*)(l + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void*)(B + b_src[i] + koff), (lds_void*)(l + IMG4_BYTES + i * 1024), 16, 0, 0);
    }
  };
  int a_off[4], b_off[2];  // byte offset of this lane's hi fragment inside a stage; the lo fragment is 2 slots further (XOR 2)
  int a_lo[4], b_lo[2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ar = wm * 128 + t * 32 + li, sw = (ar >> 2) & 3;
    a_off[t] = ar * 64 + ((lh ^ sw) << 4);
    a_lo[t] = ar * 64 + (((2 + lh) ^ sw) << 4);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int br = grp * 128 + wn * 64 + t * 32 + li, sw = (br >> 2) & 3;
    b_off[t] = IMG4_BYTES + br * 64 + ((lh ^ sw) << 4);
    b_lo[t] = IMG4_BYTES + br * 64 + (((2 + lh) ^ sw) << 4);
  }
  bf16x8 fah[4], fal[4], fbh[2], fbl[2];
  auto load_phase = [&](int s) __attribute__((always_inline)) {
    const unsigned char* base = smem + (s & (NSTAGE4 - 1)) * STAGE4_BYTES;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      fbh[t] = *reinterpret_cast<const bf16x8*>(base + b_off[t]);
      fbl[t] = *reinterpret_cast<const bf16x8*>(base + b_lo[t]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      fah[t] = *reinterpret_cast<const bf16x8*>(base + a_off[t]);
      fal[t] = *reinterpret_cast<const bf16x8*>(base + a_lo[t]);
    }
  };
  auto mfma_phase = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[i], fbh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbh[j], acc[i][j], 0, 0, 0);
      }
  };

  // Phases p = 0 .. 2 ns (one barrier each).  Group 0: p = 2 s loads k-step s, p = 2 s + 1 computes it, the last phase is
  // empty.  Group 1: p = 0 is empty, p = 2 s + 1 loads k-step s, p = 2 s + 2 computes it.  DMA (all eight waves, four
  // loads each): k-steps 0..2 up front, k-step s + 3 at the start of phase 2 s + 1.  The end of phase 2 s + 1 waits until
  // k-step s + 1 has landed: the loads of k-steps s + 2 and s + 3 (8 per wave) may still be in flight.
  const int ns = (int)(Kp / 16);
  auto phase_barrier = [&]() __attribute__((always_inline)) {
    // pinned for the scheduler: hipcc otherwise lets the register-only MFMAs of a phase trail past the barrier into the
    // group's next (load) phase, where they collide with the other group's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto odd_phase_end = [&](int s) __attribute__((always_inline)) {  // end of phase 2 s + 1: k-step s + 1 must be complete
    __builtin_amdgcn_sched_barrier(0);
    if (s + 3 < ns) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if (s + 2 < ns) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    phase_barrier();
  };
  auto even_phase_end = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    phase_barrier();
  };
  if (ns > 0) {
    dma_step(0);
    if (ns > 1) dma_step(1);
    if (ns > 2) dma_step(2);
    if (ns > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ns > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    phase_barrier();
    if (grp == 0) {
      for (int s = 0; s < ns; ++s) {
        load_phase(s);
        even_phase_end();
        if (s + 3 < ns) dma_step(s + 3);
        mfma_phase();
        odd_phase_end(s);
      }
      even_phase_end();
    } else {
      even_phase_end();
      for (int s = 0; s < ns; ++s) {
        if (s + 3 < ns) dma_step(s + 3);
        load_phase(s);
        odd_phase_end(s);
        mfma_phase();
        even_phase_end();
      }
    }
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t col = n0 + grp * 128 + wn * 64 + j * 32 + li;
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

// ---- kernel 4: 256 x 256 block tile, 8 waves (2 x 4), two staggered wave groups, 16-KB pieces in a 128-KB ring -------
// The schedule of the guide's 256^2 8-phase bf16 kernel (cdna_hip_programming.md "The 256^2 8-phase template"),
// re-derived for split operands.  A *stage* is one 32-wide k-tile of all 512 rows = 512 whole 128-byte [hi32 | lo32]
// lines = 64 KB (A image, then B image); LDS holds two stages.  A wave owns a 128 x 64 piece of the output (4 x 2
// MFMA tiles, 128 accumulator registers) and runs a stage in FOUR phases of 12 MFMAs (384 matrix-pipe cycles):
//   phase p = 2 ih + ks + 1:  row tiles i in {2 ih, 2 ih + 1}, k-step ks of the stage, both column tiles.
// B fragments of a stage are kept in VGPRs (k-step 1 read in phase 1, k-step 0 in phase 4 of the stage before); A
// fragments (4 reads) are read per phase.
// * Wave groups: waves 0-3 (rows 0-127) and 4-7 (rows 128-255) sit one per SIMD each.  Every phase is
//   [read slot] s_barrier [MFMA slot] s_barrier, and group 1 runs one barrier behind group 0, so between any two
//   barriers one wave of each SIMD feeds the matrix pipe while the other reads fragments and issues LDS-DMA.
// * Feed: a stage is four 16-KB *pieces* (B rows 0-127, B rows 128-255, A rows of ih = 0, A rows of ih = 1); a piece
//   is 16 LDS-DMA instructions of 8 whole lines, two per wave.  Pieces are issued in consumption order, one per phase,
//   SIX pieces ahead: phase g issues piece g + 6 into the ring slot of piece g - 2, whose last read was in phase
//   <= g - 1 (B: phase 4 of the stage before and phase 1 of its stage, A(ih): phases 2 ih + 1, 2 ih + 2) and was retired by the lgkmcnt(0) that
//   precedes that phase's barrier.  5-7 pieces (80-112 KB per CU) are in flight at any time, ~5 phases ~ 2 us of lead:
//   the depth the 48-KB single-buffer kernel 2 lacks.
// * Waits: counted, never zero in the steady state.  What phase g + 1 reads must have been waited for in phase g's
//   read slot, before its barrier (the guide's "read one phase after the wait"): phase 2 of a stage waits vmcnt(8)
//   (A1 of this stage; four younger pieces may fly), phase 3 vmcnt(6) (B0, B1 of the next stage), phase 4 vmcnt(6)
//   (A0 of the next stage).  Raw s_barrier only: __syncthreads() would add vmcnt(0) and drain the ring.
// * LDS image: unpadded 128-byte rows, 16-byte slots XOR-swizzled by (row >> 1) & 7 on the DMA's SOURCE address and on
//   the fragment read (as kernel 2): conflict-free ds_read_b128.
// Per output element the products are accumulated in the same order as in kernels 1-3: bit-identical results.
constexpr int BM5 = 256, BN5 = 256;
constexpr int IMG5_BYTES = 256 * 128;         // one operand of one stage: 32 KB
constexpr int STAGE5_BYTES = 2 * IMG5_BYTES;  // A image, B image

template <int N_>
struct IntC {
  static constexpr int value = N_;
};

__device__ __forceinline__ void wait_vm_pieces(int n) {  // at most n pieces (2 loads each) of this wave still in flight
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
  }
}

// SWZ: 0 = tiles in launch order (row-major), 1 = every XCD gets a contiguous range of tiles, walked in bands of
// GROUP_M tile rows (neighbouring workgroups of one L2 share A and B panels)
// ABLATE (measurement only, results are garbage): 1 = no LDS-DMA in the k loop, 2 = no fragment reads, 4 = no barriers,
// 8 = LDS-DMA issued but never waited for, 16 = LDS-DMA always re-reads stage 0 (cache-hot source),
// 32 = half of the LDS-DMA instructions, 64 = LDS-DMA of 4 bytes per lane instead of 16 (waits disabled with 8)
// DMAPOS: 0 = the phase's two LDS-DMA instructions are issued in the read slot, 1 = by the same wave inside its MFMA
// slot (after the 4th and the 8th MFMA), 2 = after the 2nd and 3rd MFMA
template <class Epi, int SWZ, int EARLY = 0, int ABLATE = 0, int LOOK = 6, int DMAPOS = 0>
__global__ __launch_bounds__(512, 2) void gemm3_nt_8phase_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                                                 int64_t M, int64_t N, int64_t Kp, int tiles_m, int tiles_n, Epi epi) {
  __shared__ __align__(1024) unsigned char smem[2 * STAGE5_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7
  const int wr = w >> 2, wc = w & 3;
  const int li = lane & 31, lh = lane >> 5;
  int tile = blockIdx.x;
  int tm_i, tn_i;
  if (SWZ == 1) {
    const int nwg = tiles_m * tiles_n;
    const int xcd = tile & 7, idx = tile >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;  // bijective for any nwg
    constexpr int GROUP_M = 4;
    const int band = tile / (GROUP_M * tiles_n);
    const int first_m = band * GROUP_M;
    const int rows = tiles_m - first_m < GROUP_M ? tiles_m - first_m : GROUP_M;
    const int in_band = tile - band * GROUP_M * tiles_n;
    tm_i = first_m + in_band % rows;
    tn_i = in_band / rows;
  } else {
    tm_i = tile / tiles_n;
    tn_i = tile % tiles_n;
  }
  const int64_t m0 = (int64_t)tm_i * BM5;
  const int64_t n0 = (int64_t)tn_i * BN5;
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

  // ---- LDS-DMA plan.  Piece q of a stage: 0 = B rows 0-127, 1 = B rows 128-255, 2 = A rows {0-63, 128-191} (ih = 0 of
  // both wave groups), 3 = A rows {64-127, 192-255}.  Wave w moves row groups 2 w and 2 w + 1 (8 rows each) of a piece.
  // Lane L lands in row L >> 3 of its group, slot L & 7, and fetches chunk (L & 7) ^ ((row >> 1) & 7) of the row's line.
  int64_t src[4][2];  // element offset of this lane's chunk in the first k-tile
  int dst[4][2];      // wave-uniform LDS byte offset of the row group inside a stage
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int rg = w * 2 + g;  // 0..15
      int row0;
      if (q < 2) row0 = q * 128 + rg * 8;
      else row0 = (rg < 8 ? rg * 8 : 128 + (rg - 8) * 8) + (q - 2) * 64;
      const int row = row0 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      if (q < 2) {
        src[q][g] = (n0 + row < N ? n0 + row : N - 1) * 2 * Kp + chunk * 8;  // rows past the edge are clamped (never stored)
        dst[q][g] = IMG5_BYTES + row0 * 128;
      } else {
        src[q][g] = (m0 + row < M ? m0 + row : M - 1) * 2 * Kp + chunk * 8;
        dst[q][g] = row0 * 128;
      }
    }
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int npieces = (int)(Kp / 32) * 4;
  // piece n = 4 stage + q -> ring slot (stage & 1, q)
  auto issue = [&](int stage, auto Qc, int g_lo = 0, int g_hi = 2) __attribute__((always_inline)) {
    constexpr int q = decltype(Qc)::value;
    const uint16_t* base = q < 2 ? B : A;
    unsigned char* l = smem + (stage & 1) * STAGE5_BYTES;
#pragma unroll
    for (int g = g_lo; g < ((ABLATE & 32) ? 1 : g_hi); ++g) {
      if (ABLATE & 64)
        __builtin_amdgcn_global_load_lds((glb_void*)(base + src[q][g] + (int64_t)stage * 64), (lds_void*)(l + dst[q][g]), 4, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((glb_void*)(base + src[q][g] + ((ABLATE & 16) ? (int64_t)0 : (int64_t)stage * 64)),
                                         (lds_void*)(l + dst[q][g]), 16, 0, 0);
    }
  };

  // ---- fragment addresses: hi chunk of k-step 0 for this lane; k-step 1 is the address ^ 32, the lo half ^ 64
  int a_addr[4], b_addr[2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ar = wr * 128 + t * 32 + li;
    a_addr[t] = ar * 128 + ((lh ^ ((ar >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int br = wc * 64 + t * 32 + li;
    b_addr[t] = IMG5_BYTES + br * 128 + ((lh ^ ((br >> 1) & 7)) << 4);
  }
  bf16x8 bh[2][2], bl[2][2];  // [ks][j]: B fragments of the whole stage
  bf16x8 ah[2], al[2];        // A fragments of the current phase

  auto raw_barrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    if (!(ABLATE & 4)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // one phase; P = 1..4, TAIL: the stage is one of the last two (pieces to issue may not exist, waits are exact)
  auto phase = [&](int stage, auto Pc, auto Tc) __attribute__((always_inline)) {
    constexpr int P = decltype(Pc)::value;
    constexpr bool TAIL = decltype(Tc)::value != 0;
    constexpr int ih = (P - 1) >> 1, ks = (P - 1) & 1;
    const unsigned char* buf = smem + (stage & 1) * STAGE5_BYTES;
    // ---- read slot: A fragments of this phase; the B fragments a stage needs are spread over the two phases whose MFMAs
    // do not use the registers being refilled (k-step 1 of this stage in phase 1, k-step 0 of the NEXT stage in phase 4):
    // 8 / 4 / 4 / 8 ds_read_b128 per phase.  (All 12 B + A reads of a stage in phase 1 made that slot LDS-bandwidth
    // bound: 4 waves x 12 KB = 384 cycles at 128 B/clk, 455 cycles per slot on average instead of 384.)
    if (P == 1 && !(ABLATE & 2)) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh[1][j] = *reinterpret_cast<const bf16x8*>(buf + (b_addr[j] ^ 32));
        bl[1][j] = *reinterpret_cast<const bf16x8*>(buf + (b_addr[j] ^ 32 ^ 64));
      }
    }
    if (!(ABLATE & 2)) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        ah[t] = *reinterpret_cast<const bf16x8*>(buf + (a_addr[2 * ih + t] ^ (ks * 32)));
        al[t] = *reinterpret_cast<const bf16x8*>(buf + (a_addr[2 * ih + t] ^ (ks * 32) ^ 64));
      }
    } else {  // keep the stale fragments live and opaque so the MFMAs are not folded
      asm volatile("" : "+v"(ah[0]), "+v"(ah[1]), "+v"(al[0]), "+v"(al[1]));
    }
    if (P == 4 && !(ABLATE & 2) && (!TAIL || stage + 1 < (int)(Kp / 32))) {
      const unsigned char* nbuf = smem + ((stage + 1) & 1) * STAGE5_BYTES;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh[0][j] = *reinterpret_cast<const bf16x8*>(nbuf + b_addr[j]);
        bl[0][j] = *reinterpret_cast<const bf16x8*>(nbuf + (b_addr[j] ^ 64));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const int g = stage * 4 + P;  // global phase number; issues piece g + LOOK
    // What phase g + 1 reads must have landed for every wave before this phase's barrier.  `need` = youngest such piece;
    // pieces need + 1 .. g + LOOK - 1 (issued so far) may stay in flight.
    if (P == 2 || P == 3 || P == 4) {
      // P == 2: A1 of this stage (piece g + 1); P == 3: B0, B1 of the next stage (up to piece g + 2);
      // P == 4: A0 of the next stage (piece g + 2)
      constexpr int younger = P == 2 ? LOOK - 2 : LOOK - 3;
      const int need = P == 2 ? g + 1 : g + 2;
      if (ABLATE & 8) {
      } else if (!TAIL) {
        wait_vm_pieces(younger);  // compile-time constant: a single s_waitcnt
      } else {
        const int have = npieces - 1 - need;
        wait_vm_pieces(have < younger ? (have < 0 ? 0 : have) : younger);
      }
    }
    constexpr int q = (P + LOOK) & 3;               // which piece of its stage
    const int pstage = stage + ((P + LOOK) >> 2);   // and which stage
    const bool do_issue = !(ABLATE & 1) && (!TAIL || pstage * 4 + q < npieces);
    if (DMAPOS == 0 && do_issue) issue(pstage, IntC<q>());
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments in registers; this phase's LDS reads are retired
    raw_barrier();
    // ---- MFMA slot.  Order: per accumulator still lo*hi, hi*lo, hi*hi (bit-identical to kernels 1-3), but the A operand
    // changes only four times per phase and dependent MFMAs are two apart.  The closing barrier sits EARLY MFMAs before
    // the end of the slot: the other group is released while this wave still has work queued on the matrix pipe, so the
    // barrier's release latency (~70 cycles per slot when it followed the last MFMA) is covered.
    __builtin_amdgcn_s_setprio(1);
#define SL_G5_MFMA(a, b, t, j) acc[2 * ih + t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[2 * ih + t][j], 0, 0, 0)
#define SL_G5_DMA(g)                                  \
  do {                                                \
    __builtin_amdgcn_sched_barrier(0);                \
    if (do_issue) issue(pstage, IntC<q>(), g, g + 1); \
    __builtin_amdgcn_sched_barrier(0);                \
  } while (0)
    SL_G5_MFMA(al[0], bh[ks][0], 0, 0);
    SL_G5_MFMA(al[0], bh[ks][1], 0, 1);
    if (DMAPOS == 2) SL_G5_DMA(0);
    SL_G5_MFMA(al[1], bh[ks][0], 1, 0);
    if (DMAPOS == 2) SL_G5_DMA(1);
    SL_G5_MFMA(al[1], bh[ks][1], 1, 1);
    if (DMAPOS == 1) SL_G5_DMA(0);
    SL_G5_MFMA(ah[0], bl[ks][0], 0, 0);
    SL_G5_MFMA(ah[0], bl[ks][1], 0, 1);
    SL_G5_MFMA(ah[0], bh[ks][0], 0, 0);
    SL_G5_MFMA(ah[0], bh[ks][1], 0, 1);
    if (DMAPOS == 1) SL_G5_DMA(1);
    if (EARLY == 4) raw_barrier();
    SL_G5_MFMA(ah[1], bl[ks][0], 1, 0);
    if (EARLY == 3) raw_barrier();
    SL_G5_MFMA(ah[1], bl[ks][1], 1, 1);
    if (EARLY == 2) raw_barrier();
    SL_G5_MFMA(ah[1], bh[ks][0], 1, 0);
    if (EARLY == 1) raw_barrier();
    SL_G5_MFMA(ah[1], bh[ks][1], 1, 1);
#undef SL_G5_MFMA
#undef SL_G5_DMA
    if (EARLY == 0) {
      __builtin_amdgcn_s_setprio(0);
      raw_barrier();
    } else {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(0);
    }
  };
  auto run_stage = [&](int stage, auto Tc) __attribute__((always_inline)) {
    phase(stage, IntC<1>(), Tc);
    phase(stage, IntC<2>(), Tc);
    phase(stage, IntC<3>(), Tc);
    phase(stage, IntC<4>(), Tc);
  };

  const int ns = (int)(Kp / 32);
#ifdef SL_GEMM_CLOCKPROBE
  unsigned long long probe_c1 = 0;
#endif
  if (ns > 0) {
    // prologue ("phase 0"): pieces 0..LOOK, then B0, B1, A0 of stage 0 must have landed
    static_assert(LOOK >= 3 && LOOK <= 6, "LOOK");
    issue(0, IntC<0>());
    issue(0, IntC<1>());
    issue(0, IntC<2>());
    issue(0, IntC<3>());
    if (ns > 1) {
      if (LOOK >= 4) issue(1, IntC<0>());
      if (LOOK >= 5) issue(1, IntC<1>());
      if (LOOK >= 6) issue(1, IntC<2>());
    }
    wait_vm_pieces(ns > 1 ? LOOK - 2 : 1);
    raw_barrier();
#ifdef SL_GEMM_CLOCKPROBE
    probe_c1 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int j = 0; j < 2; ++j) {  // B fragments of k-step 0 of stage 0 (later stages: phase 4 of the stage before)
      bh[0][j] = *reinterpret_cast<const bf16x8*>(smem + b_addr[j]);
      bl[0][j] = *reinterpret_cast<const bf16x8*>(smem + (b_addr[j] ^ 64));
    }
    if (wr == 1) raw_barrier();  // group 1 runs one barrier behind
    int s = 0;
    for (; s + 2 <= ns - 2; s += 2) {
      run_stage(s, IntC<0>());
      run_stage(s + 1, IntC<0>());
    }
    for (; s < ns; ++s) run_stage(s, IntC<1>());
    if (wr == 0) raw_barrier();
  }
#ifdef SL_GEMM_CLOCKPROBE
  const unsigned long long probe_c2 = __builtin_amdgcn_s_memtime();
#endif

  // interior tiles (all but the last band / column of tiles) store without per-element predicates: the predicated form
  // costs ~10 VALU/branch instructions per 4-byte store and made the epilogue 7 % of a tile (VALU-bound, not store-bound)
  if (m0 + BM5 <= M && n0 + BN5 <= N) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int64_t col = n0 + wc * 64 + j * 32 + li;
        const float cv = epi.column(col);
        const int64_t row_base = m0 + wr * 128 + i * 32 + 4 * lh;
#pragma unroll
        for (int r = 0; r < 16; ++r) epi.store(row_base + ((r & 3) + 8 * (r >> 2)), col, acc[i][j][r], cv);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int64_t col = n0 + wc * 64 + j * 32 + li;
        const float cv = col < N ? epi.column(col) : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = m0 + wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < M && col < N) epi.store(row, col, acc[i][j][r], cv);
        }
      }
    }
  }
#ifdef SL_GEMM_CLOCKPROBE
  if (tid == 0) {
    epi.probe[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - probe_c0;
    epi.probe[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - probe_r0;
    epi.probe[131072 + 2 * blockIdx.x] = probe_c1 - probe_c0;      // prologue: first pieces landed
    epi.probe[131072 + 2 * blockIdx.x + 1] = probe_c2 - probe_c1;  // k loop
  }
#endif
}

inline int launch_split(const float* x, const float* scale, int64_t R, int64_t K, uint16_t* sp, hipStream_t st) {
  int64_t blocks = (R * split_kp(K) + 255) / 256;
  const int64_t cap = (int64_t)num_cus() * 16;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, scale, R, K, sp);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

// A (M rows), B (N rows): split matrices of K columns (layout above)
template <class Epi>
int launch_gemm3_nt(ProfScope& prof, const uint16_t* A, int64_t M, const uint16_t* B, int64_t N, int64_t K, const Epi& epi,
                    hipStream_t st) {
  const int64_t tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
  SL_REQUIRE(tm * tn < (1ll << 31), "GEMM: too many tiles");
  if (tm * tn == 0) return 0;
  const int64_t Kp = split_kp(K);
  static const int forced = [] {
    // 128: register-staged 128 x 128 tiles (kernel 1), 256: LDS-DMA staged 256 x 128 tiles (kernel 2),
    // 512: ping-pong 256 x 256 (kernel 3), 8: 8-phase 256 x 256 (kernel 4); unset: by grid size
    const char* e = getenv("SL_G3_TILE");
    return e ? atoi(e) : 0;
  }();
  static const int min8 = [] {  // kernel 4 from this many 256 x 256 tiles per CU (in percent) upwards
    const char* e = getenv("SL_G3_MIN8_PCT");
    return e ? atoi(e) : 50;  // encoder GEMMs (150-600 tiles): ViT-B/32 image encode 9.66 ms without kernel 4, 9.08 at 50 %
  }();
  const int64_t tm3 = (M + BM3 - 1) / BM3;
  const int64_t tm5 = (M + BM5 - 1) / BM5, tn5 = (N + BN5 - 1) / BN5;
  if (forced == 512) {
    SL_LAUNCH(prof, (gemm3_nt_pingpong_kernel<Epi>), dim3((unsigned)(tm3 * tn5)), dim3(512), 0, st, A, B, M, N, Kp, (int)tn5, epi);
    SL_CHECK_HIP(hipGetLastError());
    return 0;
  }
  const bool k4 = forced ? forced == 8 : tm5 * tn5 * 100 >= (int64_t)min8 * num_cus();
  const bool k2 = forced ? forced == 256 : tm3 * tn >= (int64_t)8 * num_cus();
  if (k4)  // XCD-banded tile order, DMA issued inside the MFMA slot
    SL_LAUNCH(prof, (gemm3_nt_8phase_kernel<Epi, 1, 0, 0, 6, 2>), dim3((unsigned)(tm5 * tn5)), dim3(512), 0, st, A, B, M, N, Kp, (int)tm5,
              (int)tn5, epi);
  else if (k2)
    SL_LAUNCH(prof, (gemm3_nt_dma256_kernel<Epi>), dim3((unsigned)(tm3 * tn)), dim3(256), 0, st, A, B, M, N, Kp, (int)tn, epi);
  else
    SL_LAUNCH(prof, (gemm3_nt_kernel<Epi>), dim3((unsigned)(tm * tn)), dim3(256), 0, st, A, B, M, N, Kp, (int)tn, epi);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace gemm3
}  // namespace sl
