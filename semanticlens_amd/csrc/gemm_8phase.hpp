// 256 x 256 "8-phase" NT GEMM for the MI355X matrix cores, in two arithmetic modes that share one feed schedule:
//   MODE_BF16X3  split-bf16 operands (gemm_bf16x3.hpp), three v_mfma_f32_32x32x16_bf16 per product
//   MODE_F32     fp32 operands, v_mfma_f32_32x32x2_f32 (exact fp32 fma chain)
// In both modes a 32-wide k-tile of one operand row is ONE 128-byte line in memory ([hi32 | lo32] bf16, or 32 floats),
// so tiles, LDS images, the DMA plan and the ring are identical; only the fragment reads and the MFMA block differ.
//
// The schedule is the guide's 256^2 8-phase bf16 kernel (cdna_hip_programming.md "The 256^2 8-phase template")
// re-derived for these operands:
// * Workgroup = 512 threads = 8 waves (2 x 4); a wave owns a 128 x 64 piece of the output (4 x 2 MFMA tiles of 32 x 32,
//   128 accumulator registers).  A *stage* is one k-tile of all 512 rows = 512 lines = 64 KB (A image, then B image);
//   LDS holds two stages (128 KB, one workgroup per CU).
// * A stage is computed in FOUR phases, phase p = 2 ih + kh + 1: row tiles i in {2 ih, 2 ih + 1}, k-half kh of the stage,
//   both column tiles (bf16x3: 12 MFMAs = 384 matrix-pipe cycles; f32: 32 MFMAs = 2048 cycles).  A fragments are read
//   per phase (4 ds_read_b128); B fragments of a stage live in VGPRs (k-half 1 is read in phase 1, k-half 0 in phase 4
//   of the stage before — the phases whose MFMAs do not use the registers being refilled).
// * Wave groups: waves 0-3 (rows 0-127) and 4-7 (rows 128-255) sit one per SIMD each.  Every phase is
//   [read slot] s_barrier [MFMA slot] s_barrier and group 1 runs one barrier behind group 0: between two barriers one
//   wave of each SIMD feeds the matrix pipe while the other reads its fragments.
// * Feed: a stage is four 16-KB *pieces* (B rows 0-127, B rows 128-255, A rows of ih = 0, A rows of ih = 1); a piece is
//   16 LDS-DMA instructions of eight whole lines, two per wave.  Pieces are issued in consumption order, one per phase,
//   SIX ahead: phase g issues piece g + 6 into the ring slot of piece g - 2, whose last read was in phase <= g - 1 and
//   was retired by the lgkmcnt(0) in front of that phase's barrier.  80-112 KB per CU are in flight.
//   The two DMA instructions of a phase are issued BY THE WAVE THAT IS IN ITS MFMA SLOT, between its MFMAs.  Issued from
//   the read slot of the other wave of the SIMD each of them stalled the MFMA stream by ~31 cycles whatever its size,
//   lookahead or source (measured: 452 cycles per slot instead of 384; 407 with the DMA inside the MFMA slot; 390 with
//   no DMA at all — tools/native/gemm3_lab.hip).
// * Waits are counted, never zero in the steady state.  What phase g + 1 reads must have been waited for in phase g's
//   read slot, before its barrier (the guide's "read one phase after the wait"): phase 2 waits vmcnt(8) (A1 of this
//   stage; four younger pieces may fly), phases 3 and 4 vmcnt(6) (B0/B1, then A0 of the next stage).  Raw s_barrier
//   only: __syncthreads() would add vmcnt(0) and drain the ring.
// * LDS image: unpadded 128-byte rows, 16-byte slots XOR-swizzled by (row >> 1) & 7 on the DMA's SOURCE address and on
//   the fragment read: conflict-free ds_read_b128.
// * One tile per workgroup; tile order: every XCD gets a contiguous range of tiles, walked in bands of four tile rows, so the workgroups that
//   share an L2 share A and B panels (+2-3 % here).
// * Interior tiles store without per-element predicates (the predicated epilogue was VALU-bound: 10.4 K vs 4.6 K cycles).
// Per output element the products are accumulated in the same order as in the 128 x 128 kernels of gemm_bf16x3.hpp /
// gemm_f32.hpp: results are bit-identical to theirs (tests/test_gpu_parity.py::test_gemm_tile_variants_are_bit_identical).
//
// Measured, 10000 x 9216 x 1152 on normalised random operands (tools/native/gemm3_lab.hip): bf16x3 454 TFLOP/s
// algorithmic = 1.36 PFLOP/s of bf16 MFMA work issued; in-tile matrix-pipe duty 0.88 at a shader clock of 1.70 GHz (the
// part is power-limited: the 256 x 128 kernel runs 0.60 duty at 2.0 GHz).
#pragma once
#include "common.hpp"
#include "gemm_epilogue.hpp"

namespace sl {
namespace gemm8 {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// One 16-byte fragment register quad, either mode.  It must be an ext_vector type: read through HIP's struct `uint4`, hipcc
// (ROCm 7.2) cannot tell the fragment reads from the LDS-DMA writes apart and puts `s_waitcnt vmcnt(0)` in front of the
// first ds_read of every phase, which drains the ring (455 instead of 407 cycles per slot).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int MODE_BF16X3 = 0, MODE_F32 = 1;
constexpr int BM = 256, BN = 256;
constexpr int IMG_BYTES = 256 * 128;        // one operand of one stage: 32 KB
constexpr int STAGE_BYTES = 2 * IMG_BYTES;  // A image, B image
constexpr int LOOK = 6;                     // pieces issued ahead of the phase that runs

template <int N_>
struct IntC {
  static constexpr int value = N_;
};

// An epilogue that reads memory per element (LinearEpi with a residual / positional table) exposes
// `static constexpr bool kFetches = true`, `fetch(row, col)` and `store_fetched(row, col, acc, column, fetched)`.
__device__ __forceinline__ void wait_vm_pieces(int n) {  // at most n pieces (2 loads each) of this wave still in flight
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
  }
}

// A (M rows), B (N rows): byte pointers; a row is `row_bytes` bytes and its k-tile kt is the 128-byte line at kt * 128.
// NO_DMA (measurement only, garbage results): the k loop issues no LDS-DMA — what the feed costs the MFMA stream.
template <int MODE, class Epi, bool NO_DMA = false>
__global__ __launch_bounds__(512, 2) void gemm_nt_8phase_kernel(const unsigned char* __restrict__ A, const unsigned char* __restrict__ B,
                                                                int64_t M, int64_t N, int64_t row_bytes, int ns, int tiles_m,
                                                                int tiles_n, Epi epi) {
  __shared__ __align__(1024) unsigned char smem[2 * STAGE_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7
  const int wr = w >> 2, wc = w & 3;
  const int li = lane & 31, lh = lane >> 5;
  int tm_i, tn_i;
  {  // XCD-aware tile order (bijective for any grid): XCD x = blockIdx % 8 owns a contiguous range of tiles, walked in
     // bands of GROUP_M tile rows so that the workgroups sharing an L2 share A and B panels
    const int nwg = tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    constexpr int GROUP_M = 4;
    const int band = tile / (GROUP_M * tiles_n);
    const int first_m = band * GROUP_M;
    const int rows = tiles_m - first_m < GROUP_M ? tiles_m - first_m : GROUP_M;
    const int in_band = tile - band * GROUP_M * tiles_n;
    tm_i = first_m + in_band % rows;
    tn_i = in_band / rows;
  }
  const int64_t m0 = (int64_t)tm_i * BM;
  const int64_t n0 = (int64_t)tn_i * BN;
#ifdef SL_GEMM_CLOCKPROBE  // tools/native/gemm3_lab.hip
  const unsigned long long probe_c0 = __builtin_amdgcn_s_memtime(), probe_r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long probe_c1 = 0, probe_c2 = 0;
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
  uint32_t src[4][2];  // byte offset of this lane's chunk in the first k-tile (operands < 4 GB: checked by launch())
  int dst[4][2];       // wave-uniform LDS byte offset of the row group inside a stage
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
        src[q][g] = (uint32_t)((n0 + row < N ? n0 + row : N - 1) * row_bytes + chunk * 16);  // rows past the edge are clamped (never stored)
        dst[q][g] = IMG_BYTES + row0 * 128;
      } else {
        src[q][g] = (uint32_t)((m0 + row < M ? m0 + row : M - 1) * row_bytes + chunk * 16);
        dst[q][g] = row0 * 128;
      }
    }
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int npieces = ns * 4;
  // instruction g of piece n = 4 stage + q -> ring slot (stage & 1, q)
  auto issue = [&](int stage, auto Qc, int g) __attribute__((always_inline)) {
    constexpr int q = decltype(Qc)::value;
    // wave-uniform base (SGPR pair) + 32-bit lane offset: the saddr form of global_load_lds, no 64-bit VALU address math
    const unsigned char* base = (q < 2 ? B : A) + (int64_t)stage * 128;
    unsigned char* l = smem + (stage & 1) * STAGE_BYTES;
    __builtin_amdgcn_global_load_lds((glb_void*)(base + src[q][g]), (lds_void*)(l + dst[q][g]), 16, 0, 0);
  };

  // ---- fragment addresses (first 16-byte chunk this lane reads in a stage; the others are XORs of it)
  //   bf16x3: chunk c = 2 kh + lh holds hi k [8 c, 8 c + 8) of the stage, chunk 4 + c the lo halves: kh -> ^32, lo -> ^64
  //   f32:    chunk c = 4 lh + u holds k [16 lh + 4 u, +4), u = 2 kh + (0|1):                 u&1 -> ^16, kh -> ^32
  int a_addr[4], b_addr[2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ar = wr * 128 + t * 32 + li;
    const int c0 = MODE == MODE_BF16X3 ? lh : 4 * lh;
    a_addr[t] = ar * 128 + ((c0 ^ ((ar >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int br = wc * 64 + t * 32 + li;
    const int c0 = MODE == MODE_BF16X3 ? lh : 4 * lh;
    b_addr[t] = IMG_BYTES + br * 128 + ((c0 ^ ((br >> 1) & 7)) << 4);
  }
  // fragments: [k-half][column tile][0|1] and [row tile of the phase][0|1]; the last index is hi / lo (bf16x3) or the
  // two float4 of the k-half (f32).  uint4 carries either.
  u32x4 fb[2][2][2], fa[2][2];
  constexpr int SUB = MODE == MODE_BF16X3 ? 64 : 16;  // address XOR between the two fragments of a (tile, k-half)

  auto raw_barrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // one phase; P = 1..4, TAIL: the stage is one of the last two (pieces to issue may not exist, waits are exact)
  auto phase = [&](int stage, auto Pc, auto Tc) __attribute__((always_inline)) {
    constexpr int P = decltype(Pc)::value;
    constexpr bool TAIL = decltype(Tc)::value != 0;
    constexpr int ih = (P - 1) >> 1, kh = (P - 1) & 1;
    const unsigned char* buf = smem + (stage & 1) * STAGE_BYTES;
    // ---- read slot: 8 / 4 / 4 / 8 ds_read_b128
    if (P == 1) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        fb[1][j][0] = *reinterpret_cast<const u32x4*>(buf + (b_addr[j] ^ 32));
        fb[1][j][1] = *reinterpret_cast<const u32x4*>(buf + (b_addr[j] ^ 32 ^ SUB));
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      fa[t][0] = *reinterpret_cast<const u32x4*>(buf + (a_addr[2 * ih + t] ^ (kh * 32)));
      fa[t][1] = *reinterpret_cast<const u32x4*>(buf + (a_addr[2 * ih + t] ^ (kh * 32) ^ SUB));
    }
    if (P == 4 && (!TAIL || stage + 1 < ns)) {
      const unsigned char* nbuf = smem + ((stage + 1) & 1) * STAGE_BYTES;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        fb[0][j][0] = *reinterpret_cast<const u32x4*>(nbuf + b_addr[j]);
        fb[0][j][1] = *reinterpret_cast<const u32x4*>(nbuf + (b_addr[j] ^ SUB));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const int g = stage * 4 + P;  // global phase number; its MFMA slot issues piece g + LOOK
    // What phase g + 1 reads must have landed for every wave before this phase's barrier.  `need` = youngest such piece;
    // pieces need + 1 .. g + LOOK - 1 (issued so far) may stay in flight.
    if (P >= 2) {
      // P == 2: A1 of this stage (piece g + 1); P == 3: B0, B1 of the next stage (up to piece g + 2); P == 4: its A0 (g + 2)
      constexpr int younger = P == 2 ? LOOK - 2 : LOOK - 3;
      const int need = P == 2 ? g + 1 : g + 2;
      if (!TAIL) {
        wait_vm_pieces(younger);  // a compile-time constant here: one s_waitcnt
      } else {
        const int have = npieces - 1 - need;
        wait_vm_pieces(have < younger ? (have < 0 ? 0 : have) : younger);
      }
    }
    constexpr int q = (P + LOOK) & 3;              // the piece this phase issues: which of its stage
    const int pstage = stage + ((P + LOOK) >> 2);  // and which stage
    const bool do_issue = !NO_DMA && (!TAIL || pstage * 4 + q < npieces);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments in registers; this phase's LDS reads are retired
    raw_barrier();
    // ---- MFMA slot; the phase's two LDS-DMA instructions go out after the 2nd and the 3rd MFMA
    __builtin_amdgcn_s_setprio(1);
#define SL_G8_DMA(gi)                                   \
  do {                                                  \
    __builtin_amdgcn_sched_barrier(0);                  \
    if (do_issue) issue(pstage, IntC<q>(), gi);         \
    __builtin_amdgcn_sched_barrier(0);                  \
  } while (0)
    if constexpr (MODE == MODE_BF16X3) {
      // per accumulator: lo*hi, hi*lo, hi*hi (small terms first), as in gemm_bf16x3.hpp; the A operand changes four times
#define SL_G8_MFMA(a, b, t, j)                                                                                          \
  acc[2 * ih + t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), \
                                                               acc[2 * ih + t][j], 0, 0, 0)
      SL_G8_MFMA(fa[0][1], fb[kh][0][0], 0, 0);
      SL_G8_MFMA(fa[0][1], fb[kh][1][0], 0, 1);
      SL_G8_DMA(0);
      SL_G8_MFMA(fa[1][1], fb[kh][0][0], 1, 0);
      SL_G8_DMA(1);
      SL_G8_MFMA(fa[1][1], fb[kh][1][0], 1, 1);
      SL_G8_MFMA(fa[0][0], fb[kh][0][1], 0, 0);
      SL_G8_MFMA(fa[0][0], fb[kh][1][1], 0, 1);
      SL_G8_MFMA(fa[0][0], fb[kh][0][0], 0, 0);
      SL_G8_MFMA(fa[0][0], fb[kh][1][0], 0, 1);
      SL_G8_MFMA(fa[1][0], fb[kh][0][1], 1, 0);
      SL_G8_MFMA(fa[1][0], fb[kh][1][1], 1, 1);
      SL_G8_MFMA(fa[1][0], fb[kh][0][0], 1, 0);
      SL_G8_MFMA(fa[1][0], fb[kh][1][0], 1, 1);
#undef SL_G8_MFMA
    } else {
      // per accumulator: u ascending, then the four k of the float4, as in gemm_f32.hpp
#pragma unroll
      for (int uu = 0; uu < 2; ++uu) {
        float av[2][4], bv[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f32x4 f = __builtin_bit_cast(f32x4, fa[t][uu]);
          av[t][0] = f[0], av[t][1] = f[1], av[t][2] = f[2], av[t][3] = f[3];
          const f32x4 h = __builtin_bit_cast(f32x4, fb[kh][t][uu]);
          bv[t][0] = h[0], bv[t][1] = h[1], bv[t][2] = h[2], bv[t][3] = h[3];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[2 * ih][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][e], bv[0][e], acc[2 * ih][0], 0, 0, 0);
          acc[2 * ih][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][e], bv[1][e], acc[2 * ih][1], 0, 0, 0);
          if (uu == 0 && e == 0) SL_G8_DMA(0);
          acc[2 * ih + 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][e], bv[0][e], acc[2 * ih + 1][0], 0, 0, 0);
          if (uu == 0 && e == 0) SL_G8_DMA(1);
          acc[2 * ih + 1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][e], bv[1][e], acc[2 * ih + 1][1], 0, 0, 0);
        }
      }
    }
#undef SL_G8_DMA
    __builtin_amdgcn_s_setprio(0);
    raw_barrier();
  };
  auto run_stage = [&](int stage, auto Tc) __attribute__((always_inline)) {
    phase(stage, IntC<1>(), Tc);
    phase(stage, IntC<2>(), Tc);
    phase(stage, IntC<3>(), Tc);
    phase(stage, IntC<4>(), Tc);
  };

  // One tile per workgroup.  (A persistent variant — min(tiles, CUs) workgroups looping over tiles, the next tile's
  // prologue DMA issued ahead of the epilogue stores — was built and measured SLOWER, 430 vs 442 TFLOP/s: stores and loads
  // share vmcnt and complete out of order with each other, so the only safe wait for that DMA is vmcnt(0), which also
  // waits for the acknowledgement of all 128 stores per lane, ~10 K cycles a dying workgroup never pays.)
  if (ns > 0) {
    // prologue ("phase 0"): pieces 0..LOOK, then B0, B1, A0 of stage 0 must have landed
#pragma unroll
    for (int g = 0; g < 2; ++g) issue(0, IntC<0>(), g);
#pragma unroll
    for (int g = 0; g < 2; ++g) issue(0, IntC<1>(), g);
#pragma unroll
    for (int g = 0; g < 2; ++g) issue(0, IntC<2>(), g);
#pragma unroll
    for (int g = 0; g < 2; ++g) issue(0, IntC<3>(), g);
    if (ns > 1) {
#pragma unroll
      for (int g = 0; g < 2; ++g) issue(1, IntC<0>(), g);
#pragma unroll
      for (int g = 0; g < 2; ++g) issue(1, IntC<1>(), g);
#pragma unroll
      for (int g = 0; g < 2; ++g) issue(1, IntC<2>(), g);
    }
    wait_vm_pieces(ns > 1 ? LOOK - 2 : 1);
    raw_barrier();
#ifdef SL_GEMM_CLOCKPROBE
    probe_c1 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int j = 0; j < 2; ++j) {  // B fragments of k-half 0 of stage 0 (later stages: phase 4 of the stage before)
      fb[0][j][0] = *reinterpret_cast<const u32x4*>(smem + b_addr[j]);
      fb[0][j][1] = *reinterpret_cast<const u32x4*>(smem + (b_addr[j] ^ SUB));
    }
    if (wr == 1) raw_barrier();  // group 1 runs one barrier behind
    int s = 0;
    for (; s + 2 <= ns - 2; s += 2) {  // every piece these stages issue exists: 4 s + 4 + LOOK < 4 ns
      run_stage(s, IntC<0>());
      run_stage(s + 1, IntC<0>());
    }
    for (; s < ns; ++s) run_stage(s, IntC<1>());
    if (wr == 0) raw_barrier();
  }
#ifdef SL_GEMM_CLOCKPROBE
  probe_c2 = __builtin_amdgcn_s_memtime();
#endif

  // C/D layout of the 32 x 32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  // (gemm_epilogue.hpp: epilogues that read memory per element fetch the 16 values of an MFMA tile ahead of its stores)
  if (m0 + BM <= M && n0 + BN <= N) {  // interior tile: no per-element predicates
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        store_mfma_tile<false>(epi, m0 + wr * 128 + i * 32 + 4 * lh, n0 + wc * 64 + j * 32 + li, acc[i][j], M, N);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        store_mfma_tile<true>(epi, m0 + wr * 128 + i * 32 + 4 * lh, n0 + wc * 64 + j * 32 + li, acc[i][j], M, N);
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

inline int64_t tiles_of(int64_t M, int64_t N) { return ((M + BM - 1) / BM) * ((N + BN - 1) / BN); }
// lane offsets are 32-bit
inline bool fits(int64_t M, int64_t N, int64_t row_bytes) { return (M > N ? M : N) * row_bytes < (1ll << 32); }

// rows of `row_bytes` bytes (16-byte aligned, as the bases), ns = k-tiles of 32 (128-byte lines) per row
template <int MODE, class Epi>
int launch(ProfScope& prof, const void* A, int64_t M, const void* B, int64_t N, int64_t row_bytes, int64_t ns, const Epi& epi,
           hipStream_t st) {
  const int64_t tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
  SL_REQUIRE(tm * tn < (1ll << 31) && ns < (1ll << 29), "GEMM: too many tiles");
  SL_REQUIRE(fits(M, N, row_bytes), "GEMM: operand larger than 4 GB (use another kernel)");
  if (tm * tn == 0) return 0;
  SL_LAUNCH(prof, (gemm_nt_8phase_kernel<MODE, Epi>), dim3((unsigned)(tm * tn)), dim3(512), 0, st, (const unsigned char*)A,
            (const unsigned char*)B, M, N, row_bytes, (int)ns, (int)tm, (int)tn, epi);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

// kernel choice shared by the bf16x3 and the f32 launchers: the 8-phase kernel from half a tile per CU upwards (the encoder's
// 150-600-tile GEMMs gain, ViT-B/32 image encode 9.66 -> 9.08 ms)
inline bool worth_it(int64_t M, int64_t N) { return tiles_of(M, N) * 2 >= (int64_t)num_cus(); }

// fp32-input MFMA mode: a tile takes 5.3x longer than in bf16x3 mode and the 128 x 128 kernel reaches 0.76-0.83 of peak on
// its own, so the big kernel only pays when its last round is nearly full or there is a single round
// (tools/native/gemm3_lab, K = 1152: 150 tiles 84 vs 79 TFLOP/s, 600 tiles 112 vs 120, 1280 tiles 140 vs 126, 1440 tiles
// 132 vs 123).
inline bool worth_it_f32(int64_t M, int64_t N) {
  if (!worth_it(M, N)) return false;
  const int64_t t = tiles_of(M, N), cus = num_cus();
  const int64_t rounds = (t + cus - 1) / cus;
  return rounds == 1 || t * 10 >= rounds * cus * 9;
}

}  // namespace gemm8
}  // namespace sl
