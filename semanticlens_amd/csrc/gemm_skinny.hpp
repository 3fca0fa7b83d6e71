// 64 x 64 (and 128 x 128) split-bf16 x3 NT GEMM for SMALL AND MID-SIZE GRIDS (round 3): few output tiles, long k loops.
//
// A GEMM with a handful of rows — one image (50 tokens), the class-token rows of the last transformer block (M = batch),
// interactive probing — is a few tiles of any size, and each workgroup walks the whole K alone.  What matters then is the time
// per k-step and how many workgroups pull the weights in parallel, not tile efficiency: the 128 x 128 register-staged kernel
// keeps ONE k-tile in flight and pays a memory round trip per step (0.9 us: fc2 of ViT-B/32 at M = 256, K = 3072, 89 us on 12
// workgroups; the image tower at B = 1: 2.7 ms against 2.0 ms for torch / hipBLASLt).  Here:
// * 64 x 64 tiles (4x the workgroups), four waves as 2 x 2, one 32 x 32 accumulator tile each;
// * an EIGHT-slot LDS ring of whole stages (one 32-wide k-tile: 64 A lines + 64 B lines of 128 bytes = 16 KB, four LDS-DMA
//   instructions per wave): seven stages (~112 KB) in flight per workgroup, a step costs its six dependent MFMAs (~0.15 us);
// * one workgroup barrier per stage; stage s + 7 is requested right behind the barrier of stage s into the slot stage s - 1
//   left; a wave waits for its own share of stage s with a counted `vmcnt(24)`.
// * the same schedule with 128 x 128 tiles (each wave 64 x 64, four 32-KB slots) serves the grids between this regime and
//   the 256 x 256 kernels' (where the register-staged 128 x 128 kernel of round 1 used to run).
// Same operands, LDS image, swizzle, fragment layout and per-element accumulation order (k ascending; lo*hi, hi*lo, hi*hi per
// 16-wide k-step) as every kernel of gemm_bf16x3.hpp: bit-identical results, so an embedding does not depend on the batch it
// was computed in (tests/test_gpu_parity.py::test_gemm_tile_variants_are_bit_identical, test_gpu_native_clip.py).
#pragma once
#include "common.hpp"
#include "gemm_epilogue.hpp"

namespace sl {
namespace gemmsk {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// WT = accumulator tiles per wave and side: 1 -> 64 x 64 tiles, eight 16-KB slots; 2 -> 128 x 128 tiles (each wave 64 x 64),
// four 32-KB slots, for the mid-size grids above (the same ring in front of four times the MFMAs per step)
template <int WT>
struct Cfg {
  static constexpr int BM = 64 * WT, BN = 64 * WT;
  static constexpr int A_BYTES = BM * 128;
  static constexpr int STAGE_BYTES = (BM + BN) * 128;  // 16 / 32 KB
  static constexpr int NSLOT = 8 / WT;
  static constexpr int NI = (BM + BN) / 32;            // LDS-DMA instructions (8 lines each) per wave and stage: 4 / 8
  static constexpr int LEAD = NSLOT - 1;               // stages requested ahead of the one being computed
};

template <int WT, class Epi>
__global__ __launch_bounds__(256, 1) void gemm3_nt_skinny_kernel(const unsigned char* __restrict__ A, const unsigned char* __restrict__ B,
                                                                 int64_t M, int64_t N, int64_t row_bytes, int ns, int tiles_n,
                                                                 Epi epi) {
  typedef Cfg<WT> C;
  constexpr int BM = C::BM, BN = C::BN, A_BYTES = C::A_BYTES, STAGE_BYTES = C::STAGE_BYTES, NSLOT = C::NSLOT, NI = C::NI, LEAD = C::LEAD;
  __shared__ __align__(1024) unsigned char smem[NSLOT * STAGE_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * BM;
  const int64_t n0 = (int64_t)(blockIdx.x % tiles_n) * BN;

  floatx16 acc[WT][WT];
#pragma unroll
  for (int t = 0; t < WT; ++t)
#pragma unroll
    for (int j = 0; j < WT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][j][e] = 0.f;

  // LDS-DMA plan: row group g = w + 4 i of the stage: the first BM / 8 groups are A rows 8 g .., the rest B rows
  uint32_t src[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int g = w + 4 * i;
    const bool isa = g < BM / 8;
    const int row0 = isa ? g * 8 : (g - BM / 8) * 8;
    const int row = row0 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    if (isa) src[i] = (uint32_t)((m0 + row < M ? m0 + row : M - 1) * row_bytes + chunk * 16);
    else src[i] = (uint32_t)((n0 + row < N ? n0 + row : N - 1) * row_bytes + chunk * 16);
  }
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  auto issue = [&](int stage) __attribute__((always_inline)) {
    unsigned char* slot = smem + (stage & (NSLOT - 1)) * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const bool isa = w + 4 * i < BM / 8;  // wave-uniform
      const unsigned char* base = (isa ? A : B) + (int64_t)stage * 128;
      __builtin_amdgcn_global_load_lds((glb_void*)(base + src[i]), (lds_void*)(slot + (w + 4 * i) * 1024), 16, 0, 0);
    }
  };
  // fragment addresses inside a stage slot (k-half 0, hi halves; k-half -> ^32, lo -> ^64)
  // (accumulator tile t / j of the wave: + 4096 t / + 4096 j — 32 lines; the swizzle repeats every 16 lines)
  const int a_addr = (wm * 32 * WT + li) * 128 + ((lh ^ ((li >> 1) & 7)) << 4);
  const int b_addr = A_BYTES + (wn * 32 * WT + li) * 128 + ((lh ^ ((li >> 1) & 7)) << 4);

  auto wait_landed = [&](int ahead) __attribute__((always_inline)) {  // at most `ahead` younger stages still in flight
    switch (ahead) {
#define SL_SK_CASE(n) case n: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n * NI) : "memory"); break;
      SL_SK_CASE(0) SL_SK_CASE(1) SL_SK_CASE(2) SL_SK_CASE(3) SL_SK_CASE(4) SL_SK_CASE(5)
#undef SL_SK_CASE
      default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LEAD - 1) * NI) : "memory"); break;
    }
  };
  auto step = [&](int s, bool fetch) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();  // stage s is visible to everyone; everyone is done with the slot of stage s - 1
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* buf = smem + (s & (NSLOT - 1)) * STAGE_BYTES;
    u32x4 fa[2][WT][2], fb[2][WT][2];  // [k-half][tile][hi, lo]
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int t = 0; t < WT; ++t)
#pragma unroll
        for (int lo = 0; lo < 2; ++lo) {
          fa[kh][t][lo] = *reinterpret_cast<const u32x4*>(buf + ((a_addr ^ (kh * 32) ^ (lo * 64)) + t * 4096));
          fb[kh][t][lo] = *reinterpret_cast<const u32x4*>(buf + ((b_addr ^ (kh * 32) ^ (lo * 64)) + t * 4096));
        }
    __builtin_amdgcn_sched_barrier(0);
    if (fetch) issue(s + LEAD);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)  // per accumulator: lo*hi, hi*lo, hi*hi (small terms first), as in every other kernel
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int t = 0; t < WT; ++t)
#pragma unroll
          for (int j = 0; j < WT; ++j)
            acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[kh][t][p == 0 ? 1 : 0]),
                                                                __builtin_bit_cast(bf16x8, fb[kh][j][p == 1 ? 1 : 0]), acc[t][j], 0, 0, 0);
    // the reads of this slot have returned (the MFMAs consumed them) before this wave reaches the next barrier
  };

  if (ns > 0) {
    const int pre = ns < LEAD ? ns : LEAD;
    for (int s = 0; s < pre; ++s) issue(s);
    int s = 0;
    for (; s + LEAD < ns; ++s) {  // steady state: stages s .. s + LEAD - 1 requested, stage s + LEAD goes out in this step
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LEAD - 1) * NI) : "memory");
      step(s, true);
    }
    for (; s < ns; ++s) {  // tail: ns - 1 - s younger stages remain in flight
      wait_landed(ns - 1 - s);
      step(s, false);
    }
  }

  const bool inside = m0 + BM <= M && n0 + BN <= N;
#pragma unroll
  for (int t = 0; t < WT; ++t)
#pragma unroll
    for (int j = 0; j < WT; ++j) {
      const int64_t row0 = m0 + (wm * WT + t) * 32 + 4 * lh, col = n0 + (wn * WT + j) * 32 + li;
      if (inside) store_mfma_tile<false>(epi, row0, col, acc[t][j], M, N);
      else store_mfma_tile<true>(epi, row0, col, acc[t][j], M, N);
    }
}

template <int WT, class Epi>
int launch(ProfScope& prof, const void* A, int64_t M, const void* B, int64_t N, int64_t row_bytes, int64_t ns, const Epi& epi,
           hipStream_t st) {
  constexpr int BM = Cfg<WT>::BM, BN = Cfg<WT>::BN;
  const int64_t tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
  SL_REQUIRE(tm * tn < (1ll << 31) && ns < (1ll << 29), "GEMM: too many tiles");
  SL_REQUIRE((M > N ? M : N) * row_bytes < (1ll << 32), "GEMM: operand larger than 4 GB (use another kernel)");
  if (tm * tn == 0) return 0;
  SL_LAUNCH(prof, (gemm3_nt_skinny_kernel<WT, Epi>), dim3((unsigned)(tm * tn)), dim3(256), 0, st, (const unsigned char*)A,
            (const unsigned char*)B, M, N, row_bytes, (int)ns, (int)tn, epi);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

// Up to two 64 x 64 tiles per CU, k loops long enough for the ring to matter; beyond that the 128 x 128 instance (the caller's
// fallback for mid-size grids) is as fast or faster.  Measured on the ViT-B/32 block shapes (tools/enc_gemm_lab.py <M>,
// SL_G3_TILE = 64 / 1280 / 128 = this kernel / its 128 x 128 instance / the register-staged 128 x 128 kernel), us:
//   M =   256: o-proj  8 /  - / 30, fc2 22 /  - / 90, qkv  9 /  - / 30, fc1  9 /  - / 35
//   M = 1 600: o-proj 19 / 29 / 32, fc2 48 / 68 / 89 (300 tiles); qkv 34 / 29 / 33 (900), fc1 47 / 50 / 53 (1 200)
//   M = 3 200: o-proj 28 / 24 / 36, fc2 74 / 67 / 95 (600 tiles); qkv 67 / 45 / 50, fc1 88 / 69 / 73
inline bool prefer(int64_t M, int64_t N, int64_t ns) {
  const int64_t t64 = ((M + 63) / 64) * ((N + 63) / 64);
  return t64 <= 2 * (int64_t)num_cus() && ns >= 8;
}

}  // namespace gemmsk
}  // namespace sl
