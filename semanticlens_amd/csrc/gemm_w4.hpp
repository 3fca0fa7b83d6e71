// (32 TM) x 256 split-bf16 x3 NT GEMM, FOUR waves per workgroup — one per SIMD, software-pipelined (round 3).
//
// A first, eight-wave version of this tile (two wave groups one barrier apart; removed) showed what a 160-row tile is worth
// (one round instead of two on the 150-tile GEMMs of the encoder) and what that schedule costs: four workgroup barriers per
// 30-MFMA stage and every A fragment read by all eight waves.  Here each SIMD holds ONE wave that owns (32 TM) rows x 64
// columns (2 TM accumulator tiles: 160 AGPRs + 156 VGPRs of its 512 registers at TM = 5) and overlaps its own LDS reads and
// LDS-DMA issues with its own MFMAs:
// * a stage (one 32-wide k-tile: A 32 TM lines + B 256 lines of 128 bytes) is two halves of 6 TM MFMAs (one 16-wide k-step:
//   lo*hi, hi*lo, hi*hi per accumulator).  While the matrix pipe runs half h from fragment set h, the wave reads the
//   2 TM + 4 fragments of the NEXT half into the other set — one `ds_read_b128` behind each of the first MFMAs — and
//   issues its share of the LDS-DMA feed behind the later ones.  LDS reads per stage: 4 waves x 2 x (2 TM + 4) KB = 112 KB
//   at TM = 5 (the eight-wave kernel: 192 KB).
// * ONE workgroup barrier per stage, between the halves of stage s: before it a wave waits for its own LDS-DMA of stage
//   s + 1 (`vmcnt`) and for its last reads of slot s (`lgkmcnt`); after it stage s + 1 may be read by anyone and slot s may be
//   overwritten.  THREE slots: stage s + 3 is requested during the second half of stage s and the first half of stage
//   s + 1 and has until the barrier of stage s + 2 to land — two stage times (~2 us) of lead.
// Same operands, LDS image, swizzle and per-element accumulation order as every kernel of gemm_bf16x3.hpp: bit-identical
// results (tests/test_gpu_parity.py::test_gemm_tile_variants_are_bit_identical).
#pragma once
#include "common.hpp"
#include "gemm_epilogue.hpp"

namespace sl {
namespace gemmw4 {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BN = 256;
constexpr int NSLOT = 3;
#ifndef SL_GW4_RD
#define SL_GW4_RD 0  // lab: 1 = all reads of a half in one burst behind its first MFMA, 2 = in pairs behind every second
#endif
#ifndef SL_GW4_EXP
#define SL_GW4_EXP 0  // lab only (garbage results): 1 = no LDS-DMA in the k loop, 2 = no fragment reads, 3 = neither, 4 = no epilogue
#endif

template <int N_>
struct IntC {
  static constexpr int value = N_;
};
template <int I, int N_, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N_) {
    f(IntC<I>());
    static_for<I + 1, N_>(f);
  }
}

template <int TM>
struct Cfg {
  static constexpr int BM = 32 * TM;
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int NI = (BM + BN) / 32;  // LDS-DMA instructions (8 lines each) per wave and stage
  static constexpr int NI0 = NI - NI / 2;    // of these, issued in a first half (the rest in the second half before it)
  static constexpr int NM = 6 * TM;          // MFMAs per half
  static constexpr int NR = 2 * TM + 4;      // fragment reads per half
  static constexpr int SMEM_BYTES = NSLOT * STAGE_BYTES;
  static_assert(SMEM_BYTES <= 160 * 1024, "three stage slots must fit the LDS");
  static_assert(NR + NI0 <= NM, "one read or LDS-DMA issue behind each MFMA");
};

template <int TM, class Epi>
__global__ __launch_bounds__(256, 1) void gemm3_nt_w4_kernel(const unsigned char* __restrict__ A, const unsigned char* __restrict__ B,
                                                             int64_t M, int64_t N, int64_t row_bytes, int ns, int tiles_m,
                                                             int tiles_n, Epi epi) {
  typedef Cfg<TM> C;
  __shared__ __align__(1024) unsigned char smem[C::SMEM_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..3: columns 64 w ..
  const int li = lane & 31, lh = lane >> 5;
  int tm_i, tn_i;
  {  // XCD-aware tile order, as in gemm_8phase.hpp
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
  const int64_t m0 = (int64_t)tm_i * C::BM;
  const int64_t n0 = (int64_t)tn_i * BN;

  floatx16 acc[TM][2];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][j][e] = 0.f;

  // ---- LDS-DMA plan: row group g = w + 4 i of the stage: groups 0 .. 4 TM - 1 are A rows 8 g .., the rest B rows
  uint32_t src[C::NI];  // byte offset of this lane's 16 bytes in k-tile 0 (operands < 4 GB)
#pragma unroll
  for (int i = 0; i < C::NI; ++i) {
    const int g = w + 4 * i;
    const bool isa = g < 4 * TM;
    const int row0 = isa ? g * 8 : (g - 4 * TM) * 8;
    const int row = row0 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    if (isa) src[i] = (uint32_t)((m0 + row < M ? m0 + row : M - 1) * row_bytes + chunk * 16);
    else src[i] = (uint32_t)((n0 + row < N ? n0 + row : N - 1) * row_bytes + chunk * 16);
  }
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  auto issue = [&](int stage, int slot_off, auto Ic) __attribute__((always_inline)) {
    constexpr int i = decltype(Ic)::value;
    const bool isa = w + 4 * i < 4 * TM;  // wave-uniform
    const unsigned char* base = (isa ? A : B) + (int64_t)stage * 128;
    unsigned char* l = smem + slot_off + (w + 4 * i) * 1024;  // A rows then B rows: row group g sits at g KiB of the slot
    __builtin_amdgcn_global_load_lds((glb_void*)(base + src[i]), (lds_void*)l, 16, 0, 0);
  };

  // ---- fragment addresses inside a stage slot (k-half 0, hi halves; the others are XORs: k-half -> ^32, lo -> ^64)
  int a_addr, b_addr;  // tile t / column tile j: + 4096 t / + 4096 j (32 lines; the swizzle repeats every 16 lines)
  a_addr = li * 128 + ((lh ^ ((li >> 1) & 7)) << 4);
  b_addr = C::A_BYTES + (w * 64 + li) * 128 + ((lh ^ ((li >> 1) & 7)) << 4);
  u32x4 fa[2][TM][2], fb[2][2][2];  // [set][tile][hi, lo]

  // read r of the NR fragment reads of k-half `kh` of the slot at `buf` into set `set`
  auto read_frag = [&](const unsigned char* buf, auto Setc, auto Khc, auto Rc) __attribute__((always_inline)) {
    constexpr int set = decltype(Setc)::value, kh = decltype(Khc)::value, r = decltype(Rc)::value;
    if constexpr (r < 2 * TM) {
      constexpr int t = r >> 1, lo = r & 1;
      fa[set][t][lo] = *reinterpret_cast<const u32x4*>(buf + ((a_addr ^ (kh * 32) ^ (lo * 64)) + t * 4096));
    } else {
      constexpr int j = (r - 2 * TM) >> 1, lo = r & 1;
      fb[set][j][lo] = *reinterpret_cast<const u32x4*>(buf + ((b_addr ^ (kh * 32) ^ (lo * 64)) + j * 4096));
    }
  };
  auto raw_barrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // One half: 6 TM MFMAs of k-half KH of the stage in slot `cur` from fragment set KH; behind MFMA i < NR, read i of the next
  // half (k-half 1 of the same slot, or k-half 0 of the slot at `rd_off`); behind the later ones this half's LDS-DMA issues:
  // KH = 0: instructions NI/2 .. NI - 1 of stage `fetch_stage` = s + 2, KH = 1: instructions 0 .. NI/2 - 1 of stage s + 3.
  auto half = [&](int rd_off, int fetch_stage, int fetch_off, bool do_fetch, auto Khc) __attribute__((always_inline)) {
    constexpr int KH = decltype(Khc)::value;
    constexpr int ND = KH == 0 ? C::NI0 : C::NI / 2;
    constexpr int D0 = KH == 0 ? C::NI / 2 : 0;
    const unsigned char* rbuf = smem + rd_off;
    static_for<0, C::NM>([&](auto Ic) __attribute__((always_inline)) {
      constexpr int i = decltype(Ic)::value;
      // per accumulator: lo*hi, hi*lo, hi*hi (small terms first), as in every kernel of gemm_bf16x3.hpp
      constexpr int p = i / (2 * TM), t = (i % (2 * TM)) >> 1, j = i & 1;
      constexpr int alo = p == 0 ? 1 : 0, blo = p == 1 ? 1 : 0;
      acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[KH][t][alo]),
                                                          __builtin_bit_cast(bf16x8, fb[KH][j][blo]), acc[t][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#if SL_GW4_RD == 1
      if constexpr (i == 0) {
        if (!(SL_GW4_EXP & 2))
          static_for<0, C::NR>([&](auto Rc) __attribute__((always_inline)) { read_frag(rbuf, IntC<1 - KH>(), IntC<1 - KH>(), Rc); });
      }
      if constexpr (i < C::NR) {
      } else {
#elif SL_GW4_RD == 2
      if constexpr (i < C::NR) {
        if constexpr (i % 2 == 0 && i / 2 < C::NR)
          if (!(SL_GW4_EXP & 2)) {
            read_frag(rbuf, IntC<1 - KH>(), IntC<1 - KH>(), IntC<i>());
            read_frag(rbuf, IntC<1 - KH>(), IntC<1 - KH>(), IntC<i + 1>());
          }
      } else {
#else
      if constexpr (i < C::NR) {
        if (!(SL_GW4_EXP & 2)) read_frag(rbuf, IntC<1 - KH>(), IntC<1 - KH>(), Ic);
      } else {
#endif
        // LDS-DMA issue d of this half sits behind MFMA NR + d (NM - NR) / ND
        constexpr int span = C::NM - C::NR;
        static_for<0, ND>([&](auto Dc) __attribute__((always_inline)) {
          constexpr int d = decltype(Dc)::value;
          if constexpr (i == C::NR + d * span / ND) {
            if (do_fetch && !(SL_GW4_EXP & 1)) issue(fetch_stage, fetch_off, IntC<D0 + d>());
          }
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  if (ns > 0) {
    // prologue: stages 0 and 1 and the second-half share (instructions 0 .. NI/2 - 1) of stage 2 requested; stage 0 must
    // have landed for every wave before anyone reads it
    static_for<0, C::NI>([&](auto Ic) __attribute__((always_inline)) { issue(0, 0, Ic); });
    int outstanding_after0 = 0;
    if (ns > 1) {
      static_for<0, C::NI>([&](auto Ic) __attribute__((always_inline)) { issue(1, C::STAGE_BYTES, Ic); });
      outstanding_after0 = C::NI;
    }
    if (ns > 2) {
      static_for<0, C::NI / 2>([&](auto Ic) __attribute__((always_inline)) { issue(2, 2 * C::STAGE_BYTES, Ic); });
      outstanding_after0 = C::NI + C::NI / 2;
    }
    if (outstanding_after0 == C::NI + C::NI / 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NI + C::NI / 2) : "memory");
    else if (outstanding_after0 == C::NI) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();
    static_for<0, C::NR>([&](auto Rc) __attribute__((always_inline)) { read_frag(smem, IntC<0>(), IntC<0>(), Rc); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    // slot offsets of stages s, s + 1, s + 2 (advanced modulo three slots); stage s + 3 goes where stage s is
    int cur = 0, nx1 = C::STAGE_BYTES, nx2 = 2 * C::STAGE_BYTES;
    auto stage_body = [&](int s, auto Tc) __attribute__((always_inline)) {
      constexpr bool TAIL = decltype(Tc)::value != 0;
      // first half: reads k-half 1 of this slot; requests the rest of stage s + 2
      half(cur, s + 2, nx2, !TAIL || s + 2 < ns, IntC<0>());
      // stage s + 1 (this wave's share) must have landed before the barrier lets anyone read it; stage s + 2 may stay in flight
      if (!TAIL || s + 2 < ns) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NI) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      raw_barrier();
      // second half: reads k-half 0 of stage s + 1; requests the first share of stage s + 3 into the slot just released
      half(nx1, s + 3, cur, !TAIL || s + 3 < ns, IntC<1>());
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      const int c = cur;
      cur = nx1;
      nx1 = nx2;
      nx2 = c;
    };
    int s = 0;
    for (; s + 3 < ns; ++s) stage_body(s, IntC<0>());  // everything requested exists: constant waits
    for (; s < ns; ++s) stage_body(s, IntC<1>());
  }

  if (SL_GW4_EXP & 4) {  // lab: no epilogue (one impossible store keeps the accumulators alive)
    float t = 0.f;
#pragma unroll
    for (int tt = 0; tt < TM; ++tt)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) t += acc[tt][j][e];
    if (t == 12345.678f) epi.store(m0, n0, t, epi.column(n0));
    return;
  }
  if (m0 + C::BM <= M && n0 + BN <= N) {
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        store_mfma_tile<false>(epi, m0 + t * 32 + 4 * lh, n0 + w * 64 + j * 32 + li, acc[t][j], M, N);
  } else {
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        store_mfma_tile<true>(epi, m0 + t * 32 + 4 * lh, n0 + w * 64 + j * 32 + li, acc[t][j], M, N);
  }
}

template <int TM, class Epi>
int launch(ProfScope& prof, const void* A, int64_t M, const void* B, int64_t N, int64_t row_bytes, int64_t ns, const Epi& epi,
           hipStream_t st) {
  typedef Cfg<TM> C;
  const int64_t tm = (M + C::BM - 1) / C::BM, tn = (N + BN - 1) / BN;
  SL_REQUIRE(tm * tn < (1ll << 31) && ns < (1ll << 29), "GEMM: too many tiles");
  SL_REQUIRE((M > N ? M : N) * row_bytes < (1ll << 32), "GEMM: operand larger than 4 GB (use another kernel)");
  if (tm * tn == 0) return 0;
  SL_LAUNCH(prof, (gemm3_nt_w4_kernel<TM, Epi>), dim3((unsigned)(tm * tn)), dim3(256), 0, st, (const unsigned char*)A,
            (const unsigned char*)B, M, N, row_bytes, (int)ns, (int)tm, (int)tn, epi);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

// 160-row tiles when they shorten the makespan: rounds x tile time.  A 160 x 256 tile does 0.625 of the work of a 256 x 256
// one and measures 0.80 (K = 768) to 0.87 (K = 3072) of its time (tools/enc_gemm_lab.py, bare epilogue: o-proj 51 -> 41 us,
// fc2 147 -> 128 us), so it is chosen only where a whole round is saved.
inline bool prefer(int64_t M, int64_t N) {
  const int64_t cus = num_cus();
  const int64_t t256 = ((M + 255) / 256) * ((N + 255) / 256), t160 = ((M + 159) / 160) * ((N + 255) / 256);
  if (t256 * 2 < cus) return false;  // small grids stay on the 128 x 128 kernel's side of the choice
  const double c256 = (double)((t256 + cus - 1) / cus);
  const double c160 = (double)((t160 + cus - 1) / cus) * 0.88;
  return c160 < c256 * 0.97;
}

}  // namespace gemmw4
}  // namespace sl
