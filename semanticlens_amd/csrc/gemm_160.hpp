// 160 x 256 split-bf16 x3 NT GEMM for grids the 256 x 256 kernel quantises badly (round 3).
//
// The makespan of T equal tiles on 256 CUs is ceil(T / 256) tile times whatever the idle CUs do, so a GEMM of 150 tiles of
// 256 x 256 (the out-projection and fc2 of a ViT-B/32 batch: 12 800 x 768) runs one full tile time on 59 % of the chip, and
// neither split-K (300 / 450 items: two sub-rounds again, plus partial-sum traffic) nor half tiles (300 items) shorten it.
// Only SMALLER equal items do: 160 x 256 tiles are 80 x 3 = 240 items of 0.625 of the work each — one round, 94 % of the CUs.
//
// Same operands, LDS image, swizzle, fragment reads and per-element accumulation order (k ascending; lo*hi, hi*lo, hi*hi per
// 16-wide k-step) as gemm_8phase.hpp, so results are bit-identical to every other kernel of gemm_bf16x3.hpp.  What differs:
// * 8 waves, each owning ALL 160 rows x 32 columns (5 MFMA tiles stacked, 80 accumulator registers): no row split, every wave
//   reads every A fragment (the LDS read volume per stage is that of the 256 x 256 kernel: 192 KB).
// * a stage (one 32-wide k-tile: A 160 lines + B 256 lines = 52 KB) is computed in TWO phases, one per 16-wide k-half:
//   [read 10 A + 2 B fragments] s_barrier [15 MFMAs] s_barrier; waves 0-3 and 4-7 run one barrier apart, so each SIMD has one
//   wave on the matrix pipe while the other reads.
// * THREE stage slots (156 KB of LDS + 1 KB spare): stage s + 2 is fetched (LDS-DMA, 52 row groups of eight 128-byte lines =
//   7 instructions per wave, the 7th of waves 4-7 a dummy into the spare KiB so that every wave counts the same) during the
//   MFMA slots of stage s — four instructions in the first, three in the second — into the slot stage s - 1 left; a wave waits
//   for its own stage-(s + 1) instructions with `vmcnt(4)` in the second phase's read slot, before the barrier that lets
//   anyone read them.  Two stages (~2 us) of lead instead of the 8-phase kernel's six 16-KB pieces.
#pragma once
#include "common.hpp"
#include "gemm_epilogue.hpp"

namespace sl {
namespace gemm160 {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 160, BN = 256;
constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;  // 53 248
constexpr int NSLOT = 3;
constexpr int NI = 7;                           // LDS-DMA instructions per wave and stage (52 row groups + 4 dummies)
constexpr int SMEM_BYTES = NSLOT * STAGE_BYTES + 1024;
#ifndef SL_G160_EXP
#define SL_G160_EXP 0  // lab only (garbage results): 1 = no LDS-DMA in the k loop, 2 = no fragment reads, 3 = neither
#endif

template <int N_>
struct IntC {
  static constexpr int value = N_;
};

template <class Epi>
__global__ __launch_bounds__(512, 2) void gemm3_nt_160_kernel(const unsigned char* __restrict__ A, const unsigned char* __restrict__ B,
                                                              int64_t M, int64_t N, int64_t row_bytes, int ns, int tiles_m,
                                                              int tiles_n, Epi epi) {
  __shared__ __align__(1024) unsigned char smem[SMEM_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7
  const int grp = w >> 2;                                  // wave group: 0 leads, 1 runs one barrier behind
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
  const int64_t m0 = (int64_t)tm_i * BM;
  const int64_t n0 = (int64_t)tn_i * BN;

  floatx16 acc[5];
#pragma unroll
  for (int t = 0; t < 5; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  // ---- LDS-DMA plan: row group g = w + 8 i (i = 0..6) of the stage: groups 0-19 are A rows 8 g .., 20-51 B rows 8 (g - 20) ..
  uint32_t src[NI];  // byte offset of this lane's 16 bytes in k-tile 0 (operands < 4 GB)
  int dst[NI];       // wave-uniform LDS offset of the row group inside a stage slot (or the spare KiB)
  bool from_a[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    int g = w + 8 * i;
    const bool dummy = g >= 52;
    if (dummy) g = w;  // any valid lines; they land in the spare KiB
    const bool isa = g < 20;
    const int row0 = isa ? g * 8 : (g - 20) * 8;
    const int row = row0 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    if (isa) src[i] = (uint32_t)((m0 + row < M ? m0 + row : M - 1) * row_bytes + chunk * 16);
    else src[i] = (uint32_t)((n0 + row < N ? n0 + row : N - 1) * row_bytes + chunk * 16);
    dst[i] = dummy ? -1 : (isa ? row0 * 128 : A_BYTES + row0 * 128);
    from_a[i] = isa;
  }
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  // `slot_off` = byte offset of the stage's slot (the caller keeps it as a counter: no division by three in the loop)
  auto issue = [&](int stage, int slot_off, auto Ic) __attribute__((always_inline)) {
    constexpr int i = decltype(Ic)::value;
    const unsigned char* base = (from_a[i] ? A : B) + (int64_t)stage * 128;
    unsigned char* l = dst[i] < 0 ? smem + NSLOT * STAGE_BYTES : smem + slot_off + dst[i];
    __builtin_amdgcn_global_load_lds((glb_void*)(base + src[i]), (lds_void*)l, 16, 0, 0);
  };

  // ---- fragment addresses inside a stage slot (k-half 0, hi halves; the others are XORs: k-half -> ^32, lo -> ^64)
  int a_addr[5], b_addr;
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const int ar = t * 32 + li;
    a_addr[t] = ar * 128 + ((lh ^ ((ar >> 1) & 7)) << 4);
  }
  {
    const int br = w * 32 + li;
    b_addr = A_BYTES + br * 128 + ((lh ^ ((br >> 1) & 7)) << 4);
  }
  u32x4 fa[5][2], fb[2];

  auto raw_barrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // one phase = one k-half of a stage.  TAIL: one of the last two stages (the stage to fetch may not exist; waits are exact)
  // cur_off / nxt_off: slot offsets of `stage` and of `stage + 2`
  auto phase = [&](int stage, int cur_off, int nxt_off, auto Kc, auto Tc) __attribute__((always_inline)) {
    constexpr int kh = decltype(Kc)::value;
    constexpr bool TAIL = decltype(Tc)::value != 0;
    const unsigned char* buf = smem + cur_off;
    // ---- read slot: 10 A + 2 B fragments
    if (!(SL_G160_EXP & 2) || stage == 0)
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      fa[t][0] = *reinterpret_cast<const u32x4*>(buf + (a_addr[t] ^ (kh * 32)));
      fa[t][1] = *reinterpret_cast<const u32x4*>(buf + (a_addr[t] ^ (kh * 32) ^ 64));
    }
    fb[0] = *reinterpret_cast<const u32x4*>(buf + (b_addr ^ (kh * 32)));
    fb[1] = *reinterpret_cast<const u32x4*>(buf + (b_addr ^ (kh * 32) ^ 64));
    __builtin_amdgcn_sched_barrier(0);
    const bool fetch = stage + 2 < ns;  // this stage's MFMA slots fetch stage + 2
    if (kh == 1 && (!TAIL || stage + 1 < ns)) {
      // stage + 1 (this wave's 7 instructions, issued during stage - 1) must have landed before the barrier below lets any
      // wave read it; the 4 instructions of stage + 2 issued in this stage's first MFMA slot may stay in flight
      if (!TAIL || fetch) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    raw_barrier();
    // ---- MFMA slot: 15 MFMAs; the LDS-DMA instructions of the phase go out between them, issued by the wave that computes
    __builtin_amdgcn_s_setprio(1);
    const bool do_issue = (!TAIL || fetch) && !(SL_G160_EXP & 1);
#define SL_G160_DMA(ii)                                  \
  do {                                                   \
    __builtin_amdgcn_sched_barrier(0);                   \
    if (do_issue) issue(stage + 2, nxt_off, IntC<ii>()); \
    __builtin_amdgcn_sched_barrier(0);                   \
  } while (0)
#define SL_G160_MFMA(a, b, t) \
  acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[t], 0, 0, 0)
    // per accumulator: lo*hi, hi*lo, hi*hi (small terms first), as in every kernel of gemm_bf16x3.hpp
    SL_G160_MFMA(fa[0][1], fb[0], 0);
    SL_G160_MFMA(fa[1][1], fb[0], 1);
    SL_G160_MFMA(fa[2][1], fb[0], 2);
    SL_G160_DMA(kh == 0 ? 0 : 4);
    SL_G160_MFMA(fa[3][1], fb[0], 3);
    SL_G160_MFMA(fa[4][1], fb[0], 4);
    SL_G160_MFMA(fa[0][0], fb[1], 0);
    SL_G160_DMA(kh == 0 ? 1 : 5);
    SL_G160_MFMA(fa[1][0], fb[1], 1);
    SL_G160_MFMA(fa[2][0], fb[1], 2);
    SL_G160_MFMA(fa[3][0], fb[1], 3);
    SL_G160_DMA(kh == 0 ? 2 : 6);
    SL_G160_MFMA(fa[4][0], fb[1], 4);
    SL_G160_MFMA(fa[0][0], fb[0], 0);
    SL_G160_MFMA(fa[1][0], fb[0], 1);
    if constexpr (kh == 0) SL_G160_DMA(3);
    SL_G160_MFMA(fa[2][0], fb[0], 2);
    SL_G160_MFMA(fa[3][0], fb[0], 3);
    SL_G160_MFMA(fa[4][0], fb[0], 4);
#undef SL_G160_MFMA
#undef SL_G160_DMA
    __builtin_amdgcn_s_setprio(0);
    raw_barrier();
  };

  if (ns > 0) {
    // prologue: stages 0 and 1 requested; stage 0 must have landed for every wave before anyone reads it
    issue(0, 0, IntC<0>()); issue(0, 0, IntC<1>()); issue(0, 0, IntC<2>()); issue(0, 0, IntC<3>());
    issue(0, 0, IntC<4>()); issue(0, 0, IntC<5>()); issue(0, 0, IntC<6>());
    if (ns > 1) {
      issue(1, STAGE_BYTES, IntC<0>()); issue(1, STAGE_BYTES, IntC<1>()); issue(1, STAGE_BYTES, IntC<2>());
      issue(1, STAGE_BYTES, IntC<3>()); issue(1, STAGE_BYTES, IntC<4>()); issue(1, STAGE_BYTES, IntC<5>());
      issue(1, STAGE_BYTES, IntC<6>());
      asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    raw_barrier();
    if (grp == 1) raw_barrier();  // group 1 runs one barrier behind
    int s = 0;
    int cur = 0, nxt = 2 * STAGE_BYTES;  // slot offsets of stage s and stage s + 2, advanced modulo three slots
    auto advance = [&]() __attribute__((always_inline)) {
      cur = cur + STAGE_BYTES == NSLOT * STAGE_BYTES ? 0 : cur + STAGE_BYTES;
      nxt = nxt + STAGE_BYTES == NSLOT * STAGE_BYTES ? 0 : nxt + STAGE_BYTES;
    };
    for (; s + 2 < ns; ++s) {  // stage s + 2 exists: constant waits
      phase(s, cur, nxt, IntC<0>(), IntC<0>());
      phase(s, cur, nxt, IntC<1>(), IntC<0>());
      advance();
    }
    for (; s < ns; ++s) {
      phase(s, cur, nxt, IntC<0>(), IntC<1>());
      phase(s, cur, nxt, IntC<1>(), IntC<1>());
      advance();
    }
    if (grp == 0) raw_barrier();
  }

  if (m0 + BM <= M && n0 + BN <= N) {
#pragma unroll
    for (int t = 0; t < 5; ++t) store_mfma_tile<false>(epi, m0 + t * 32 + 4 * lh, n0 + w * 32 + li, acc[t], M, N);
  } else {
#pragma unroll
    for (int t = 0; t < 5; ++t) store_mfma_tile<true>(epi, m0 + t * 32 + 4 * lh, n0 + w * 32 + li, acc[t], M, N);
  }
}

inline int64_t tiles_of(int64_t M, int64_t N) { return ((M + BM - 1) / BM) * ((N + BN - 1) / BN); }

template <class Epi>
int launch(ProfScope& prof, const void* A, int64_t M, const void* B, int64_t N, int64_t row_bytes, int64_t ns, const Epi& epi,
           hipStream_t st) {
  const int64_t tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
  SL_REQUIRE(tm * tn < (1ll << 31) && ns < (1ll << 29), "GEMM: too many tiles");
  SL_REQUIRE((M > N ? M : N) * row_bytes < (1ll << 32), "GEMM: operand larger than 4 GB (use another kernel)");
  if (tm * tn == 0) return 0;
  SL_LAUNCH(prof, (gemm3_nt_160_kernel<Epi>), dim3((unsigned)(tm * tn)), dim3(512), 0, st, (const unsigned char*)A,
            (const unsigned char*)B, M, N, row_bytes, (int)ns, (int)tm, (int)tn, epi);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

// 160-row tiles when they shorten the makespan: rounds x tile time.  A 160 x 256 tile does 0.625 of the work of a 256 x 256
// one but measures 0.85 (K = 768) to 0.92 (K = 3072) of its time (tools/enc_gemm_lab.py: o-proj 60 -> 52 us, fc2 160 -> 147 us):
// with the k loop stripped to MFMAs + barriers the kernel runs 122 us on the fc2 shape, fragment reads add 5, the LDS-DMA
// feed 15-20 (-DSL_G160_EXP=3 / 1 / 2) — the feed lands 52 KB per 30-MFMA stage in an LDS that serves the same 192 KB of
// fragment reads as the big tile's 48-MFMA stage.  So it is chosen only where a whole round is saved.
inline bool prefer(int64_t M, int64_t N) {
  const int64_t cus = num_cus();
  const int64_t t256 = ((M + 255) / 256) * ((N + 255) / 256), t160 = tiles_of(M, N);
  if (t256 * 2 < cus) return false;  // small grids stay on the 128 x 128 kernel's side of the choice
  const double c256 = (double)((t256 + cus - 1) / cus);
  const double c160 = (double)((t160 + cus - 1) / cus) * 0.92;
  return c160 < c256 * 0.97;
}

}  // namespace gemm160
}  // namespace sl
