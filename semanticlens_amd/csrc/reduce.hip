// K1 / K2 — activation reduce kernels (HBM-bound).
//
// Replaces the reference's aggregators (component_visualization/aggregators.py:38-244:
// `tensor.clone().flatten(2).amax(-1)` etc. followed by `.cpu()`) and the bf16 cast of
// ActMax.update (activation_caching.py:133).  One pass over the activation, no clone, no
// host round trip; output is the (B,C) candidate matrix the top-k merge consumes.
//
// Roofline: HBM.  Algorithmic bytes per launch = B*C*S*sizeof(act) read (+ B*C*2 written).
//
// Three code paths, chosen on the host from the strides:
//   rowreduce<G>  — rows contiguous (NCHW): the tensor is a flat stream of R = B*C rows of
//                   S floats.  G lanes own one row and read it as 16-byte pieces from the
//                   16-byte-aligned window that covers it (rows such as 7x7 = 196 B are not
//                   16-B aligned, so head/tail lanes mask by element index).  64/G rows share
//                   one 1-KiB wave-load; reduction across the G lanes is DPP.
//   colreduce     — reduced axis strided, component axis contiguous (tokens (B,T,F), or
//                   channels_last conv): lanes along F with 16-byte loads, the 4 waves of a
//                   workgroup split T and combine through LDS.
//   generic       — any strides / fp16 / bf16: one lane per output element.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.hpp"

namespace sl {
namespace {

enum Op : int { OP_MAX = 0, OP_SUM = 1, OP_ABSMAX = 2, OP_ABSSUM = 3 };

// ---- cache policy of the row-reduce streams (sl_set_reduce_policy; environment SL_NT_MIN_BYTES / SL_REDUCE_TAIL_MB) ----
// A COLD input streams best with the nt (read-once) policy: 6.4 vs 5.9 TB/s on 411 MB.  Inside a model the input was
// written by the previous kernel microseconds ago; what still sits (dirty) in the 256 MiB Infinity Cache reads faster
// with the default policy and nt on it LOSES (in-bench average 5.0 TB/s all-nt vs 5.9 mixed).  The kernel cannot know
// its producer, so the default assumes the common case — a forward hook on the layer that just ran: inputs below
// `nt_min_bytes` (default 256 MiB) are read with the default policy; of larger ones the last `tail_bytes` (default
// 240 MiB: what the cache still holds) likewise and the head with nt.  tail_bytes = 0 and nt_min_bytes = 0 = all nt,
// the right setting for inputs known to be cold.
int64_t g_nt_min_bytes = -1, g_tail_bytes = -1;
int64_t nt_min_bytes_() {
  if (g_nt_min_bytes < 0) {
    const char* e = getenv("SL_NT_MIN_BYTES");
    g_nt_min_bytes = e ? (int64_t)atoll(e) : (int64_t)256 << 20;
  }
  return g_nt_min_bytes;
}
int64_t tail_bytes_() {
  if (g_tail_bytes < 0) {
    const char* e = getenv("SL_REDUCE_TAIL_MB");
    g_tail_bytes = (e ? (int64_t)atoll(e) : (int64_t)240) << 20;
  }
  return g_tail_bytes;
}

// ---- cross-lane helpers ------------------------------------------------------------------
template <int CTRL>
__device__ inline int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
// float -> int whose signed order equals the float order with +NaN on top
__device__ inline int f32_sort_key(float f) {
  int b = (int)f32_bits(f);
  return b ^ ((b >> 31) & 0x7FFFFFFF);
}
__device__ inline float sort_key_f32(int k) { return bits_f32((uint32_t)(k ^ ((k >> 31) & 0x7FFFFFFF))); }

template <bool SUM>
__device__ inline float combine(float a, float b) {
  if constexpr (SUM) return a + b;
  return sort_key_f32(max(f32_sort_key(a), f32_sort_key(b)));  // NaN-propagating max
}

// all-reduce over aligned groups of G lanes (G = 1,2,4,...,64)
template <int G, bool SUM>
__device__ inline float group_allreduce(float v) {
  if constexpr (SUM) {
    if constexpr (G >= 2) v += bits_f32((uint32_t)dpp_i32<0xB1>((int)f32_bits(v)));   // quad_perm [1,0,3,2]
    if constexpr (G >= 4) v += bits_f32((uint32_t)dpp_i32<0x4E>((int)f32_bits(v)));   // quad_perm [2,3,0,1]
    if constexpr (G >= 8) v += bits_f32((uint32_t)dpp_i32<0x141>((int)f32_bits(v)));  // row_half_mirror
    if constexpr (G >= 16) v += bits_f32((uint32_t)dpp_i32<0x140>((int)f32_bits(v))); // row_mirror
    if constexpr (G >= 32) v += __shfl_xor(v, 16, 64);
    if constexpr (G >= 64) v += __shfl_xor(v, 32, 64);
    return v;
  } else {
    int k = f32_sort_key(v);
    if constexpr (G >= 2) k = max(k, dpp_i32<0xB1>(k));
    if constexpr (G >= 4) k = max(k, dpp_i32<0x4E>(k));
    if constexpr (G >= 8) k = max(k, dpp_i32<0x141>(k));
    if constexpr (G >= 16) k = max(k, dpp_i32<0x140>(k));
    if constexpr (G >= 32) k = max(k, __shfl_xor(k, 16, 64));
    if constexpr (G >= 64) k = max(k, __shfl_xor(k, 32, 64));
    return sort_key_f32(k);
  }
}

// per-lane accumulator: running max ignores NaN (v_max_f32) and remembers it separately
template <int OP>
struct Acc {
  float v;
  bool nan;
  __device__ inline void init() {
    v = (OP == OP_SUM || OP == OP_ABSSUM) ? 0.f : -__builtin_huge_valf();
    nan = false;
  }
  __device__ inline void add(float x, bool valid) {
    if constexpr (OP == OP_ABSMAX || OP == OP_ABSSUM) x = __builtin_fabsf(x);
    if constexpr (OP == OP_SUM || OP == OP_ABSSUM) {
      v += valid ? x : 0.f;
    } else {
      x = valid ? x : -__builtin_huge_valf();
      nan |= (x != x);
      v = __builtin_fmaxf(v, x);
    }
  }
  __device__ inline float lane_value() const {
    if constexpr (OP == OP_SUM || OP == OP_ABSSUM) return v;
    return nan ? bits_f32(0x7FC00000u) : v;
  }
};

template <int OP>
__device__ inline float finish(float v, float count) {
  if constexpr (OP == OP_SUM || OP == OP_ABSSUM) return v / count;  // torch: sum / n
  return v;
}

__device__ inline void store_outputs(float r, int64_t idx, uint16_t* cand, float* outf) {
  if (outf) outf[idx] = r;
  if (cand) cand[idx] = f32_to_bf16_rne(r);
}

// ---- rowreduce: contiguous rows -------------------------------------------------------------
// x: 16-byte aligned, R rows of S floats back to back.  A wave works on a batch of U tasks; a task
// is 64/G consecutive rows covered by one 1-KiB wave-load per step (G lanes per row).
//
// Cost model (HBM-bound: ~13 B/clk/CU at 8 TB/s => one 1-KiB wave-load per ~78 clk per CU):
//  * max: v_max_f32 ignores NaN, torch.amax propagates it.  Instead of testing every element, a
//    running SUM rides along (NaN in => NaN out); only when a row's sum is NaN (a NaN, or +inf and
//    -inf together) the row is re-scanned exactly.  4 max + 4 add per 16-byte piece.
//  * element masks for rows that are not 16-byte aligned (e.g. 7x7 = 49 floats) depend only on the
//    lane when 64/G is a multiple of 4, so they are computed once per kernel.
//  * addressing: wave-uniform 64-bit batch base + 32-bit lane offsets.
template <bool SUMOP>
__device__ inline float dpp_combine(float v, float o) {
  if constexpr (SUMOP) return v + o;
  return __builtin_fmaxf(v, o);
}
template <int CTRL, bool SUMOP>
__device__ inline float dpp_step(float v) {
  return dpp_combine<SUMOP>(v, bits_f32((uint32_t)dpp_i32<CTRL>((int)f32_bits(v))));
}
// all-reduce (plain float max / add) over aligned groups of G lanes
template <int G, bool SUMOP>
__device__ inline float group_allreduce_f(float v) {
  if constexpr (G >= 2) v = dpp_step<0xB1, SUMOP>(v);
  if constexpr (G >= 4) v = dpp_step<0x4E, SUMOP>(v);
  if constexpr (G >= 8) v = dpp_step<0x141, SUMOP>(v);
  if constexpr (G >= 16) v = dpp_step<0x140, SUMOP>(v);
  if constexpr (G >= 32) v = dpp_combine<SUMOP>(v, __shfl_xor(v, 16, 64));
  if constexpr (G >= 64) v = dpp_combine<SUMOP>(v, __shfl_xor(v, 32, 64));
  return v;
}

// TAIL: total = R*S is not a multiple of 4, so the tensor ends inside a 16-byte piece; the last
// (total & 3) floats are masked out of the vector loads and added by scalar loads to the rows that
// own them (up to three rows when S < 4).
template <int G, int U, int OP, bool TAIL>
__global__ __launch_bounds__(256) void rowreduce_kernel(const float* __restrict__ x, int64_t R, int S, float denom,
                                                         uint16_t* __restrict__ cand, float* __restrict__ outf) {
  constexpr int RPT = kWave / G;
  constexpr bool SUMOP = (OP == OP_SUM || OP == OP_ABSSUM);
  constexpr bool ABS = (OP == OP_ABSMAX || OP == OP_ABSSUM);
  constexpr bool HOIST_H = (RPT % 4 == 0);  // row phase h = (row * S) & 3 depends on the lane only
  const float fill = SUMOP ? 0.f : -__builtin_huge_valf();
  const int lane = threadIdx.x & 63;
  const int li = lane & (G - 1);
  const int g = lane / G;
  const int64_t total = R * (int64_t)S;
  const int64_t total4 = total & ~3ll;  // floats readable as whole 16-byte pieces
  const int64_t nbatch = (R + U * RPT - 1) / (U * RPT);
  // wave-uniform by construction; readfirstlane lets the compiler keep the batch base in SGPRs
  const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int nsteps = ((S + 6) / 4 + G - 1) / G;  // 16-byte pieces of a row window, per lane
  const int h_lane = (g * S) & 3;

  for (int64_t tb = wave0; tb < nbatch; tb += nwaves) {
    const int64_t row0 = tb * (int64_t)(U * RPT);  // wave-uniform
    const int64_t e_batch = row0 * (int64_t)S;
    const int delta = (int)(e_batch & 3);
    const int64_t a0 = e_batch - delta;
    const float4* __restrict__ A = reinterpret_cast<const float4*>(x + a0);  // wave-uniform, 16-byte aligned
    // last whole piece of the tensor, relative to A: lanes whose piece would start beyond it are
    // clamped onto it; every element they then hold is masked by its row position anyway
    // (a batch that starts at or beyond the last whole piece — always the case for a tensor of fewer than four floats —
    // clamps onto its own first piece: aligned, holds at least one float of the tensor, hence readable; index -1 would
    // be the 16 bytes in front of the tensor)
    const int64_t lim = (total4 - a0) / 4 - 1;
    const int idx_max = lim > 0x7FFFFFFF ? 0x7FFFFFFF : (lim < 0 ? 0 : (int)lim);
    const int64_t rel = total4 - a0;  // floats of whole pieces left from A on
    const int rel_lim = rel > 0x7FFFFFFF ? 0x7FFFFFFF : (int)rel;

    float m[U], sum[U];
    int rs[U];  // row start in elements relative to A
#pragma unroll
    for (int u = 0; u < U; ++u) {
      m[u] = fill;
      sum[u] = 0.f;
      rs[u] = delta + (u * RPT + g) * S;
    }

    for (int step = 0; step < nsteps; ++step) {
      const int q = step * G + li;
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = A[min((rs[u] >> 2) + q, idx_max)];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int h = HOIST_H ? h_lane : (rs[u] & 3);
        const int pos0 = q * 4 - h;  // row-local index of v.x (negative in the head piece)
        int lim_s = S;
        if constexpr (TAIL) {  // also drop elements at or beyond the last whole piece of the tensor
          const int left = rel_lim - rs[u];  // row-local index of the first float not covered by whole pieces
          lim_s = left < S ? (left > 0 ? left : 0) : S;
        }
        float e0 = v[u].x, e1 = v[u].y, e2 = v[u].z, e3 = v[u].w;
        if constexpr (ABS) {
          e0 = __builtin_fabsf(e0); e1 = __builtin_fabsf(e1); e2 = __builtin_fabsf(e2); e3 = __builtin_fabsf(e3);
        }
        e0 = (unsigned)(pos0 + 0) < (unsigned)lim_s ? e0 : fill;
        e1 = (unsigned)(pos0 + 1) < (unsigned)lim_s ? e1 : fill;
        e2 = (unsigned)(pos0 + 2) < (unsigned)lim_s ? e2 : fill;
        e3 = (unsigned)(pos0 + 3) < (unsigned)lim_s ? e3 : fill;
        if constexpr (!SUMOP)
          m[u] = __builtin_fmaxf(__builtin_fmaxf(m[u], __builtin_fmaxf(e0, e1)), __builtin_fmaxf(e2, e3));
        sum[u] += (e0 + e1) + (e2 + e3);
      }
    }

#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = row0 + (u * RPT + g);
      const bool row_ok = row < R;
      if constexpr (TAIL) {
        if (row_ok && li == 0 && (row + 1) * (int64_t)S > total4) {  // this row owns floats behind the last whole piece
          const int64_t lo = row * (int64_t)S > total4 ? row * (int64_t)S : total4;
          for (int64_t i = lo; i < (row + 1) * (int64_t)S; ++i) {
            float e = x[i];
            if constexpr (ABS) e = __builtin_fabsf(e);
            if constexpr (!SUMOP) m[u] = __builtin_fmaxf(m[u], e);
            sum[u] += e;
          }
        }
      }
      float r;
      if constexpr (SUMOP) {
        r = group_allreduce_f<G, true>(sum[u]) / denom;  // torch: sum / n (denom = 1: plain sum)
      } else {
        r = group_allreduce_f<G, false>(m[u]);
        const float sred = group_allreduce_f<G, true>(sum[u]);
        if (__builtin_expect(__any(row_ok && sred != sred), 0)) {
          // exact re-scan of this lane-group's row: does it really hold a NaN?
          bool nan = false;
          if (row_ok && sred != sred) {
            const float* rowp = x + row * (int64_t)S;
            for (int i = li; i < S; i += G) nan |= (rowp[i] != rowp[i]);
          }
          const float f = group_allreduce_f<G, false>(nan ? 1.f : 0.f);
          if (f > 0.f) r = bits_f32(0x7FC00000u);
        }
      }
      if (li == 0 && row_ok) store_outputs(r, row, cand, outf);
    }
  }
}

// ---- rowreduce_fast: the streaming path for the common shapes -----------------------------------
// Preconditions (checked by the launcher): the row phase h = (row*S)&3 is the same for every row a
// lane ever touches, i.e. S % 4 == 0 (ALIGNED: h = 0, a row is a whole number of 16-byte pieces) or
// 64/G % 4 == 0 with one piece per lane (h = (g*S)&3).  Then
//   * the lane's byte offset inside a task and its four element masks are loop invariant,
//   * the task base is wave-uniform, so every load is `global_load_dwordx4 v, v_off, s[base]`,
//   * ALIGNED rows need no element masks at all: lanes past the row's last piece re-read that piece
//     (max is idempotent; for sums the piece is masked as a whole).
// VALU per 16-byte piece: 2 v_max3 + 4 v_add (+ 4 v_cndmask when rows are unaligned); per row one DPP
// max-reduction.  hipcc's own fmaxf lowering (canonicalising v_max pairs, unfused DPP moves) cost ~3x
// that and capped the kernel near 3.6 TB/s, hence the few single-instruction asm helpers below.
__device__ inline float v_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ inline float v_max2(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// r = op(a, dpp(a)); the s_nop covers the VALU-write -> DPP-read hazard (2 wait states), which the
// compiler does not pad inside an asm statement.
#define SL_DPP_OP(name, insn, ctrl)                                                           \
  __device__ inline float name(float a) {                                                     \
    float r;                                                                                  \
    asm("s_nop 1\n\t" insn " %0, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(a)); \
    return r;                                                                                 \
  }
SL_DPP_OP(max_qp1, "v_max_f32_dpp", "quad_perm:[1,0,3,2]")
SL_DPP_OP(max_qp2, "v_max_f32_dpp", "quad_perm:[2,3,0,1]")
SL_DPP_OP(max_hmir, "v_max_f32_dpp", "row_half_mirror")
SL_DPP_OP(max_mir, "v_max_f32_dpp", "row_mirror")
SL_DPP_OP(add_qp1, "v_add_f32_dpp", "quad_perm:[1,0,3,2]")
SL_DPP_OP(add_qp2, "v_add_f32_dpp", "quad_perm:[2,3,0,1]")
SL_DPP_OP(add_hmir, "v_add_f32_dpp", "row_half_mirror")
SL_DPP_OP(add_mir, "v_add_f32_dpp", "row_mirror")
#undef SL_DPP_OP

template <int G, bool SUMOP>
__device__ inline float group_allreduce_asm(float v) {
  if constexpr (SUMOP) {
    if constexpr (G >= 2) v = add_qp1(v);
    if constexpr (G >= 4) v = add_qp2(v);
    if constexpr (G >= 8) v = add_hmir(v);
    if constexpr (G >= 16) v = add_mir(v);
    if constexpr (G >= 32) v += __shfl_xor(v, 16, 64);
    if constexpr (G >= 64) v += __shfl_xor(v, 32, 64);
  } else {
    if constexpr (G >= 2) v = max_qp1(v);
    if constexpr (G >= 4) v = max_qp2(v);
    if constexpr (G >= 8) v = max_hmir(v);
    if constexpr (G >= 16) v = max_mir(v);
    if constexpr (G >= 32) v = v_max2(v, __shfl_xor(v, 16, 64));
    if constexpr (G >= 64) v = v_max2(v, __shfl_xor(v, 32, 64));
  }
  return v;
}

// Row-broadcast steps of a wave64 reduction (gfx9 DPP): row_bcast:15 folds the last lane of rows 0 / 2 into rows 1 / 3,
// row_bcast:31 folds lane 31 into rows 2 and 3.  In-place (rows that are masked out keep their value).  After the
// within-16 all-reduce plus these, lane 31 (G = 32: and lane 63) / lane 63 (G = 64) holds the group's result — without
// the two ds_bpermute round trips (~200 dependent cycles per row) that __shfl_xor costs.
#define SL_DPP_BCAST(name, insn, ctrl, mask)                                                              \
  __device__ inline float name(float a) {                                                                 \
    asm("s_nop 1\n\t" insn " %0, %0, %0 " ctrl " row_mask:" mask " bank_mask:0xf" : "+v"(a));            \
    return a;                                                                                             \
  }
SL_DPP_BCAST(max_bc15, "v_max_f32_dpp", "row_bcast:15", "0xa")
SL_DPP_BCAST(max_bc31, "v_max_f32_dpp", "row_bcast:31", "0xc")
SL_DPP_BCAST(add_bc15, "v_add_f32_dpp", "row_bcast:15", "0xa")
SL_DPP_BCAST(add_bc31, "v_add_f32_dpp", "row_bcast:31", "0xc")
#undef SL_DPP_BCAST

// all-reduce over aligned groups of G lanes whose result every lane of the group needs: DPP within 16 lanes, then for
// G = 32 / 64 row broadcasts + v_readlane (the value comes back wave-uniform per group)
template <int G, bool SUMOP>
__device__ inline float group_allreduce_bcast(float v, int lane) {
  if constexpr (G <= 16) return group_allreduce_asm<G, SUMOP>(v);
  v = group_allreduce_asm<16, SUMOP>(v);
  v = SUMOP ? add_bc15(v) : max_bc15(v);
  if constexpr (G == 64) {
    v = SUMOP ? add_bc31(v) : max_bc31(v);
    return bits_f32((uint32_t)__builtin_amdgcn_readlane((int)f32_bits(v), 63));
  } else {
    const float lo = bits_f32((uint32_t)__builtin_amdgcn_readlane((int)f32_bits(v), 31));
    const float hi = bits_f32((uint32_t)__builtin_amdgcn_readlane((int)f32_bits(v), 63));
    return lane < 32 ? lo : hi;
  }
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef SL_BIG_U
#define SL_BIG_U 4
#endif
#ifndef SL_LOAD_AUX
#define SL_LOAD_AUX 2  // cache-policy bits of the streaming loads: 2 = nt (read-once stream; +4..10 % over 0, A/B measured)
#endif

template <int G, int U, int OP, bool ALIGNED, int AUX>
__global__ __launch_bounds__(256) void rowreduce_fast_kernel(const float* __restrict__ x, int64_t R, int S, float denom,
                                                              uint16_t* __restrict__ cand,
                                                              float* __restrict__ outf, int reverse, int64_t tail_from) {
  constexpr int RPT = kWave / G;
  constexpr bool SUMOP = (OP == OP_SUM || OP == OP_ABSSUM);
  constexpr bool ABS = (OP == OP_ABSMAX || OP == OP_ABSSUM);
  const float fill = SUMOP ? 0.f : -__builtin_huge_valf();
  const int lane = threadIdx.x & 63;
  const int li = lane & (G - 1);
  const int g = lane / G;
  const int64_t ntask = R / RPT;  // launcher guarantees R % RPT == 0 and total % 4 == 0
  const int64_t nbatch = (ntask + U - 1) / U;
  const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int npieces = ALIGNED ? S / 4 : (S + 6) / 4;     // pieces of one row window
  const int nsteps = ALIGNED ? (npieces + G - 1) / G : 1;  // unaligned rows: one piece per lane
  const int h = ALIGNED ? 0 : ((g * S) & 3);
  const uint32_t row_byte0 = (uint32_t)(((g * S) >> 2) * 16);  // lane-group's row start inside the task
  const uint32_t task_bytes = (uint32_t)(RPT * S) * 4u;        // multiple of 16
  // element masks of this lane's piece (unaligned rows only; loop invariant)
  const int pos0 = li * 4 - h;
  const bool k0 = (unsigned)(pos0 + 0) < (unsigned)S, k1 = (unsigned)(pos0 + 1) < (unsigned)S;
  const bool k2 = (unsigned)(pos0 + 2) < (unsigned)S, k3 = (unsigned)(pos0 + 3) < (unsigned)S;

  for (int64_t tbi = wave0; tbi < nbatch; tbi += nwaves) {
    // reverse: walk the tensor from its end, i.e. most-recently-written first when the producer kernel
    // has just finished and its tail is still in the L2 / Infinity Cache
    const int64_t tb = reverse ? nbatch - 1 - tbi : tbi;
    const int64_t task0 = tb * U;
    int nu = U;  // tasks that exist in this batch (wave-uniform)
    if (task0 + U > ntask) nu = (int)(ntask - task0);
    // Buffer descriptor over this batch's bytes, built from provably wave-uniform halves of the base
    // pointer so every load is `buffer_load_dwordx4 v, v_off, s[rsrc], s_off offen` (no 64-bit VALU
    // address math, no waterfall loop); the hardware range check makes out-of-batch reads return 0.
    const uint64_t bptr = (uint64_t)(reinterpret_cast<const char*>(x) + task0 * (int64_t)task_bytes);
    const uint32_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bptr);
    const uint32_t bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bptr >> 32));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((uint64_t)bhi << 32) | blo), 0, (int)((uint32_t)nu * task_bytes), 0x00020000);
    float m[U], sum[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      m[u] = fill;
      sum[u] = 0.f;
    }
    for (int step = 0; step < nsteps; ++step) {
      const int q = step * G + li;
      uint32_t off;
      bool piece_ok = true;
      if constexpr (ALIGNED) {
        piece_ok = q < npieces;
        off = row_byte0 + (uint32_t)(piece_ok ? q : npieces - 1) * 16u;  // past the row: re-read its last piece
      } else {
        // past the row's window (li >= npieces): clamp onto the window's last piece; all masks are false there
        off = row_byte0 + (uint32_t)(li < npieces ? li : npieces - 1) * 16u;
      }
      float4 v[U];
      // batches from `tail_from` on (the part of a just-produced input that is still in the Infinity Cache) are read
      // with the default policy, the rest (already evicted to HBM) with the streaming one; wave-uniform choice
      if (AUX != 0 && __builtin_amdgcn_readfirstlane((int)(tb >= tail_from))) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, (int)((uint32_t)u * task_bytes), 0);
          v[u] = make_float4(bits_f32(w[0]), bits_f32(w[1]), bits_f32(w[2]), bits_f32(w[3]));
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, (int)((uint32_t)u * task_bytes), AUX);
          v[u] = make_float4(bits_f32(w[0]), bits_f32(w[1]), bits_f32(w[2]), bits_f32(w[3]));
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float e0 = v[u].x, e1 = v[u].y, e2 = v[u].z, e3 = v[u].w;
        if constexpr (ABS) {
          e0 = __builtin_fabsf(e0); e1 = __builtin_fabsf(e1); e2 = __builtin_fabsf(e2); e3 = __builtin_fabsf(e3);
        }
        if constexpr (!ALIGNED) {
          e0 = k0 ? e0 : fill; e1 = k1 ? e1 : fill; e2 = k2 ? e2 : fill; e3 = k3 ? e3 : fill;
        }
        if constexpr (SUMOP) {
          float ps = (e0 + e1) + (e2 + e3);
          if constexpr (ALIGNED) ps = piece_ok ? ps : 0.f;
          sum[u] += ps;
        } else {
          m[u] = v_max3(v_max3(m[u], e0, e1), e2, e3);
          sum[u] += (e0 + e1) + (e2 + e3);  // NaN detector only
        }
      }
    }
    float r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (SUMOP) {
        r[u] = group_allreduce_asm<G, true>(sum[u]) / denom;  // torch: sum / n (denom = 1: plain sum)
      } else {
        r[u] = group_allreduce_asm<G, false>(m[u]);
        // any lane of the wave saw a NaN sum (a NaN, or +inf and -inf)?  Rare: re-scan those rows exactly.
        const bool row_ok = u < nu;
        if (__builtin_expect(__any(row_ok && sum[u] != sum[u]), 0)) {
          const float sred = group_allreduce_f<G, true>(sum[u]);
          bool nan = false;
          if (row_ok && sred != sred) {
            const float* rowp = x + ((task0 + u) * RPT + g) * (int64_t)S;
            for (int i = li; i < S; i += G) nan |= (rowp[i] != rowp[i]);
          }
          const float f = group_allreduce_f<G, false>(nan ? 1.f : 0.f);
          if (f > 0.f) r[u] = bits_f32(0x7FC00000u);
        }
      }
    }
    // After the all-reduce every lane of a group holds its row's result for each u.  Lane li of group g
    // keeps r[p + li] and stores it: one masked store instruction per G tasks instead of one per task.
#pragma unroll
    for (int p = 0; p < U; p += G) {
      float sel = r[p];
#pragma unroll
      for (int u = p + 1; u < U && u < p + G; ++u) sel = (li == u - p) ? r[u] : sel;
      const int uu = p + li;
      if (li < G && uu < nu) store_outputs(sel, (task0 + uu) * RPT + g, cand, outf);
    }
  }
}

// streaming (read-once) 16-byte load with the nt cache policy, like the buffer loads of rowreduce_fast
__device__ inline float4 nt_load4(const float* p) {
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}

// ---- 2-byte activations (fp16 / bf16 models): element tags are _Float16 and uint16_t (bf16 bits) ---------------------
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <typename T>
__device__ inline void unpack2(uint32_t w, float& lo, float& hi);
template <>
__device__ inline void unpack2<uint16_t>(uint32_t w, float& lo, float& hi) {
  lo = bits_f32(w << 16);
  hi = bits_f32(w & 0xFFFF0000u);
}
template <>
__device__ inline void unpack2<_Float16>(uint32_t w, float& lo, float& hi) {
  const f16x2 h = __builtin_bit_cast(f16x2, w);
  lo = (float)h[0];
  hi = (float)h[1];
}
template <typename T>
__device__ inline float elem_as_f32(T v) {
  if constexpr (sizeof(T) == 4) {
    return v;
  } else {
    float lo, hi;
    unpack2<T>((uint32_t)__builtin_bit_cast(uint16_t, v), lo, hi);
    return lo;
  }
}
// four consecutive elements as floats: one 16-byte (fp32) or 8-byte (fp16 / bf16) load, streaming policy or default
template <typename T, bool NT>
__device__ inline float4 load4_as_f32(const T* p) {
  if constexpr (sizeof(T) == 4) {
    if constexpr (NT) return nt_load4(reinterpret_cast<const float*>(p));
    else return *reinterpret_cast<const float4*>(p);
  } else {
    u32x2 w;
    if constexpr (NT) w = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
    else w = *reinterpret_cast<const u32x2*>(p);
    float4 r;
    unpack2<T>(w[0], r.x, r.y);
    unpack2<T>(w[1], r.z, r.w);
    return r;
  }
}
// round to the activation dtype first (the reference aggregates in that dtype), then report
template <typename T>
__device__ inline float round_to_dtype(float v) { return v; }
template <>
__device__ inline float round_to_dtype<_Float16>(float v) { return (float)(_Float16)v; }
template <>
__device__ inline float round_to_dtype<uint16_t>(float v) { return bf16_to_f32(f32_to_bf16_rne(v)); }

// ---- rowreduce_dma: the same row arithmetic fed through a wave-private LDS ring ----------------------------------
// rowreduce_fast maps G lanes onto a row and loads the row's 16-byte pieces straight into VGPRs, so a row of 49 (196, 784)
// floats keeps 13 of 16 (49 of 64, 196 of 256) load lanes busy: ~23 % of every wave-load re-reads a clamped piece, and a
// cold 103-411 MB input streamed at 5.4-6.1 TB/s.  Here the global side is decoupled from the row structure:
//   * a *batch* (U tasks = U * 64 / G rows, <= 4 KiB, contiguous in memory) is fetched by up to four LDS-DMA
//     instructions (global_load_lds_dwordx4, nt): 64 lanes x 16 consecutive bytes each, every lane useful, lanes past
//     the batch masked off;
//   * each wave owns TWO slots of exactly one batch each (dynamic LDS: 8 slots + 1 KiB per workgroup), which lets 5-8
//     workgroups = 20-32 waves share a CU: one batch is in flight while one is reduced; counted `s_waitcnt vmcnt`; no
//     barrier, the wave reads only what it fetched itself.  (tools/native/stream_lab.hip: a bare read-once stream
//     reaches 6.6-6.85 TB/s through LDS-DMA, 6.5-6.7 through VGPR loads; occupancy, not ring depth, is what this
//     kernel responds to: 12 waves x 3 slots 6.06 / 5.50 TB/s on the 411 / 206 MB inputs, 24 waves x 2 slots 6.37 / 5.88);
//   * lanes then read their row's pieces with ds_read_b128 from the slot — masked / clamped lanes cost LDS bandwidth,
//     of which the kernel uses ~10 %.
// Arithmetic, NaN handling, rounding and the output packing are those of rowreduce_fast.
#ifndef SL_REDUCE_LAB
#define SL_REDUCE_LAB 0  // tools/native/reduce_lab.hip: 1 = no per-row reduce / store, 2 = no NaN-detector sum (garbage results)
#endif
// lab only: cache-policy bits of the head / tail loads; TAIL_FIRST 0 = walk the tensor front to back
#ifndef SL_REDUCE_LAB_HEAD_AUX
#define SL_REDUCE_LAB_HEAD_AUX 2
#endif
#ifndef SL_REDUCE_LAB_TAIL_AUX
#define SL_REDUCE_LAB_TAIL_AUX 0
#endif
#ifndef SL_REDUCE_LAB_TAIL_FIRST
#define SL_REDUCE_LAB_TAIL_FIRST 1
#endif
#ifndef SL_REDUCE_LAB_ROT_MB
#define SL_REDUCE_LAB_ROT_MB 0  // lab only: inputs read entirely with the default policy start this many MiB before their end
#endif
constexpr int kDmaMaxBatch = 4096;  // bytes: four 1-KiB LDS-DMA instructions
constexpr int kDmaDepth = 2;        // slots per wave: one batch in flight while one is reduced
constexpr int kDmaLdsPerCu = 160 * 1024;

// T = float, _Float16 or uint16_t (bf16 bits): a 16-byte piece holds EPP = 4 or 8 elements; 2-byte rows start on any
// 2-byte boundary, so the element masks of an unaligned row cover eight positions instead of four (round 3: fp16 / bf16
// NCHW activations took rowreduce_h's VGPR loads, 3.5-3.7 TB/s at 14 x 14 and 7 x 7).
// MULTI (unaligned rows only): a row's window has more pieces than its G lanes — odd maps of 13 x 13 .. 15 x 15 (fp32, four
// rows per task) or up to 30 x 30 (fp16 with S % 4 == 0, two rows per task): the lanes walk the window in steps of G pieces
// and the element masks are recomputed per step (only a window's first and last piece are partial).
// NI = 1-KiB LDS-DMA instructions per batch: 4, or 16 for MULTI tasks of 4-16 KiB (17 x 17 .. 31 x 31 maps), one per batch.
template <typename T, int G, int U, int OP, bool ALIGNED, bool MULTI = false, int NI = kDmaMaxBatch / 1024>
__global__ __launch_bounds__(256) void rowreduce_dma_kernel(const T* __restrict__ x, int64_t R, int S, float denom, int slot_bytes,
                                                             int64_t tail_from, uint16_t* __restrict__ cand,
                                                             float* __restrict__ outf) {
  constexpr int RPT = kWave / G;
  constexpr int EPP = 16 / (int)sizeof(T);  // elements per 16-byte piece
  constexpr bool SUMOP = (OP == OP_SUM || OP == OP_ABSSUM);
  constexpr bool ABS = (OP == OP_ABSMAX || OP == OP_ABSSUM);
  extern __shared__ __align__(1024) unsigned char smem[];  // 4 waves x kDmaDepth slots of `slot_bytes` + 1 KiB (masked tail)
  const float fill = SUMOP ? 0.f : -__builtin_huge_valf();
  const int lane = threadIdx.x & 63;
  const int li = lane & (G - 1);
  const int g = lane / G;
  const int64_t ntask = R / RPT;  // launcher guarantees R % RPT == 0
  const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + wave_in_block;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  static_assert(!(ALIGNED && MULTI), "aligned rows always walk in steps");
  const int npieces = ALIGNED ? S / EPP : (S + 2 * EPP - 2) / EPP;
  const int nsteps = (ALIGNED || MULTI) ? (npieces + G - 1) / G : 1;
  const int h = ALIGNED ? 0 : ((g * S) & (EPP - 1));
  const uint32_t row_byte0 = (uint32_t)(((g * S) / EPP) * 16);
  const uint32_t task_bytes = (uint32_t)(RPT * S) * (uint32_t)sizeof(T);  // multiple of 16
  const int pos0 = li * EPP - h;
  bool km[EPP];  // element e of this lane's piece belongs to the lane's row (unaligned rows; constant per lane)
#pragma unroll
  for (int e = 0; e < EPP; ++e) km[e] = (unsigned)(pos0 + e) < (unsigned)S;
  unsigned char* ring = smem + wave_in_block * (kDmaDepth * slot_bytes);
  unsigned char* spare = smem + 4 * kDmaDepth * slot_bytes;
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const unsigned char* xb = reinterpret_cast<const unsigned char*>(x);

  // batch tb of the tensor -> slot.  ALWAYS four instructions per batch, so the counted waits are compile-time
  // constants and the loop has no data-dependent branches: lanes past the batch's bytes are masked; an instruction that
  // would be empty (short batches; the tensor's last batch) keeps lane 0 alive on the batch's first 16 bytes.
  const int tail32 = tail_from > (int64_t)0x7fffffff ? 0x7fffffff : (int)tail_from;  // first task of the default-policy tail
  auto issue = [&](int task0, int nu, int slot) __attribute__((always_inline)) {
    const uint32_t nb = (uint32_t)nu * task_bytes;
    const unsigned char* src = xb + task0 * (int64_t)task_bytes;
    unsigned char* d = ring + slot * slot_bytes;
    // cache policy, wave-uniform per batch: streaming (nt) for bytes that come from HBM, default for the part of a
    // just-written input that the Infinity Cache still holds (see launch_rowreduce_dma)
    const bool stream = task0 < tail32;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const uint32_t byte = (uint32_t)i * 1024u + (uint32_t)lane * 16u;
      const bool in = byte < nb;
      // an instruction with no byte of the batch left still goes out (lane 0, the batch's first piece) but lands in the
      // workgroup's spare KiB, not in a slot
      unsigned char* dst_i = (uint32_t)i * 1024u < nb ? d + i * 1024 : spare;
      if (in || lane == 0) {
        if (stream) __builtin_amdgcn_global_load_lds((glb_void*)(src + (in ? byte : 0u)), (lds_void*)dst_i, 16, 0, SL_REDUCE_LAB_HEAD_AUX /* 2 = nt */);
        else __builtin_amdgcn_global_load_lds((glb_void*)(src + (in ? byte : 0u)), (lds_void*)dst_i, 16, 0, SL_REDUCE_LAB_TAIL_AUX);
      }
    }
  };
  auto wait_batches = [&](int younger) __attribute__((always_inline)) {  // at most `younger` batches still in flight
    switch (younger * NI) {
#define SL_VMCNT_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
      SL_VMCNT_CASE(0) SL_VMCNT_CASE(4) SL_VMCNT_CASE(8) SL_VMCNT_CASE(12) SL_VMCNT_CASE(16) SL_VMCNT_CASE(24)
#undef SL_VMCNT_CASE
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };

  // Work split.  `full` rounds in which every wave owns a whole batch of U tasks, batches interleaved over the waves
  // (neighbouring waves read neighbouring bytes); the remaining `rem` tasks (< one round) are split EVENLY over the waves
  // as one short batch each instead of leaving most waves idle for a round (the layer4 shape has 5.33 batches per wave:
  // a 6th round for a third of the waves cost 10 %).  The short round covers the tensor's first tasks and is walked last.
  // Walk order of the full rounds: when a tail policy is active (tail_from inside the tensor) the tail goes FIRST —
  // the most recently written bytes are read while the Infinity Cache still holds them, before the head's traffic can
  // displace them (in-pipeline 411 MB: 6.07 -> 6.44 TB/s).
  // 32-bit task / batch indices (the launcher keeps ntask < 2^31): 64-bit scalar compares compile to VALU compares
  // whose result the scalar branch then waits for.
  const int ntask32 = (int)ntask, nwaves32 = (int)nwaves, w0 = (int)wave0;
  const int round_tasks = nwaves32 * U;
  const int full = ntask32 / round_tasks;
  const int rem = ntask32 - full * round_tasks;
  const int u_last = (rem + nwaves32 - 1) / nwaves32;  // <= U
  const int nfull = full * nwaves32;                   // whole batches
  int rot = 0;
  if (SL_REDUCE_LAB_TAIL_FIRST && tail32 > rem && tail32 < ntask32) rot = (tail32 - rem) / U;
  if (SL_REDUCE_LAB_ROT_MB > 0 && tail32 == 0) {
    const int back = (int)(((int64_t)SL_REDUCE_LAB_ROT_MB << 20) / ((int64_t)U * task_bytes));
    rot = back < nfull ? nfull - back : 0;
  }
  const int nmine = full + ((int64_t)w0 * u_last < rem ? 1 : 0);
  auto work_of = [&](int it, int& t0, int& n) __attribute__((always_inline)) {
    if (it < full) {
      int v = w0 + it * nwaves32 + rot;
      if (v >= nfull) v -= nfull;
      t0 = rem + v * U;
      n = U;
    } else {
      t0 = w0 * u_last;
      n = rem - t0 < u_last ? rem - t0 : u_last;
    }
  };
  int task0 = 0, task_next = 0;
  int nu = 0, nu_next = 0;
  if (nmine > 0) {
    work_of(0, task_next, nu_next);
    issue(task_next, nu_next, 0);
  }
  static_assert(kDmaDepth == 2, "the loop below keeps exactly one batch in flight beside the one being reduced");
  for (int it = 0; it < nmine; ++it) {
    const int slot = it & 1;
    task0 = task_next;
    nu = nu_next;
    if (it + 1 < nmine) {
      work_of(it + 1, task_next, nu_next);
      issue(task_next, nu_next, slot ^ 1);
      wait_batches(1);  // a constant: one s_waitcnt
    } else {
      wait_batches(0);  // the wave's last batch: drain
    }
    const unsigned char* sl_ = ring + slot * slot_bytes;
    float m[U], sum[U];
    f32x2 sum2[U];  // two partial sums per task, added with one v_pk_add_f32 per half piece
#pragma unroll
    for (int u = 0; u < U; ++u) {
      m[u] = fill;
      sum2[u] = f32x2{0.f, 0.f};
    }
    for (int step = 0; step < nsteps; ++step) {
      const int q = step * G + li;
      uint32_t off;
      bool piece_ok = true;
      if constexpr (ALIGNED) {
        piece_ok = q < npieces;
        off = row_byte0 + (uint32_t)(piece_ok ? q : npieces - 1) * 16u;
      } else if constexpr (MULTI) {
        off = row_byte0 + (uint32_t)(q < npieces ? q : npieces - 1) * 16u;
        const int pos = q * EPP - h;  // q >= npieces: pos >= S, every mask false
#pragma unroll
        for (int e = 0; e < EPP; ++e) km[e] = (unsigned)(pos + e) < (unsigned)S;
      } else {
        off = row_byte0 + (uint32_t)(li < npieces ? li : npieces - 1) * 16u;
      }
      f32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const f32x4*>(sl_ + (uint32_t)u * task_bytes + off);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float e[EPP];
        if constexpr (sizeof(T) == 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) e[i] = v[u][i];
        } else {
#pragma unroll
          for (int d = 0; d < 4; ++d) unpack2<T>(f32_bits(v[u][d]), e[2 * d], e[2 * d + 1]);
        }
#pragma unroll
        for (int i = 0; i < EPP; ++i) {
          if constexpr (ABS) e[i] = __builtin_fabsf(e[i]);
          if constexpr (!ALIGNED) e[i] = km[i] ? e[i] : fill;
          if constexpr (SUMOP && ALIGNED) e[i] = piece_ok ? e[i] : 0.f;
        }
        f32x2 ps = f32x2{e[0], e[1]} + f32x2{e[2], e[3]};
        if constexpr (EPP == 8) ps += f32x2{e[4], e[5]} + f32x2{e[6], e[7]};
        if constexpr (SUMOP) {
          sum2[u] += ps;
        } else {
          m[u] = v_max3(v_max3(m[u], e[0], e[1]), e[2], e[3]);
          if constexpr (EPP == 8) m[u] = v_max3(v_max3(m[u], e[4], e[5]), e[6], e[7]);
          if (!(SL_REDUCE_LAB & 2)) sum2[u] += ps;  // NaN detector only
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) sum[u] = sum2[u][0] + sum2[u][1];
    // every ds_read of the slot has returned before a later iteration's DMA may overwrite it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (SL_REDUCE_LAB & 1) {  // measurement only: no per-row reduction, no stores
      float t = 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u) t += m[u] + sum[u];
      if (t == 12345.f && outf) outf[0] = t;
      continue;
    }
    float r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if constexpr (SUMOP) {
        // sums accumulate in fp32 and are rounded ONCE to the activation dtype, like torch's (identity for fp32)
        r[u] = round_to_dtype<T>(group_allreduce_bcast<G, true>(sum[u], lane) / denom);
      } else {
        r[u] = group_allreduce_bcast<G, false>(m[u], lane);
        const bool row_ok = u < nu;
        // rows of a short last batch read stale slot bytes: their sums are ignored (row_ok)
        if (__builtin_expect(__any(row_ok && sum[u] != sum[u]), 0)) {
          const float sred = group_allreduce_f<G, true>(sum[u]);
          bool nan = false;
          if (row_ok && sred != sred) {
            const T* rowp = x + ((int64_t)(task0 + u) * RPT + g) * (int64_t)S;
            for (int i = li; i < S; i += G) {
              const float ev = elem_as_f32<T>(rowp[i]);
              nan |= (ev != ev);
            }
          }
          const float f = group_allreduce_f<G, false>(nan ? 1.f : 0.f);
          if (f > 0.f) r[u] = bits_f32(0x7FC00000u);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < U; p += G) {
      float sel = r[p];
#pragma unroll
      for (int u = p + 1; u < U && u < p + G; ++u) sel = (li == u - p) ? r[u] : sel;
      const int uu = p + li;
      if (li < G && uu < nu) store_outputs(sel, (int64_t)(task0 + uu) * RPT + g, cand, outf);
    }
  }
}

// ---- colreduce: out[b][f] = op_t x[b][t][f], f contiguous ------------------------------------
// One workgroup (NW = 4 waves, or 16 when there are too few (b, chunk) tasks to fill the chip — small batches of long
// token sequences) per (b, 256-float chunk of F); the waves split T; LDS combine.
// Cache policy as in the row kernels (top of this file): tasks below `tail_from` (in memory order: b-major) stream with
// nt, the rest use the default policy, and the walk starts at `tail_from` so that the bytes written last are read first.
template <typename E, int OP, int NW>
__global__ __launch_bounds__(64 * NW) void colreduce_kernel(const E* __restrict__ x, int64_t B, int T, int64_t F,
                                                         int64_t sb, int64_t st, int t_begin, int t_end,
                                                         float denom, int64_t tail_from, uint16_t* __restrict__ cand,
                                                         float* __restrict__ outf) {
  __shared__ float s_part[NW][256];
  constexpr bool SUM = (OP == OP_SUM || OP == OP_ABSSUM);
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int64_t nchunk = (F + 255) / 256;
  const int64_t ntask = B * nchunk;
  const int64_t rot = (tail_from > 0 && tail_from < ntask) ? tail_from : 0;
  for (int64_t ti = blockIdx.x; ti < ntask; ti += gridDim.x) {
    int64_t task = ti + rot;
    if (task >= ntask) task -= ntask;
    const int64_t b = task / nchunk;
    const int64_t f0 = (task % nchunk) * 256 + lane * 4;
    Acc<OP> a0, a1, a2, a3;
    a0.init(); a1.init(); a2.init(); a3.init();
    const bool in = f0 < F;  // F % 4 == 0 on this path
    const E* base = x + b * sb + f0;
    auto walk = [&](auto NT) __attribute__((always_inline)) {
      constexpr bool nt = decltype(NT)::value;
      auto ld = [&](const E* p) __attribute__((always_inline)) { return load4_as_f32<E, nt>(p); };
      int t = t_begin + w;
#pragma unroll 1
      for (; t + 7 * NW < t_end; t += 8 * NW) {  // 8 loads in flight per lane
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ld(base + (int64_t)(t + NW * j) * st);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          a0.add(v[j].x, true); a1.add(v[j].y, true); a2.add(v[j].z, true); a3.add(v[j].w, true);
        }
      }
      for (; t < t_end; t += NW) {
        float4 v = ld(base + (int64_t)t * st);
        a0.add(v.x, true); a1.add(v.y, true); a2.add(v.z, true); a3.add(v.w, true);
      }
    };
    if (in) {
      if (task < tail_from) walk(std::true_type());
      else walk(std::false_type());
    }
    s_part[w][lane * 4 + 0] = a0.lane_value();
    s_part[w][lane * 4 + 1] = a1.lane_value();
    s_part[w][lane * 4 + 2] = a2.lane_value();
    s_part[w][lane * 4 + 3] = a3.lane_value();
    __syncthreads();
    if (threadIdx.x < 256) {
      const int f = threadIdx.x;  // 256 threads -> 256 features of the chunk
      const int64_t fg = (task % nchunk) * 256 + f;
      if (fg < F) {
        float r = s_part[0][f];
#pragma unroll
        for (int i = 1; i < NW; ++i) r = combine<SUM>(r, s_part[i][f]);
        r = round_to_dtype<E>(finish<OP>(r, denom));  // the reference aggregates in the activation's dtype
        store_outputs(r, b * F + fg, cand, outf);
      }
    }
    __syncthreads();
  }
}

// ---- colreduce2 (round 4): 16-byte pieces for every dtype, LPR lanes per row, loads never drain ------------------------------
// What the round-4 lab (tools/k2_lab.py, profiles/r04_k2_lab.txt) found wrong with colreduce_kernel on (256, 197, 768):
//  * its row tail ran ONE load per lane and iteration: 197 = 6 x 32 + 5 rows left two serialised memory round trips (~2 us
//    each) at the end of a 26-us launch; (256, 196, 1024) — one tail trip, 16 waves per CU — ran 6.3 TB/s, this shape 5.9;
//  * a round of eight loads was reduced before the next eight were issued: the bytes in flight swung between 8 KB per wave and
//    nothing (the LDS-DMA ring kernel above removes that too, but its rings cap a CU at 16 waves: 5.6-6.1 TB/s);
//  * half-precision rows were read with 8-byte loads (512 B per wave instruction): 5.0 TB/s where fp32 reads 5.9.
// Here every load is 16 bytes (a *piece*: 4 fp32 or 8 half components).  A row chunk is LPR pieces (64, 32 or 16 lanes), so a
// wave instruction covers 64 / LPR rows x LPR x 16 bytes = 1 KiB whatever the row length: fp16 F = 768 (96 pieces) takes LPR = 32
// (three full chunks) instead of 1.5 chunks of 64; lanes that share a piece column combine once at the end (one xor-shuffle per
// level).  Rows are walked in blocks of INFL loads per lane; block k + 1 is issued BEFORE block k is reduced (two register
// sets, ping-pong), so 8-16 loads per lane are in flight from the first block to the last.  No predicated loads: a row past the
// end is clamped to the last row (a cache hit) and its values are discarded by the accumulator's `valid` flag, so the last
// block costs one round trip like any other.
// Round 5: the input is a TABLE of up to kMaxReduceSources same-shape tensors (the outputs of L identical transformer blocks,
// kept alive until the last one exists): "virtual" batch b of the B = L * per batches lives in tensor b / per at batch b % per,
// and the (L, per, F) outputs are one contiguous buffer, so nothing else in the kernel changes.  One 1.9 GB launch instead of
// twelve 155 MB ones: the ~2.5 us a launch costs beyond bytes / 6.45 TB/s is paid once (a single tensor is a table of one).
constexpr int kMaxReduceSources = 32;
struct MultiSrc {
  const void* ptr[kMaxReduceSources];
  int64_t per;  // batches per tensor
};

template <typename E, int OP, int NW, int LPR, int INFL>
__global__ __launch_bounds__(64 * NW) void colreduce2_kernel(MultiSrc src, int64_t B, int T, int64_t F, int64_t sb,
                                                              int64_t st, int t_begin, int t_end, float denom, int64_t tail_from,
                                                              uint16_t* __restrict__ cand, float* __restrict__ outf) {
  constexpr int EPP = 16 / (int)sizeof(E);
  constexpr int RPI = 64 / LPR;  // rows per wave instruction
  constexpr int CW = LPR * EPP;  // components per chunk
  constexpr bool SUM = (OP == OP_SUM || OP == OP_ABSSUM);
  constexpr bool ABS = (OP == OP_ABSMAX || OP == OP_ABSSUM);
  constexpr int STEP = NW * RPI;  // rows one instruction of every wave of the workgroup covers
  __shared__ float s_part[NW][CW];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int sub = lane / LPR, pl = lane % LPR;
  const int64_t nchunk = (F + CW - 1) / CW;
  const int64_t ntask = B * nchunk;
  const int64_t rot = (tail_from > 0 && tail_from < ntask) ? tail_from : 0;
  const int tw = t_begin + w * RPI + sub;  // this lane's first row
  const int last = t_end - 1;
  const int rows_w = t_end - (t_begin + w * RPI);                 // rows from the wave's first row on
  const int ninst = rows_w > 0 ? (rows_w + STEP - 1) / STEP : 0;  // wave instructions that touch a valid row
  const int nblk = (ninst + INFL - 1) / INFL;
  const int64_t row_pieces = st * (int64_t)sizeof(E) / 16;
  auto elems = [&](const u32x4& v, float(&e)[EPP]) __attribute__((always_inline)) {
    if constexpr (EPP == 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) e[i] = bits_f32(v[i]);
    } else {
#pragma unroll
      for (int d = 0; d < 4; ++d) unpack2<E>(v[d], e[2 * d], e[2 * d + 1]);
    }
    if constexpr (ABS) {
#pragma unroll
      for (int i = 0; i < EPP; ++i) e[i] = __builtin_fabsf(e[i]);
    }
  };
  for (int64_t ti = blockIdx.x; ti < ntask; ti += gridDim.x) {
    int64_t task = ti + rot;
    if (task >= ntask) task -= ntask;
    const int64_t b = task / nchunk;
    const int64_t f0 = (task % nchunk) * CW + (int64_t)pl * EPP;
    const bool in = f0 < F;  // F % EPP == 0 on this path; lanes past the row re-read its first piece and are never stored
    const E* x = static_cast<const E*>(src.ptr[b / src.per]);
    const u32x4* base = reinterpret_cast<const u32x4*>(x + (b % src.per) * sb + (in ? f0 : 0));
    // max ops: v_max_f32 drops NaN, torch.amax propagates it.  As in K1 a running SUM rides along (v_pk_add_f32: NaN in => NaN
    // out) and only columns whose sum is NaN (a NaN, or +inf with -inf) are looked at again, exactly.  Rows past the end are
    // CLAMPED to the last row: a duplicate changes neither a max nor the detector's verdict; sums mask them instead.
    float m[EPP];
    f32x2 det[EPP / 2];
#pragma unroll
    for (int e = 0; e < EPP; ++e) m[e] = SUM ? 0.f : -__builtin_huge_valf();
#pragma unroll
    for (int e = 0; e < EPP / 2; ++e) det[e] = f32x2{0.f, 0.f};
    auto walk = [&](auto NT) __attribute__((always_inline)) {
      constexpr bool nt = decltype(NT)::value;
      auto load = [&](u32x4(&v)[INFL], int k) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < INFL; ++j) {
          int row = tw + (k * INFL + j) * STEP;
          row = row < last ? row : last;
          const u32x4* p = base + (int64_t)row * row_pieces;
          if constexpr (nt) v[j] = __builtin_nontemporal_load(p);
          else v[j] = *p;
        }
      };
      auto reduce = [&](const u32x4(&v)[INFL], int k) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < INFL; ++j) {
          float e[EPP];
          elems(v[j], e);
          if constexpr (SUM) {
            const bool ok = tw + (k * INFL + j) * STEP <= last;
#pragma unroll
            for (int i = 0; i < EPP; ++i) m[i] += ok ? e[i] : 0.f;
          } else {
#pragma unroll
            for (int i = 0; i < EPP; ++i) m[i] = __builtin_fmaxf(m[i], e[i]);
#pragma unroll
            for (int i = 0; i < EPP / 2; ++i) det[i] += f32x2{e[2 * i], e[2 * i + 1]};
          }
        }
      };
      u32x4 va[INFL], vb[INFL];
      if (nblk > 0) load(va, 0);
#pragma unroll 1
      for (int k = 0; k < nblk; k += 2) {
        const bool more1 = k + 1 < nblk;
        if (more1) load(vb, k + 1);
        reduce(va, k);
        if (more1) {
          if (k + 2 < nblk) load(va, k + 2);
          reduce(vb, k + 1);
        }
      }
    };
    if (task < tail_from) walk(std::true_type());
    else walk(std::false_type());
    if constexpr (!SUM) {
      bool sus = false;
#pragma unroll
      for (int i = 0; i < EPP / 2; ++i) sus |= (det[i][0] != det[i][0]) | (det[i][1] != det[i][1]);
      if (__builtin_expect(__any(sus), 0)) {  // rare: a NaN, or +inf and -inf in one column — look again, exactly
        bool nan[EPP];
#pragma unroll
        for (int i = 0; i < EPP; ++i) nan[i] = false;
        if (sus) {
          for (int row = tw; row <= last; row += STEP) {
            float e[EPP];
            elems(base[(int64_t)row * row_pieces], e);
#pragma unroll
            for (int i = 0; i < EPP; ++i) nan[i] |= (e[i] != e[i]);
          }
        }
#pragma unroll
        for (int i = 0; i < EPP; ++i) m[i] = nan[i] ? bits_f32(0x7FC00000u) : m[i];
      }
    }
#pragma unroll
    for (int e = 0; e < EPP; ++e) {
      float r = m[e];
      if constexpr (RPI >= 4) r = combine<SUM>(r, __shfl_xor(r, 16, 64));
      if constexpr (RPI >= 2) r = combine<SUM>(r, __shfl_xor(r, 32, 64));
      if (lane < LPR) s_part[w][lane * EPP + e] = r;
    }
    __syncthreads();
    for (int f = threadIdx.x; f < CW; f += 64 * NW) {
      const int64_t fg = (task % nchunk) * CW + f;
      if (fg < F) {
        float v = s_part[0][f];
#pragma unroll
        for (int i = 1; i < NW; ++i) v = combine<SUM>(v, s_part[i][f]);
        v = round_to_dtype<E>(finish<OP>(v, denom));  // the reference aggregates in the activation's dtype
        store_outputs(v, b * F + fg, cand, outf);
      }
    }
    __syncthreads();
  }
}

// ---- rowreduce_h: contiguous rows of 2-byte elements (NCHW activations of fp16 / bf16 models) -------------------------
// x 16-byte aligned, R rows of S elements back to back.  G lanes per row (64 / G rows = one *set* per wave pass); a lane
// loads 16-byte pieces (8 elements) of its row's window, converts to fp32 and masks the elements that belong to the
// neighbouring rows (rows start on 2-byte boundaries).  An aligned 16-byte piece that holds at least one valid byte never
// crosses a page, so the window's first and last piece are safe to read whole.  U sets x J pieces per lane are in flight
// before any is reduced (short rows would otherwise keep < 1 KB per wave in flight).  max: v_max_f32 drops NaN, so a
// running sum rides along as the NaN detector and a row whose sum is NaN is re-scanned exactly (as in the fp32 kernels).
// Same cache policy and tail-first walk as the fp32 kernels; sums accumulate in fp32 and are rounded to the activation
// dtype once, like torch's.
template <typename T, int G, int U, int J, int OP, bool ALIGNED>
__global__ __launch_bounds__(256) void rowreduce_h_kernel(const T* __restrict__ x, int64_t R, int S, float denom, int64_t tail_from,
                                                           uint16_t* __restrict__ cand, float* __restrict__ outf) {
  constexpr int RPW = kWave / G;
  constexpr bool SUMOP = (OP == OP_SUM || OP == OP_ABSSUM);
  constexpr bool ABS = (OP == OP_ABSMAX || OP == OP_ABSSUM);
  const float fill = SUMOP ? 0.f : -__builtin_huge_valf();
  // the same value as a pair of raw elements (-inf is 0xFF80 in bf16, 0xFC00 in fp16).  The aligned path takes |.| of
  // whole pieces, fill included, so absmax fills with +0 (|x| >= 0 makes it neutral; |-inf| would be +inf).
  const uint32_t fillw = (SUMOP || ABS) ? 0u : (std::is_same<T, uint16_t>::value ? 0xFF80FF80u : 0xFC00FC00u);
  const int lane = threadIdx.x & 63;
  const int li = lane & (G - 1);
  const int g = lane / G;
  const int64_t nsets = (R + RPW - 1) / RPW;
  const int64_t nbatch = (nsets + U - 1) / U;  // a batch = U consecutive sets
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int64_t rot = (tail_from > 0 && tail_from < nbatch) ? tail_from : 0;
  const int np_max = ALIGNED ? S / 8 : (S + 14) / 8;  // pieces a row's window can touch
  const u32x4* xp = reinterpret_cast<const u32x4*>(x);
  for (int64_t bi = wave0; bi < nbatch; bi += nwaves) {
    int64_t batch = bi + rot;
    if (batch >= nbatch) batch -= nbatch;
    int64_t row[U];
    int h[U], np[U];
    const u32x4* rp[U];
    float m[U], sum[U];
    f32x2 sum2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      row[u] = (batch * U + u) * RPW + g;
      const bool ok = row[u] < R;
      const int64_t e0 = (ok ? row[u] : 0) * (int64_t)S;
      h[u] = ALIGNED ? 0 : (int)(e0 & 7);
      np[u] = ok ? (h[u] + S + 7) >> 3 : 0;
      rp[u] = xp + (e0 >> 3);
      m[u] = fill;
      sum[u] = 0.f;
      sum2[u] = f32x2{0.f, 0.f};
    }
    auto walk = [&](auto NT) __attribute__((always_inline)) {
      constexpr bool nt = decltype(NT)::value;
      for (int q0 = 0; q0 < np_max; q0 += J * G) {
        u32x4 w[U][J];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int j = 0; j < J; ++j) {
            const int q = q0 + j * G + li;
            w[u][j] = u32x4{fillw, fillw, fillw, fillw};  // lanes without a piece contribute the fill value
            if (q < np[u]) {
              if constexpr (nt) w[u][j] = __builtin_nontemporal_load(rp[u] + q);
              else w[u][j] = rp[u][q];
            }
          }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int j = 0; j < J; ++j) {
            const int q = q0 + j * G + li;
            const bool in = q < np[u];
            const int idx0 = q * 8 - h[u];  // row-relative index of the piece's first element
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              float e[2];
              unpack2<T>(w[u][j][d], e[0], e[1]);
              if constexpr (ALIGNED) {  // whole pieces: two elements per v_max3 / v_pk_add
                if constexpr (ABS) {
                  e[0] = __builtin_fabsf(e[0]);
                  e[1] = __builtin_fabsf(e[1]);
                }
                if constexpr (!SUMOP) m[u] = v_max3(m[u], e[0], e[1]);
                sum2[u] += f32x2{e[0], e[1]};  // the sum, or the NaN detector of the max
              } else {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                  float v = ABS ? __builtin_fabsf(e[k]) : e[k];
                  const bool valid = in && (unsigned)(idx0 + 2 * d + k) < (unsigned)S;
                  v = valid ? v : fill;
                  if constexpr (!SUMOP) m[u] = __builtin_fmaxf(m[u], v);
                  sum[u] += v;  // the sum, or the NaN detector of the max
                }
              }
            }
          }
      }
    };
    if (batch < tail_from) walk(std::true_type());
    else walk(std::false_type());
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = row[u] < R;
      if constexpr (ALIGNED) sum[u] = sum2[u][0] + sum2[u][1];
      float r;
      if constexpr (SUMOP) {
        r = group_allreduce_f<G, true>(sum[u]);
      } else {
        r = group_allreduce_f<G, false>(m[u]);
        if (__builtin_expect(__any(ok && sum[u] != sum[u]), 0)) {  // a NaN, or +inf and -inf (or fill) together: look again
          const float sred = group_allreduce_f<G, true>(sum[u]);
          bool nan = false;
          if (ok && sred != sred) {
            const T* rowp = x + row[u] * (int64_t)S;
            for (int i = li; i < S; i += G) {
              float lo, hi;
              unpack2<T>((uint32_t)__builtin_bit_cast(uint16_t, rowp[i]), lo, hi);
              nan |= (lo != lo);
            }
          }
          const float f = group_allreduce_f<G, false>(nan ? 1.f : 0.f);
          if (f > 0.f) r = bits_f32(0x7FC00000u);
        }
      }
      r = round_to_dtype<T>(finish<OP>(r, denom));
      if (li == 0 && ok) store_outputs(r, row[u], cand, outf);
    }
  }
}

// ---- generic: any strides, fp32 / fp16 / bf16 -------------------------------------------------
template <typename T>
__device__ inline float load_as_f32(const void* p, int64_t i);
template <>
__device__ inline float load_as_f32<float>(const void* p, int64_t i) { return ((const float*)p)[i]; }
template <>
__device__ inline float load_as_f32<_Float16>(const void* p, int64_t i) { return (float)((const _Float16*)p)[i]; }
template <>
__device__ inline float load_as_f32<uint16_t>(const void* p, int64_t i) { return bf16_to_f32(((const uint16_t*)p)[i]); }

template <typename T, int OP>
__global__ __launch_bounds__(256) void generic_reduce_kernel(const void* __restrict__ x, int64_t B, int64_t C,
                                                              int64_t S, int64_t sb, int64_t sc, int64_t ss,
                                                              int64_t s_begin, int64_t s_end, float denom,
                                                              uint16_t* __restrict__ cand, float* __restrict__ outf) {
  const int64_t n = B * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / C, c = i % C;
    const int64_t off = b * sb + c * sc;
    Acc<OP> a;
    a.init();
    for (int64_t s = s_begin; s < s_end; ++s) a.add(load_as_f32<T>(x, off + s * ss), true);
    float r = finish<OP>(a.lane_value(), denom);
    r = round_to_dtype<T>(r);
    store_outputs(r, i, cand, outf);
  }
}

template <int G, int U, int OP>
void launch_rowreduce(ProfScope& prof, const float* x, int64_t R, int S, float denom, uint16_t* cand, float* outf, hipStream_t st) {
  constexpr int RPT = kWave / G;
  const int64_t ntasks = (R + RPT - 1) / RPT;
  const int64_t nbatch = (ntasks + U - 1) / U;
  int64_t blocks = (nbatch + 3) / 4;
  const int64_t cap = (int64_t)num_cus() * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  if ((R * (int64_t)S) % 4 == 0)
    SL_LAUNCH(prof, (rowreduce_kernel<G, U, OP, false>), dim3((unsigned)blocks), dim3(256), 0, st, x, R, S, denom, cand, outf);
  else
    SL_LAUNCH(prof, (rowreduce_kernel<G, U, OP, true>), dim3((unsigned)blocks), dim3(256), 0, st, x, R, S, denom, cand, outf);
}

template <int G, int U, int OP, bool ALIGNED>
void launch_rowreduce_fast(ProfScope& prof, const float* x, int64_t R, int S, float denom, uint16_t* cand, float* outf, hipStream_t st) {
  constexpr int RPT = kWave / G;
  const int64_t nbatch = (R / RPT + U - 1) / U;
  int64_t blocks = (nbatch + 3) / 4;
  const int64_t cap = (int64_t)num_cus() * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  constexpr int reverse = 0;  // (walking the batches backwards was an A/B of round 2; the kernel keeps the parameter)
  const int64_t nt_min_bytes = nt_min_bytes_(), tail_bytes = tail_bytes_();  // cache policy: see the top of this file
  const int64_t bytes = R * (int64_t)S * 4;
  if (bytes >= nt_min_bytes) {
    const int64_t batch_bytes = (int64_t)U * RPT * S * 4;
    const int64_t tail_from = tail_bytes > 0 ? (bytes > tail_bytes ? (bytes - tail_bytes) / batch_bytes : 0) : INT64_MAX;
    SL_LAUNCH(prof, (rowreduce_fast_kernel<G, U, OP, ALIGNED, SL_LOAD_AUX>), dim3((unsigned)blocks), dim3(256), 0, st, x, R, S,
              denom, cand, outf, reverse, tail_from);
  } else {
    SL_LAUNCH(prof, (rowreduce_fast_kernel<G, U, OP, ALIGNED, 0>), dim3((unsigned)blocks), dim3(256), 0, st, x, R, S, denom,
              cand, outf, reverse, (int64_t)INT64_MAX);
  }
}

template <typename T, int G, int U, int OP, bool ALIGNED, bool MULTI = false, int NI = kDmaMaxBatch / 1024>
void launch_rowreduce_dma(ProfScope& prof, const T* x, int64_t R, int S, float denom, uint16_t* cand, float* outf, hipStream_t st) {
  constexpr int RPT = kWave / G;
  constexpr int ES = (int)sizeof(T);
  const int64_t nbatch = (R / RPT + U - 1) / U;
  const int slot = U * RPT * S * ES;                      // one batch, a multiple of 16 bytes
  const int lds = 4 * kDmaDepth * slot + 1024;            // + 1 KiB: the masked tail of the last slot's last instruction
  int per_cu = kDmaLdsPerCu / lds;
  if (per_cu > 8) per_cu = 8;                             // 32 waves per CU
  int64_t blocks = (nbatch + 3) / 4;
  const int64_t cap = (int64_t)num_cus() * per_cu;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  const int64_t nt_min_bytes = nt_min_bytes_(), tail_bytes = tail_bytes_();  // cache policy: see the top of this file
  const int64_t bytes = R * (int64_t)S * ES;
  int64_t tail_from = 0;  // tasks from here on use the default policy
  if (bytes >= nt_min_bytes) tail_from = tail_bytes > 0 ? (bytes > tail_bytes ? (bytes - tail_bytes) / ((int64_t)RPT * S * ES) : 0) : INT64_MAX;
  if (lds > 64 * 1024) {  // dynamic LDS past 64 KiB has to be allowed per kernel
    static const hipError_t allowed = hipFuncSetAttribute((const void*)rowreduce_dma_kernel<T, G, U, OP, ALIGNED, MULTI, NI>,
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, kDmaLdsPerCu);
    (void)allowed;
  }
  SL_LAUNCH(prof, (rowreduce_dma_kernel<T, G, U, OP, ALIGNED, MULTI, NI>), dim3((unsigned)blocks), dim3(256), (size_t)lds, st, x, R,
            S, denom, slot, tail_from, cand, outf);
}

// U = tasks per batch (<= 4) so that a batch is at most 4 KiB; false when a task alone is larger
template <int G, int OP, bool ALIGNED, typename T, bool MULTI = false>
bool try_rowreduce_dma(ProfScope& prof, const T* x, int64_t R, int S, float denom, uint16_t* cand, float* outf, hipStream_t st) {
  constexpr bool enabled = true;
  constexpr int RPT = kWave / G;
  const int64_t task_bytes = (int64_t)RPT * S * (int64_t)sizeof(T);
  if (!enabled || task_bytes > (MULTI && sizeof(T) == 2 ? 16 * 1024 : kDmaMaxBatch) || (task_bytes & 15) != 0 || R % RPT != 0 || R * (int64_t)S * (int64_t)sizeof(T) < (8ll << 20) ||
      R > 0x7fffffffll)
    return false;  // small inputs: launch-bound either way; the kernel indexes tasks with 32 bits
  const int u = (int)(kDmaMaxBatch / task_bytes);
  if constexpr (MULTI) {  // long windows: a task is 1-4 KiB, so one or two tasks per batch
    if (task_bytes > kDmaMaxBatch) {
      // 4-16 KiB: one task per batch of sixteen instructions, one workgroup per CU.  2-byte elements only: 17 x 17 fp16 maps
      // 2.8 -> 4.1 TB/s against rowreduce_h, which leaves 27 of 64 lanes idle there; fp32 rows of this length lose
      // (27 x 27: 5.7 -> 5.4 TB/s against launch_rowreduce<64, 4>) and stay on the VGPR-load kernel
      if constexpr (sizeof(T) == 2) {
        launch_rowreduce_dma<T, G, 1, OP, ALIGNED, true, 16>(prof, x, R, S, denom, cand, outf, st);
        return true;
      }
      return false;
    }
    if (u >= 2) launch_rowreduce_dma<T, G, 2, OP, ALIGNED, true>(prof, x, R, S, denom, cand, outf, st);
    else launch_rowreduce_dma<T, G, 1, OP, ALIGNED, true>(prof, x, R, S, denom, cand, outf, st);
    return true;
  }
  if (u >= 4) launch_rowreduce_dma<T, G, 4, OP, ALIGNED>(prof, x, R, S, denom, cand, outf, st);
  else if (u == 3) launch_rowreduce_dma<T, G, 3, OP, ALIGNED>(prof, x, R, S, denom, cand, outf, st);
  else if (u == 2) launch_rowreduce_dma<T, G, 2, OP, ALIGNED>(prof, x, R, S, denom, cand, outf, st);
  else launch_rowreduce_dma<T, G, 1, OP, ALIGNED>(prof, x, R, S, denom, cand, outf, st);
  return true;
}

template <int OP>
void dispatch_rowreduce(ProfScope& prof, const float* x, int64_t R, int S, float denom, uint16_t* cand, float* outf, hipStream_t st) {
  // pieces needed for a row window: up to (S + 6) / 4
  const int need = (S + 6) / 4;
  // fast path A: rows are whole 16-byte pieces
  if (S % 4 == 0 && S >= 16 && (int64_t)S * 64 * 4 * 8 < (1ll << 31)) {
    const int np = S / 4;
    // Rows longer than an LDS-DMA batch (> 4 KiB: 56 x 56 maps and larger) stay on rowreduce_fast<64, 4>: 6.5 TB/s cold AND behind
    // a producer on (256, 192, 56, 56) (617 MB, 0.82 of spec).  Round 4 built a ping-pong stream kernel for them (a wave walks
    // (row, block-of-4-loads) items, the next item issued before the current one is reduced): 6.1 cold / 4.6 behind a producer in
    // fp32, 5.4 against rowreduce_h's 6.1-6.3 in fp16 — removed (tools/k1_long_rows.py, profiles/r04_k1_long_rows.txt).
#define SL_ROWREDUCE(G_, U_, AL_)                                                          \
  do {                                                                                     \
    if (try_rowreduce_dma<G_, OP, AL_>(prof, x, R, S, denom, cand, outf, st)) return;             \
    return launch_rowreduce_fast<G_, U_, OP, AL_>(prof, x, R, S, denom, cand, outf, st);          \
  } while (0)
    // LDS-DMA path: lanes read from LDS, where clamped lanes are free, so four rows share a task (G = 16, up to four
    // steps per row) and their DPP reductions run in the same instructions
    if (np > 4 && np <= 64 && R % 4 == 0 && try_rowreduce_dma<16, OP, true>(prof, x, R, S, denom, cand, outf, st)) return;
    if (np <= 4 && R % 16 == 0) SL_ROWREDUCE(4, 8, true);
    if (np <= 8 && R % 8 == 0) SL_ROWREDUCE(8, 8, true);
    if (np <= 16 && R % 4 == 0) SL_ROWREDUCE(16, 8, true);
    if (np <= 32 && R % 2 == 0) SL_ROWREDUCE(32, 8, true);
    if (np <= 64) SL_ROWREDUCE(64, 8, true);
    SL_ROWREDUCE(64, 4, true);
  }
  // fast path B: short unaligned rows (e.g. 7x7 = 49 floats), >= 4 rows per wave-load
  if (S % 4 != 0 && need <= 16) {
    if (need <= 4 && R % 16 == 0) SL_ROWREDUCE(4, 8, false);
    if (need <= 8 && R % 8 == 0) SL_ROWREDUCE(8, 8, false);
    if (R % 4 == 0) SL_ROWREDUCE(16, 8, false);
#undef SL_ROWREDUCE
  }
  // longer unaligned rows whose tasks still fit an LDS-DMA batch (<= 4 KiB): two rows per task when S is even (S <= 512),
  // four otherwise (S <= 256: 13 x 13, 15 x 15 maps); the lanes walk a row's window in steps
  if (S % 4 != 0 && need > 16) {
    if (S % 2 == 0 && need > 32 && try_rowreduce_dma<32, OP, false, float, true>(prof, x, R, S, denom, cand, outf, st)) return;
    if (try_rowreduce_dma<16, OP, false, float, true>(prof, x, R, S, denom, cand, outf, st)) return;
  }
  // Longer unaligned fp32 rows stay on the round-1 kernel (0.58-0.72 of spec cold).  Tried and dropped in round 3: reading
  // each row from its own 4-byte-aligned start with unaligned 16-byte loads (legal on this part:
  // tools/native/unaligned_probe.hip) — 17 x 17 4.7 -> 4.4 TB/s, 27 x 27 5.8 -> 5.4, 111 x 111 4.95 -> 5.26.
  if (need <= 4) launch_rowreduce<4, 8, OP>(prof, x, R, S, denom, cand, outf, st);
  else if (need <= 8) launch_rowreduce<8, 8, OP>(prof, x, R, S, denom, cand, outf, st);
  else if (need <= 16) launch_rowreduce<16, 8, OP>(prof, x, R, S, denom, cand, outf, st);
  else if (need <= 32) launch_rowreduce<32, 8, OP>(prof, x, R, S, denom, cand, outf, st);
  else if (need <= 64) launch_rowreduce<64, 8, OP>(prof, x, R, S, denom, cand, outf, st);
  else launch_rowreduce<64, 4, OP>(prof, x, R, S, denom, cand, outf, st);
}

// colreduce2: the default component-contiguous kernel where 16-byte pieces are legal (see the kernel's header).
// loads per lane and block, two blocks in flight.  4 (tools/k2_lab.py): 72 VGPRs in fp32 / 100 in half precision (6-7 / 4 waves
// per SIMD); with 8 the half-precision kernels need 140 registers and fall from 5.3 to 4.0 TB/s, fp32 gains nothing
constexpr int kCol2Infl = 4;
template <typename T, int OP, int NW, int LPR>
void launch_colreduce2_as(ProfScope& prof, const MultiSrc& x, int64_t B, int T_, int64_t F, int64_t sb, int64_t st_, int t0, int t1,
                          float denom, int64_t tail_from, uint16_t* cand, float* outf, hipStream_t st) {
  constexpr int CW = LPR * (16 / (int)sizeof(T));
  int64_t blocks = B * ((F + CW - 1) / CW);
  const int64_t cap = (int64_t)num_cus() * 8;
  if (blocks > cap) blocks = cap;
  SL_LAUNCH(prof, (colreduce2_kernel<T, OP, NW, LPR, kCol2Infl>), dim3((unsigned)blocks), dim3(64 * NW), 0, st, x, B, T_, F, sb, st_, t0,
            t1, denom, tail_from, cand, outf);
}

// `x`: a table of L = B / x.per tensors of x.per batches each (L = 1: one tensor); B counts the batches of all of them
template <typename T, int OP>
bool launch_colreduce2(ProfScope& prof, const MultiSrc& x, int64_t B, int T_, int64_t F, int64_t sb, int64_t st_, int t0, int t1,
                       float denom, uint16_t* cand, float* outf, hipStream_t st) {
  const int forced_nw = (int)option(OPT_COLREDUCE_NW);  // sl_set_option("colreduce_nw", 4 / 8 / 16): tests walk every instance
  constexpr int EPP = 16 / (int)sizeof(T);
  const int64_t rows = t1 - t0;
  const int64_t L = B / x.per;
  bool aligned = true;
  for (int64_t l = 0; l < L; ++l) aligned = aligned && ((uintptr_t)x.ptr[l] & 15) == 0;
  if (!aligned || (F % EPP) != 0 || ((st_ * (int64_t)sizeof(T)) & 15) != 0 ||
      ((sb * (int64_t)sizeof(T)) & 15) != 0 || rows < 1)
    return false;
  // lanes per row: the widest chunk that wastes no lane, else the one that wastes least (ties: wider = fewer tasks)
  const int64_t pr = F / EPP;  // pieces per row
  int lpr = 64;
  double best = 0;
  for (int c : {64, 32, 16}) {
    const double util = (double)pr / (double)(((pr + c - 1) / c) * c);
    if (util > best + 1e-9) best = util, lpr = c;
  }
  const int64_t cw = (int64_t)lpr * EPP, nchunk = (F + cw - 1) / cw, cus = num_cus();
  const int64_t nt_min_bytes = nt_min_bytes_();
  const int64_t per_b = (int64_t)T_ * F * (int64_t)sizeof(T), all = B * per_b;
  // of a table of tensors only the LAST one was written a moment ago: the default-policy tail never reaches into the others
  const int64_t tail_bytes = (L > 1 && tail_bytes_() > x.per * per_b) ? x.per * per_b : tail_bytes_();
  int64_t tail_from = 0;
  if (all >= nt_min_bytes) tail_from = tail_bytes > 0 ? (all > tail_bytes ? (all - tail_bytes) / per_b * nchunk : 0) : INT64_MAX;
  // waves per task split the reduced axis; a wave instruction covers 64 / lpr rows, so short axes want few waves
  const int64_t inst_rows = rows * lpr / 64;  // wave instructions per task
  // 8 waves per task only below two tasks per CU.  tools/k2_lab.py (an elementwise producer, then K2) showed the fp32 kernel
  // 2-5 % faster with 8 waves up to four tasks per CU ((256, 197, 768): 6.09 -> 6.19 TB/s cold), but INSIDE the bench's leg
  // (behind a ViT block's GEMMs, tools/k2_leg_probe.py) the same shape reads 0.717 of spec with four waves and 0.671 with
  // eight; the half-precision kernels (100 registers, 16 waves per CU) lose 15 % with eight once there are two tasks per CU
  // ((256, 257, 1024) bf16: 6.07 -> 5.15 TB/s).  profiles/r04_k2_lab.txt
  // A table of L tensors takes the wave count ONE of its tensors would take alone: the waves split the reduced axis, so the
  // fp32 summation order of mean / absmean / sum follows NW, and a layer's candidates must not depend on whether its batch
  // was reduced alone (a collector's first batch, SEMANTICLENS_AMD_GROUP_LAYERS=0) or as a member of a group.
  const int64_t tasks_one = x.per * nchunk;
  int nw = 4;
  if (tasks_one * 2 < cus && inst_rows >= 128 && lpr == 64) nw = 16;
  else if (tasks_one < (sizeof(T) == 4 ? 2 : 1) * cus && inst_rows >= 64) nw = 8;
  if (forced_nw == 4 || forced_nw == 8 || (forced_nw == 16 && lpr == 64)) nw = forced_nw;
#define SL_COL2(NW_, LPR_) launch_colreduce2_as<T, OP, NW_, LPR_>(prof, x, B, T_, F, sb, st_, t0, t1, denom, tail_from, cand, outf, st)
#define SL_COL2_NW(LPR_)                 \
  do {                                   \
    if (nw == 16) SL_COL2(16, LPR_);     \
    else if (nw == 8) SL_COL2(8, LPR_);  \
    else SL_COL2(4, LPR_);               \
  } while (0)
  if (lpr == 64) {
    SL_COL2_NW(64);
  } else if (lpr == 32) {
    if (nw == 8) SL_COL2(8, 32);
    else SL_COL2(4, 32);
  } else {
    if (nw == 8) SL_COL2(8, 16);
    else SL_COL2(4, 16);
  }
#undef SL_COL2_NW
#undef SL_COL2
  return true;
}

template <typename T, int OP>
void launch_colreduce(ProfScope& prof, const T* x, int64_t B, int T_, int64_t F, int64_t sb, int64_t st_, int t0, int t1,
                      float denom, uint16_t* cand, float* outf, hipStream_t st) {
  MultiSrc one;
  one.ptr[0] = x;
  one.per = B;
  if (launch_colreduce2<T, OP>(prof, one, B, T_, F, sb, st_, t0, t1, denom, cand, outf, st)) return;
  int64_t blocks = B * ((F + 255) / 256);
  const int64_t cap = (int64_t)num_cus() * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  // cache policy (top of this file), in tasks = (b, chunk) pairs, b-major like the bytes
  const int64_t nt_min_bytes = nt_min_bytes_(), tail_bytes = tail_bytes_();
  const int64_t nchunk = (F + 255) / 256, per_b = (int64_t)T_ * F * (int64_t)sizeof(T), bytes = B * per_b;
  int64_t tail_from = 0;  // everything with the default policy
  if (bytes >= nt_min_bytes) tail_from = tail_bytes > 0 ? (bytes > tail_bytes ? (bytes - tail_bytes) / per_b * nchunk : 0) : INT64_MAX;
  // waves per task (they split the reduced axis): enough of them that a CU holds ~24 waves with 8 loads in flight each.
  // (B, 197, 768) at B = 256 is 768 tasks: 4-wave workgroups put 12 waves on a CU (5.4 TB/s cold), 8-wave ones 24.
  const int forced_nw = (int)option(OPT_COLREDUCE_NW);
  const int64_t tasks = B * nchunk, rows = t1 - t0, cus = num_cus();
  int nw = 4;
  // measured cold (tools/reduce_dtype_bench.py, SL_COLREDUCE_NW = 4 / 8 / 16): (256, 197, 768) fp32 5.52 / 5.63 / 5.46 TB/s,
  // (48, 729, 1152) fp32 5.46 / 5.73 / 5.45 and fp16 3.7 / 5.45 / 5.03, channels_last 14 x 14 fp16 5.74 / 5.94 / 4.0;
  // short reduced axes (7 x 7 = 49 rows, 50 tokens) lose with more than four waves
  // In the pipeline (input just written, bench leg `tokens_collect`, seven runs) the 768-task shape reads 0.60-0.69 of spec with
  // four waves against 0.55-0.65 with eight, so eight-wave workgroups are kept for grids below two tasks per CU.
  if (tasks * 2 < cus && rows >= 128) nw = 16;
  else if (tasks < 2 * cus && rows >= 64) nw = 8;
  if (forced_nw == 4 || forced_nw == 8 || forced_nw == 16) nw = forced_nw;
  if (nw == 16)
    SL_LAUNCH(prof, (colreduce_kernel<T, OP, 16>), dim3((unsigned)blocks), dim3(1024), 0, st, x, B, T_, F, sb, st_, t0, t1, denom,
              tail_from, cand, outf);
  else if (nw == 8)
    SL_LAUNCH(prof, (colreduce_kernel<T, OP, 8>), dim3((unsigned)blocks), dim3(512), 0, st, x, B, T_, F, sb, st_, t0, t1, denom,
              tail_from, cand, outf);
  else
    SL_LAUNCH(prof, (colreduce_kernel<T, OP, 4>), dim3((unsigned)blocks), dim3(256), 0, st, x, B, T_, F, sb, st_, t0, t1, denom,
              tail_from, cand, outf);
}

template <typename T, int G, int U, int J, int OP, bool ALIGNED>
void launch_rowreduce_h(ProfScope& prof, const T* x, int64_t R, int S, float denom, uint16_t* cand, float* outf, hipStream_t st) {
  constexpr int RPW = kWave / G;
  const int64_t nsets = (R + RPW - 1) / RPW;
  const int64_t nbatch = (nsets + U - 1) / U;
  int64_t blocks = (nbatch + 3) / 4;
  const int64_t cap = (int64_t)num_cus() * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  const int64_t nt_min_bytes = nt_min_bytes_(), tail_bytes = tail_bytes_();
  const int64_t batch_bytes = (int64_t)U * RPW * S * 2, bytes = R * (int64_t)S * 2;
  int64_t tail_from = 0;
  if (bytes >= nt_min_bytes) tail_from = tail_bytes > 0 ? (bytes > tail_bytes ? (bytes - tail_bytes) / batch_bytes : 0) : INT64_MAX;
  SL_LAUNCH(prof, (rowreduce_h_kernel<T, G, U, J, OP, ALIGNED>), dim3((unsigned)blocks), dim3(256), 0, st, x, R, S, denom, tail_from,
            cand, outf);
}

// G = lanes per row: the smallest power of two that covers the pieces of a row's window (at most 64: longer rows loop)
template <typename T, int OP>
void dispatch_rowreduce_h(ProfScope& prof, const T* x, int64_t R, int S, float denom, uint16_t* cand, float* outf, hipStream_t st) {
  const bool al = S % 8 == 0;
  const int np = al ? S / 8 : (S + 14) / 8;
  // LDS-DMA ring kernel first (rowreduce_dma_kernel<T>: the fp32 kernel's feed with 8-element pieces); it takes inputs of
  // >= 8 MB whose tasks (64 / G rows) are whole 16-byte pieces
  if (al) {
    if (np > 4 && np <= 64 && try_rowreduce_dma<16, OP, true>(prof, x, R, S, denom, cand, outf, st)) return;
    if (np <= 4 && try_rowreduce_dma<4, OP, true>(prof, x, R, S, denom, cand, outf, st)) return;
    if (np > 64 && try_rowreduce_dma<64, OP, true>(prof, x, R, S, denom, cand, outf, st)) return;
  } else {
    if (np <= 4 && try_rowreduce_dma<4, OP, false>(prof, x, R, S, denom, cand, outf, st)) return;
    if (np <= 8 && try_rowreduce_dma<8, OP, false>(prof, x, R, S, denom, cand, outf, st)) return;
    if (np <= 16 && try_rowreduce_dma<16, OP, false>(prof, x, R, S, denom, cand, outf, st)) return;
    if (np <= 32 && try_rowreduce_dma<32, OP, false>(prof, x, R, S, denom, cand, outf, st)) return;
    // windows longer than a task's lanes, or rows whose short tasks are not whole pieces: walk the window in steps with the
    // fewest rows per task that make it whole (2 rows when S % 4 == 0: S <= 1024; 4 when S is even: S <= 512; else 8: S <= 256)
    if (S % 4 == 0 && try_rowreduce_dma<32, OP, false, T, true>(prof, x, R, S, denom, cand, outf, st)) return;
    if (S % 2 == 0 && try_rowreduce_dma<16, OP, false, T, true>(prof, x, R, S, denom, cand, outf, st)) return;
    if (try_rowreduce_dma<8, OP, false, T, true>(prof, x, R, S, denom, cand, outf, st)) return;
  }
#define SL_ROWH(G, U, J)                                                                          \
  do {                                                                                            \
    if (al) launch_rowreduce_h<T, G, U, J, OP, true>(prof, x, R, S, denom, cand, outf, st);       \
    else launch_rowreduce_h<T, G, U, J, OP, false>(prof, x, R, S, denom, cand, outf, st);         \
    return;                                                                                       \
  } while (0)
  if (np <= 1) SL_ROWH(1, 4, 1);
  if (np <= 2) SL_ROWH(2, 4, 1);
  if (np <= 4) SL_ROWH(4, 4, 1);
  if (np <= 8) SL_ROWH(8, 4, 1);
  if (np <= 16) SL_ROWH(16, 4, 1);
  if (np <= 32) SL_ROWH(32, 4, 1);
  if (np <= 64) SL_ROWH(64, 4, 1);
  SL_ROWH(64, 2, 2);
#undef SL_ROWH
}

template <typename T, int OP>
void launch_generic(ProfScope& prof, const void* x, int64_t B, int64_t C, int64_t S, int64_t sb, int64_t sc, int64_t ss, int64_t s0,
                    int64_t s1, float denom, uint16_t* cand, float* outf, hipStream_t st) {
  int64_t blocks = (B * C + 255) / 256;
  const int64_t cap = (int64_t)num_cus() * 16;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  SL_LAUNCH(prof, (generic_reduce_kernel<T, OP>), dim3((unsigned)blocks), dim3(256), 0, st, x, B, C, S, sb, sc, ss, s0, s1,
            denom, cand, outf);
}

template <int OP>
void dispatch_generic(ProfScope& prof, const void* x, int dtype, int64_t B, int64_t C, int64_t S, int64_t sb, int64_t sc, int64_t ss,
                      int64_t s0, int64_t s1, float denom, uint16_t* cand, float* outf, hipStream_t st) {
  if (dtype == SL_F32) launch_generic<float, OP>(prof, x, B, C, S, sb, sc, ss, s0, s1, denom, cand, outf, st);
  else if (dtype == SL_F16) launch_generic<_Float16, OP>(prof, x, B, C, S, sb, sc, ss, s0, s1, denom, cand, outf, st);
  else launch_generic<uint16_t, OP>(prof, x, B, C, S, sb, sc, ss, s0, s1, denom, cand, outf, st);
}

// (B, C, S) with strides -> (B, C): reduce over s in [s0, s1).  Picks the fastest legal path.
template <int OP>
int reduce_dispatch(ProfScope& prof, const void* x, int dtype, int64_t B, int64_t C, int64_t S, int64_t sb, int64_t sc, int64_t ss,
                    int64_t s0, int64_t s1, uint16_t* cand, float* outf, hipStream_t st, bool plain_sum = false) {
  const float denom = plain_sum ? 1.f : (float)(s1 - s0);  // x / 1 is exact: the same kernels give sums
  const bool aligned = ((uintptr_t)x & 15) == 0;
  const bool full = (s0 == 0 && s1 == S);
  if (dtype == SL_F32 && aligned && full && ss == 1 && sc == S && sb == C * S && S < (1 << 28)) {
    dispatch_rowreduce<OP>(prof, (const float*)x, B * C, (int)S, denom, cand, outf, st);
  } else if (dtype == SL_F32 && aligned && sc == 1 && (C % 4) == 0 && (ss % 4) == 0 && (sb % 4) == 0 &&
             S < (1 << 30)) {
    launch_colreduce<float, OP>(prof, (const float*)x, B, (int)S, C, sb, ss, (int)s0, (int)s1, denom, cand, outf, st);
  } else if (dtype != SL_F32 && aligned && full && ss == 1 && sc == S && sb == C * S && S > 0 && S < (1 << 27)) {
    // fp16 / bf16, NCHW-contiguous rows
    if (dtype == SL_F16) dispatch_rowreduce_h<_Float16, OP>(prof, (const _Float16*)x, B * C, (int)S, denom, cand, outf, st);
    else dispatch_rowreduce_h<uint16_t, OP>(prof, (const uint16_t*)x, B * C, (int)S, denom, cand, outf, st);
  } else if (dtype != SL_F32 && ((uintptr_t)x & 7) == 0 && sc == 1 && (C % 4) == 0 && (ss % 4) == 0 && (sb % 4) == 0 &&
             S < (1 << 30)) {
    // fp16 / bf16, component axis contiguous (channels_last, tokens): 8-byte loads of four components
    if (dtype == SL_F16)
      launch_colreduce<_Float16, OP>(prof, (const _Float16*)x, B, (int)S, C, sb, ss, (int)s0, (int)s1, denom, cand, outf, st);
    else
      launch_colreduce<uint16_t, OP>(prof, (const uint16_t*)x, B, (int)S, C, sb, ss, (int)s0, (int)s1, denom, cand, outf, st);
  } else {
    dispatch_generic<OP>(prof, x, dtype, B, C, S, sb, sc, ss, s0, s1, denom, cand, outf, st);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "reduce kernel launch");
  return 0;
}

int dtype_size(int dtype) { return dtype == SL_F32 ? 4 : 2; }

// L same-shape (B, C, S) activations -> (L, B, C) candidates.  ONE launch when the component axis is contiguous (tokens,
// channels_last: colreduce2 over a table of tensors), tensor by tensor otherwise — the same values either way.
template <int OP>
int reduce_dispatch_multi(const void* const* xs, int L, int dtype, int64_t B, int64_t C, int64_t S, int64_t sb, int64_t sc, int64_t ss,
                          int64_t s0, int64_t s1, uint16_t* cand, hipStream_t st, bool plain_sum = false) {
  const float denom = plain_sum ? 1.f : (float)(s1 - s0);
  const double work = (double)B * C * (s1 - s0) * dtype_size(dtype);
  for (int l0 = 0; l0 < L; l0 += kMaxReduceSources) {
    const int n = L - l0 < kMaxReduceSources ? L - l0 : kMaxReduceSources;
    uint16_t* out = cand + (int64_t)l0 * B * C;
    bool done = false;
    if (n > 1 && sc == 1 && S < (1 << 30) && (ss % 4) == 0 && (sb % 4) == 0 && (C % 4) == 0) {
      MultiSrc src;
      for (int i = 0; i < n; ++i) src.ptr[i] = xs[l0 + i];
      src.per = B;
      ProfScope prof(SL_PROF_REDUCE, st, work * n);
      if (dtype == SL_F32)
        done = launch_colreduce2<float, OP>(prof, src, n * B, (int)S, C, sb, ss, (int)s0, (int)s1, denom, out, nullptr, st);
      else if (dtype == SL_F16)
        done = launch_colreduce2<_Float16, OP>(prof, src, n * B, (int)S, C, sb, ss, (int)s0, (int)s1, denom, out, nullptr, st);
      else
        done = launch_colreduce2<uint16_t, OP>(prof, src, n * B, (int)S, C, sb, ss, (int)s0, (int)s1, denom, out, nullptr, st);
      if (done) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "reduce kernel launch");
      }
    }
    if (!done) {
      for (int i = 0; i < n; ++i) {
        ProfScope prof(SL_PROF_REDUCE, st, work);
        const int rc = reduce_dispatch<OP>(prof, xs[l0 + i], dtype, B, C, S, sb, sc, ss, s0, s1, out + (int64_t)i * B * C, nullptr, st, plain_sum);
        if (rc) return rc;
      }
    }
  }
  return 0;
}

void set_reduce_policy(int64_t nt_min_bytes, int64_t tail_bytes) {
  g_nt_min_bytes = nt_min_bytes;
  g_tail_bytes = tail_bytes;
}

}  // namespace
}  // namespace sl

using namespace sl;

SL_API int sl_reduce_conv(const void* d_act, int dtype, int64_t B, int64_t C, int64_t S, int64_t sb, int64_t sc,
                          int64_t ss, int agg, uint16_t* d_cand_bf16, float* d_out_f32, void* stream) {
  SL_REQUIRE(d_act || B * C * S == 0, "sl_reduce_conv: null activation");
  SL_REQUIRE(dtype >= SL_F32 && dtype <= SL_BF16, "sl_reduce_conv: bad dtype %d", dtype);
  SL_REQUIRE(B >= 0 && C >= 0 && S >= 0, "sl_reduce_conv: negative shape");
  SL_REQUIRE(agg == SL_CONV_MAX || agg == SL_CONV_MEAN || agg == SL_CONV_SUM, "sl_reduce_conv: bad agg %d", agg);
  SL_REQUIRE(d_cand_bf16 || d_out_f32, "sl_reduce_conv: no output");
  if (B * C == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof(SL_PROF_REDUCE, st, (double)B * C * S * dtype_size(dtype));
  if (agg == SL_CONV_MAX) return reduce_dispatch<OP_MAX>(prof, d_act, dtype, B, C, S, sb, sc, ss, 0, S, d_cand_bf16, d_out_f32, st);
  return reduce_dispatch<OP_SUM>(prof, d_act, dtype, B, C, S, sb, sc, ss, 0, S, d_cand_bf16, d_out_f32, st, agg == SL_CONV_SUM);
}

SL_API int sl_reduce_conv_multi(const void* const* h_d_acts, int L, int dtype, int64_t B, int64_t C, int64_t S, int64_t sb, int64_t sc,
                                int64_t ss, int agg, uint16_t* d_cand_bf16, void* stream) {
  SL_REQUIRE(L >= 0 && B >= 0 && C >= 0 && S >= 0, "sl_reduce_conv_multi: negative shape");
  SL_REQUIRE(dtype >= SL_F32 && dtype <= SL_BF16, "sl_reduce_conv_multi: bad dtype %d", dtype);
  SL_REQUIRE(agg == SL_CONV_MAX || agg == SL_CONV_MEAN || agg == SL_CONV_SUM, "sl_reduce_conv_multi: bad agg %d", agg);
  if (L == 0 || B * C == 0) return 0;
  SL_REQUIRE(h_d_acts && d_cand_bf16, "sl_reduce_conv_multi: null pointer");
  for (int l = 0; l < L; ++l) SL_REQUIRE(h_d_acts[l] || S == 0, "sl_reduce_conv_multi: null activation");
  hipStream_t st = (hipStream_t)stream;
  if (agg == SL_CONV_MAX) return reduce_dispatch_multi<OP_MAX>(h_d_acts, L, dtype, B, C, S, sb, sc, ss, 0, S, d_cand_bf16, st);
  return reduce_dispatch_multi<OP_SUM>(h_d_acts, L, dtype, B, C, S, sb, sc, ss, 0, S, d_cand_bf16, st, agg == SL_CONV_SUM);
}

SL_API int sl_reduce_tokens_multi(const void* const* h_d_acts, int L, int dtype, int64_t B, int64_t T, int64_t F, int64_t sb, int64_t st_,
                                  int64_t sf, int agg, int64_t pos, uint16_t* d_cand_bf16, void* stream) {
  SL_REQUIRE(L >= 0 && B >= 0 && T >= 0 && F >= 0, "sl_reduce_tokens_multi: negative shape");
  SL_REQUIRE(dtype >= SL_F32 && dtype <= SL_BF16, "sl_reduce_tokens_multi: bad dtype %d", dtype);
  SL_REQUIRE(agg >= SL_TOK_MEAN && agg <= SL_TOK_TOKEN, "sl_reduce_tokens_multi: bad agg %d", agg);
  if (L == 0 || B * F == 0) return 0;
  SL_REQUIRE(h_d_acts && d_cand_bf16, "sl_reduce_tokens_multi: null pointer");
  for (int l = 0; l < L; ++l) SL_REQUIRE(h_d_acts[l] || T == 0, "sl_reduce_tokens_multi: null activation");
  hipStream_t st = (hipStream_t)stream;
  int64_t t0 = 0, t1 = T;
  if (agg == SL_TOK_TOKEN) {
    int64_t p = pos < 0 ? pos + T : pos;
    SL_REQUIRE(p >= 0 && p < T, "sl_reduce_tokens: token position %lld out of range for T=%lld", (long long)pos, (long long)T);
    t0 = p;
    t1 = p + 1;
  }
  switch (agg) {
    case SL_TOK_MEAN:
      return reduce_dispatch_multi<OP_SUM>(h_d_acts, L, dtype, B, F, T, sb, sf, st_, t0, t1, d_cand_bf16, st);
    case SL_TOK_ABSMEAN:
      return reduce_dispatch_multi<OP_ABSSUM>(h_d_acts, L, dtype, B, F, T, sb, sf, st_, t0, t1, d_cand_bf16, st);
    case SL_TOK_ABSMAX:
      return reduce_dispatch_multi<OP_ABSMAX>(h_d_acts, L, dtype, B, F, T, sb, sf, st_, t0, t1, d_cand_bf16, st);
    default:
      return reduce_dispatch_multi<OP_MAX>(h_d_acts, L, dtype, B, F, T, sb, sf, st_, t0, t1, d_cand_bf16, st);
  }
}

// x (B,C) fp32 in place: x[b][:] /= (sum_c |x[b][c]| + eps)   (crp ChannelConcept.reference_sampling, abs_norm)
namespace sl {
namespace {
__global__ __launch_bounds__(256) void abs_norm_rows_kernel(float* __restrict__ x, int64_t B, int64_t C, float eps) {
  __shared__ float s_part[4];
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    float* row = x + b * C;
    float s = 0.f;
    for (int64_t c = threadIdx.x; c < C; c += 256) s += __builtin_fabsf(row[c]);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = s;
    __syncthreads();
    const float tot = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]) + eps;
    for (int64_t c = threadIdx.x; c < C; c += 256) row[c] = row[c] / tot;
  }
}
}  // namespace
}  // namespace sl

SL_API int sl_abs_norm_rows(float* d_x, int64_t B, int64_t C, float eps, void* stream) {
  SL_REQUIRE(B >= 0 && C >= 0, "sl_abs_norm_rows: negative shape");
  if (B * C == 0) return 0;
  SL_REQUIRE(d_x, "sl_abs_norm_rows: null pointer");
  int64_t blocks = B < (int64_t)num_cus() * 8 ? B : (int64_t)num_cus() * 8;
  hipLaunchKernelGGL(abs_norm_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_x, B, C, eps);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

SL_API int sl_set_reduce_policy(int64_t nt_min_bytes, int64_t tail_bytes) {
  // negative = back to the environment / built-in defaults
  set_reduce_policy(nt_min_bytes < 0 ? -1 : nt_min_bytes, tail_bytes < 0 ? -1 : tail_bytes);
  return 0;
}

SL_API int sl_reduce_tokens(const void* d_act, int dtype, int64_t B, int64_t T, int64_t F, int64_t sb, int64_t st_,
                            int64_t sf, int agg, int64_t pos, uint16_t* d_cand_bf16, float* d_out_f32,
                            void* stream) {
  SL_REQUIRE(d_act || B * T * F == 0, "sl_reduce_tokens: null activation");
  SL_REQUIRE(dtype >= SL_F32 && dtype <= SL_BF16, "sl_reduce_tokens: bad dtype %d", dtype);
  SL_REQUIRE(B >= 0 && T >= 0 && F >= 0, "sl_reduce_tokens: negative shape");
  SL_REQUIRE(agg >= SL_TOK_MEAN && agg <= SL_TOK_TOKEN, "sl_reduce_tokens: bad agg %d", agg);
  SL_REQUIRE(d_cand_bf16 || d_out_f32, "sl_reduce_tokens: no output");
  if (B * F == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  // (B,T,F) reduced over T == (B, C=F, S=T) with sc = sf, ss = st
  int64_t t0 = 0, t1 = T;
  if (agg == SL_TOK_TOKEN) {
    int64_t p = pos < 0 ? pos + T : pos;
    SL_REQUIRE(p >= 0 && p < T, "sl_reduce_tokens: token position %lld out of range for T=%lld", (long long)pos,
               (long long)T);
    t0 = p;
    t1 = p + 1;
  }
  ProfScope prof(SL_PROF_REDUCE, st, (double)B * (t1 - t0) * F * dtype_size(dtype));
  switch (agg) {
    case SL_TOK_MEAN:
      return reduce_dispatch<OP_SUM>(prof, d_act, dtype, B, F, T, sb, sf, st_, t0, t1, d_cand_bf16, d_out_f32, st);
    case SL_TOK_ABSMEAN:
      return reduce_dispatch<OP_ABSSUM>(prof, d_act, dtype, B, F, T, sb, sf, st_, t0, t1, d_cand_bf16, d_out_f32, st);
    case SL_TOK_ABSMAX:
      return reduce_dispatch<OP_ABSMAX>(prof, d_act, dtype, B, F, T, sb, sf, st_, t0, t1, d_cand_bf16, d_out_f32, st);
    default:  // max, and the single-token pick (max over one element is the element itself)
      return reduce_dispatch<OP_MAX>(prof, d_act, dtype, B, F, T, sb, sf, st_, t0, t1, d_cand_bf16, d_out_f32, st);
  }
}
