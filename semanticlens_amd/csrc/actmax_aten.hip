// K3, SL_TIES_ATEN — ActMax.update with torch.topk's exact CPU tie order
// (activation_caching.py:137-141; selection order restated in aten_topk_order.hpp / aten_topk_wave.hpp).
//
// Round 5: ONE WAVEFRONT PER COMPONENT.  The row [state | batch] of n = k + B packed (order-key, position) words lives in
// LDS; libstdc++'s introselect / introsort are followed step by step, but each Hoare partition is evaluated by the whole
// wave at once from its closed form (ballots give every lane's rank in the left-stop / right-stop lists, the swaps are
// independent) and every insertion sort is a stable rank sort (aten_topk_wave.hpp has the derivation and the host check
// against libstdc++).  A partition step costs a handful of LDS round trips instead of one dependent LDS access per element:
// C = 2048, B = 256, k = 20 went from 169 us (one lane per row, round 4) to the figure in profiles/r05_k3_*.  The heap
// fallbacks (depth limit, torch's `k * 64 <= n` partial_sort branch) run sequentially on lane 0 over the same LDS row; rows
// too long for the LDS take the round-4 kernel (one lane per row, kept below).  This mode exists for bit-identity with the
// reference; SL_TIES_TOTAL is the shard-invariant path.
#include "aten_topk_order.hpp"
#include "aten_topk_wave.hpp"
#include "common.hpp"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace sl {
namespace {

struct LdsColumn {
  uint32_t* base;
  int stride;
  __device__ inline uint32_t& operator[](int i) const { return base[i * stride]; }
};

__global__ __launch_bounds__(64) void actmax_update_aten_kernel(uint16_t* __restrict__ vals,
                                                                 int64_t* __restrict__ ids, int64_t C, int k,
                                                                 const uint16_t* __restrict__ cand,
                                                                 const int64_t* __restrict__ sample_ids,
                                                                 int64_t id_base, int B, int rpb,
                                                                 int64_t* __restrict__ ws_ids,
                                                                 uint16_t* __restrict__ ws_vals) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int r = threadIdx.x;
  const int64_t c = (int64_t)blockIdx.x * rpb + r;
  if (r >= rpb || c >= C) return;
  LdsColumn a{reinterpret_cast<uint32_t*>(smem) + r, rpb};
  const int n = k + B;
  const int64_t so = c * k;
  // all_acts = cat([state, batch_acts]) — activation_caching.py:137
  for (int j = 0; j < k; ++j) a[j] = (bf16_order_key(vals[so + j]) << 16) | (uint32_t)j;
  for (int b = 0; b < B; ++b) a[k + b] = (bf16_order_key(cand[(int64_t)b * C + c]) << 16) | (uint32_t)(k + b);
  aten_order::topk_order(a, n, k);  // :140
  // gather values / ids through the selected positions — :140-141
  for (int j = 0; j < k; ++j) {
    const int pos = (int)(a[j] & 0xFFFFu);
    uint16_t v;
    int64_t i;
    if (pos < k) {
      v = vals[so + pos];
      i = ids[so + pos];
    } else {
      v = cand[(int64_t)(pos - k) * C + c];
      i = sample_ids ? sample_ids[pos - k] : id_base + (pos - k);
    }
    ws_vals[so + j] = v;
    ws_ids[so + j] = i;
  }
  for (int j = 0; j < k; ++j) {
    vals[so + j] = ws_vals[so + j];
    ids[so + j] = ws_ids[so + j];
  }
}


// ---- one wavefront per row -----------------------------------------------------------------------------------------------
__device__ inline void wave_sync() {  // LDS traffic of ONE wave executes in order; this only stops the compiler reordering it
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// move_median_to_first(result, x, y, z): every lane evaluates it, lane 0 writes the swap
__device__ inline void wave_median_to_first(uint32_t* A, int result, int x, int y, int z, int lane) {
  using aten_order::cmp;
  const uint32_t ax = A[x], ay = A[y], az = A[z], ar = A[result];
  int sel;
  if (cmp(ax, ay)) sel = cmp(ay, az) ? y : (cmp(ax, az) ? z : x);
  else sel = cmp(ax, az) ? x : (cmp(ay, az) ? z : y);
  const uint32_t vs = sel == x ? ax : (sel == y ? ay : az);
  wave_sync();
  if (lane == 0) {
    A[result] = vs;
    A[sel] = ar;
  }
  wave_sync();
}

// aten_order::lists::partition with the lists built by ballots: tL / tR hold the left-stop / right-stop positions ascending
__device__ inline int wave_partition(uint32_t* A, uint16_t* tL, uint16_t* tR, int lo, int hi, uint32_t pv, int lane) {
  const uint32_t pk = pv >> 16;
  const uint64_t below = (1ull << lane) - 1ull;
  int nL = 0, nR = 0;
  for (int p0 = lo; p0 < hi; p0 += 64) {
    const int p = p0 + lane;
    const bool valid = p < hi;
    const uint32_t key = valid ? (A[p] >> 16) : 0u;
    const bool inL = valid && key <= pk, inR = valid && key >= pk;
    const uint64_t bL = __ballot(inL), bR = __ballot(inR);
    if (inL) tL[nL + __popcll(bL & below)] = (uint16_t)p;
    if (inR) tR[nR + __popcll(bR & below)] = (uint16_t)p;
    nL += __popcll(bL);
    nR += __popcll(bR);
  }
  wave_sync();
  const int mn = nL < nR ? nL : nR;
  int m = 0;  // swaps: L[j] < R[j] is monotone in j
  for (int j0 = 0; j0 < mn; j0 += 64) {
    const int j = j0 + lane;
    const bool ok = j < mn && tL[j < mn ? j : 0] < tR[j < mn ? nR - 1 - j : 0];
    const uint64_t bb = __ballot(ok);
    m += __popcll(bb);
    if (bb != ~0ull) break;
  }
  for (int j0 = 0; j0 < m; j0 += 64) {
    const int j = j0 + lane;
    if (j < m) {
      const int pl = tL[j], pr = tR[nR - 1 - j];
      const uint32_t x = A[pl], y = A[pr];
      A[pl] = y;
      A[pr] = x;
    }
  }
  const int big = 0x7FFFFFFF;
  const int cutL = m < nL ? (int)tL[m] : big;
  const int cutR = m > 0 ? (int)tR[nR - m] : big;
  wave_sync();
  return uni(cutL < cutR ? cutL : cutR);
}

// stable descending rank sort of A[first, last) through T (>= last - first words)
__device__ inline void wave_stable_sort(uint32_t* A, uint32_t* T, int first, int last, int lane) {
  if (last - first < 2) return;
  for (int i0 = first; i0 < last; i0 += 64) {
    const int i = i0 + lane;
    const bool valid = i < last;
    const uint32_t vi = A[valid ? i : first];
    const uint32_t ki = vi >> 16;
    int rank = 0;
    for (int j = first; j < last; ++j) {
      const uint32_t kj = A[j] >> 16;  // one address for the whole wave: an LDS broadcast
      rank += (kj > ki) || (kj == ki && j < i);
    }
    if (valid) T[rank] = vi;
  }
  wave_sync();
  for (int i0 = first; i0 < last; i0 += 64) {
    const int i = i0 + lane;
    if (i < last) A[i] = T[i - first];
  }
  wave_sync();
}

// aten_order::topk_order on the LDS row A[0, n); T: n words of scratch, stk: 96 words
__device__ inline void wave_topk_row(uint32_t* A, uint32_t* T, uint32_t* stk, int n, int k, int lane) {
  using namespace aten_order;
  uint16_t* tL = reinterpret_cast<uint16_t*>(T);
  uint16_t* tR = tL + n;
  if ((int64_t)k * 64 <= (int64_t)n) {  // torch's partial_sort branch: heaps, sequential
    if (lane == 0) partial_sort(A, 0, k, n);
    wave_sync();
    return;
  }
  // std::nth_element(A, A + k - 1, A + n)
  int first = 0, last = n;
  const int nth = k - 1;
  int depth = floor_log2(n) * 2;
  bool done = false;
  while (last - first > 3) {
    if (depth == 0) {
      if (lane == 0) {
        heap_select(A, first, nth + 1, last);
        swap_at(A, first, nth);
      }
      wave_sync();
      done = true;
      break;
    }
    --depth;
    wave_median_to_first(A, first, first + 1, first + (last - first) / 2, last - 1, lane);
    const int cut = wave_partition(A, tL, tR, first + 1, last, A[first], lane);
    if (cut <= nth) first = cut;
    else last = cut;
  }
  if (!done) wave_stable_sort(A, T, first, last, lane);  // __insertion_sort of <= 3 elements
  // std::sort(A, A + k - 1): introsort loop over an explicit stack (disjoint sub-ranges: their order does not matter)
  const int s_last = k - 1;
  if (s_last <= 0) return;
  constexpr int kThreshold = 16;
  int sp = 0;
  if (lane == 0) stk[0] = 0u, stk[1] = (uint32_t)s_last | ((uint32_t)(floor_log2(s_last) * 2) << 16);
  wave_sync();
  sp = 1;
  while (sp > 0) {
    --sp;
    int f = uni((int)stk[2 * sp]);
    const uint32_t ld = stk[2 * sp + 1];
    int l = uni((int)(ld & 0xFFFFu)), d = uni((int)(ld >> 16));
    while (l - f > kThreshold) {
      if (d == 0) {
        if (lane == 0) partial_sort(A, f, l, l);
        wave_sync();
        break;
      }
      --d;
      wave_median_to_first(A, f, f + 1, f + (l - f) / 2, l - 1, lane);
      const int cut = wave_partition(A, tL, tR, f + 1, l, A[f], lane);
      if (sp < 48) {
        if (lane == 0) stk[2 * sp] = (uint32_t)cut, stk[2 * sp + 1] = (uint32_t)l | ((uint32_t)d << 16);
        wave_sync();
        ++sp;
      }
      l = cut;
    }
  }
  wave_stable_sort(A, T, 0, s_last, lane);  // __final_insertion_sort
}

// LDS bytes of one row: A and T (n words each), the original bf16 bits (n halves), the old ids (k), the stack
__host__ __device__ inline size_t wave_row_bytes(int n, int k) {
  return (size_t)n * 8 + (((size_t)n * 2 + 7) & ~(size_t)7) + (size_t)k * 8 + 96 * 4;
}

// The states of up to kMaxK3Layers layers (their own component counts, one k) updated by ONE launch (round 5: every hooked layer
// of a forward pass — the L identical blocks of a transformer, or ResNet-50's layer2-4 — merged once per batch; a single layer is
// a table of one).  "Virtual" component v of the sum of the C_l belongs to the layer with cstart[l] <= v < cstart[l + 1]; each
// layer has its own (B, C_l) candidate matrix.  The kernel's time is a chain of LDS round trips per row, not a function of the
// number of rows (17 us at 512 rows, 19.8 at 2 048, 39.6 at 9 216), so three launches of 18-20 us become one of ~24.
constexpr int kMaxK3Layers = 32;
struct K3States {
  uint16_t* vals[kMaxK3Layers];
  int64_t* ids[kMaxK3Layers];
  const uint16_t* cand[kMaxK3Layers];
  int64_t id_base[kMaxK3Layers];
  int64_t cstart[kMaxK3Layers + 1];
};

template <int ROWS>
__global__ __launch_bounds__(64 * ROWS) void actmax_update_aten_wave_kernel(K3States tab, int L, int64_t Ctot, int k,
                                                                            const int64_t* __restrict__ sample_ids, int B) {
  extern __shared__ __align__(16) unsigned char smem[];
  // where each of this workgroup's rows lives (the table is indexed with the uniform loop counter only: scalar loads)
  __shared__ const uint16_t* s_cand[ROWS];
  __shared__ uint16_t* s_vals[ROWS];
  __shared__ int64_t* s_ids[ROWS];
  __shared__ int64_t s_base[ROWS], s_c[ROWS], s_C[ROWS];
  const int n = k + B;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t c0 = (int64_t)blockIdx.x * ROWS;
  if (threadIdx.x < ROWS) {
    const int64_t v = c0 + threadIdx.x;
    for (int l = 0; l < L; ++l) {
      if (v >= tab.cstart[l] && v < tab.cstart[l + 1]) {
        s_cand[threadIdx.x] = tab.cand[l];
        s_vals[threadIdx.x] = tab.vals[l];
        s_ids[threadIdx.x] = tab.ids[l];
        s_base[threadIdx.x] = tab.id_base[l];
        s_c[threadIdx.x] = v - tab.cstart[l];
        s_C[threadIdx.x] = tab.cstart[l + 1] - tab.cstart[l];
      }
    }
  }
  __syncthreads();
  const size_t rb = wave_row_bytes(n, k);
  auto rowA = [&](int r) { return reinterpret_cast<uint32_t*>(smem + r * rb); };
  auto rowT = [&](int r) { return rowA(r) + n; };
  auto rowIds = [&](int r) { return reinterpret_cast<int64_t*>(smem + r * rb + (size_t)n * 8); };
  auto rowStk = [&](int r) { return reinterpret_cast<uint32_t*>(rowIds(r) + k); };
  auto rowRaw = [&](int r) { return reinterpret_cast<uint16_t*>(rowStk(r) + 96); };
  // all_acts = cat([state, batch_acts]) — activation_caching.py:137.  The (B, C) candidates are transposed on the way in:
  // consecutive threads take the ROWS adjacent components of one sample (ROWS x 2 contiguous bytes within a layer)
  for (int idx = threadIdx.x; idx < B * ROWS; idx += 64 * ROWS) {
    const int b = idx / ROWS, r = idx % ROWS;
    if (c0 + r < Ctot) {
      const uint16_t h = s_cand[r][(int64_t)b * s_C[r] + s_c[r]];
      rowA(r)[k + b] = (bf16_order_key(h) << 16) | (uint32_t)(k + b);
      rowRaw(r)[k + b] = h;
    }
  }
  for (int idx = threadIdx.x; idx < k * ROWS; idx += 64 * ROWS) {
    const int r = idx / k, j = idx % k;
    if (c0 + r < Ctot) {
      const int64_t so = s_c[r] * k + j;
      const uint16_t h = s_vals[r][so];
      rowA(r)[j] = (bf16_order_key(h) << 16) | (uint32_t)j;
      rowRaw(r)[j] = h;
      rowIds(r)[j] = s_ids[r][so];
    }
  }
  __syncthreads();  // the last workgroup barrier: from here on every wave owns its row
  if (c0 + w >= Ctot) return;
  uint32_t* A = rowA(w);
  wave_topk_row(A, rowT(w), rowStk(w), n, k, lane);  // :140
  // gather values / ids through the selected positions — :140-141 (the old state was copied to LDS above: in place is safe)
  const uint16_t* raw = rowRaw(w);
  const int64_t* old_ids = rowIds(w);
  const int64_t so = s_c[w] * k;
  uint16_t* vals = s_vals[w];
  int64_t* ids = s_ids[w];
  const int64_t id_base = s_base[w];
  for (int j0 = 0; j0 < k; j0 += 64) {
    const int j = j0 + lane;
    if (j < k) {
      const int pos = (int)(A[j] & 0xFFFFu);
      vals[so + j] = raw[pos];
      ids[so + j] = pos < k ? old_ids[pos] : (sample_ids ? sample_ids[pos - k] : id_base + (pos - k));
    }
  }
}

// dynamic LDS a launch may ask for: the kernel also holds 48 x ROWS bytes of static tables (s_cand ... s_C, ROWS <= 8), which count
// against the same 64 KiB (no attribute set) / 160 KiB (the CU's LDS) limits — round-5 advisor: rows within 384 bytes of a limit
// failed at launch while `_supported` said yes
constexpr size_t kLdsStatic = 48 * 8 + 128;
constexpr size_t kLdsBudget = 64 * 1024 - kLdsStatic;
constexpr size_t kLdsMax = 160 * 1024 - kLdsStatic;

constexpr bool wave_impl() { return true; }  // (the one-lane-per-row kernel of round 4 remains the path for rows too long for the LDS)

int launch_wave(ProfScope& prof, const K3States& tab, int L, int64_t k, const int64_t* d_sample_ids, int64_t B, hipStream_t st) {
  const int64_t n = k + B, Ctot = tab.cstart[L];
  const size_t rb = wave_row_bytes((int)n, (int)k);
#define SL_K3_WAVE(ROWS_)                                                                                                        \
  do {                                                                                                                           \
    const size_t lds = rb * ROWS_;                                                                                               \
    if (lds > kLdsBudget)                                                                                                        \
      SL_CHECK_HIP(hipFuncSetAttribute((const void*)actmax_update_aten_wave_kernel<ROWS_>,                                        \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                    \
    SL_LAUNCH(prof, actmax_update_aten_wave_kernel<ROWS_>, dim3((unsigned)((Ctot + ROWS_ - 1) / ROWS_)), dim3(64 * ROWS_), lds,    \
              st, tab, L, Ctot, (int)k, d_sample_ids, (int)B);                                                             \
  } while (0)
  // eight rows per workgroup (16 contiguous candidate bytes per sample) while the rows fit; fewer for long rows or few components
  if (rb * 8 <= kLdsBudget && Ctot >= 1024) SL_K3_WAVE(8);
  else if (rb * 4 <= kLdsBudget && Ctot >= 256) SL_K3_WAVE(4);
  else if (rb * 2 <= kLdsMax && Ctot >= 2) SL_K3_WAVE(2);
  else SL_K3_WAVE(1);
#undef SL_K3_WAVE
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace

int actmax_update_aten(ProfScope& prof, uint16_t* d_vals, int64_t* d_ids, int64_t C, int64_t k, const uint16_t* d_cand,
                       const int64_t* d_sample_ids, int64_t id_base, int64_t B, void* d_ws, size_t ws_bytes,
                       hipStream_t st) {
  const int64_t n = k + B;
  SL_REQUIRE(n <= 16384, "sl_actmax_update(SL_TIES_ATEN): k + B = %lld exceeds 16384", (long long)n);
  SL_REQUIRE(d_ws && ws_bytes >= sl_actmax_aten_ws_bytes(C, k, B),
             "sl_actmax_update(SL_TIES_ATEN): workspace too small (%zu < %zu)", ws_bytes,
             sl_actmax_aten_ws_bytes(C, k, B));
  if (wave_impl() && n <= 65535 && wave_row_bytes((int)n, (int)k) <= kLdsMax) {
    K3States tab;
    tab.vals[0] = d_vals, tab.ids[0] = d_ids, tab.cand[0] = d_cand, tab.id_base[0] = id_base;
    tab.cstart[0] = 0, tab.cstart[1] = C;
    return launch_wave(prof, tab, 1, k, d_sample_ids, B, st);
  }
  int rpb = (int)((C + 511) / 512);  // spread rows over >= 512 waves when C allows
  const int lds_cap = (int)(kLdsBudget / ((size_t)n * 4));
  if (rpb > lds_cap) rpb = lds_cap;
  if (rpb > 64) rpb = 64;
  if (rpb < 1) rpb = 1;
  const unsigned blocks = (unsigned)((C + rpb - 1) / rpb);
  int64_t* ws_ids = reinterpret_cast<int64_t*>(d_ws);
  uint16_t* ws_vals = reinterpret_cast<uint16_t*>(ws_ids + C * k);
  SL_LAUNCH(prof, actmax_update_aten_kernel, dim3(blocks), dim3(64), (size_t)n * rpb * 4, st, d_vals, d_ids, C, (int)k,
            d_cand, d_sample_ids, id_base, (int)B, rpb, ws_ids, ws_vals);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace sl

SL_API int sl_actmax_update_multi_supported(int64_t C, int64_t k, int64_t B) {
  const int64_t n = k + B;
  return (C >= 1 && k >= 1 && B >= 1 && sl::wave_impl() && n <= 16384 && sl::wave_row_bytes((int)n, (int)k) <= sl::kLdsMax) ? 1 : 0;
}

SL_API int sl_actmax_update_multi(uint16_t* const* h_d_vals, int64_t* const* h_d_ids, const int64_t* h_id_bases, const int64_t* h_Cs,
                                  const uint16_t* const* h_d_cands, int L, int64_t k, int64_t B, void* stream) {
  using namespace sl;
  SL_REQUIRE(L >= 0 && k >= 0 && B >= 0, "sl_actmax_update_multi: negative shape");
  if (L == 0 || k == 0 || B == 0) return 0;
  SL_REQUIRE(h_d_vals && h_d_ids && h_id_bases && h_Cs && h_d_cands, "sl_actmax_update_multi: null pointer");
  SL_REQUIRE(sl_actmax_update_multi_supported(1, k, B), "sl_actmax_update_multi: k + B = %lld rows do not fit the one-wave-per-row kernel "
             "(update the layers one by one with sl_actmax_update)", (long long)(k + B));
  hipStream_t st = (hipStream_t)stream;
  for (int l0 = 0; l0 < L; l0 += kMaxK3Layers) {
    K3States tab;
    int n = 0;
    int64_t total = 0;
    for (int i = l0; i < L && i < l0 + kMaxK3Layers; ++i) {
      SL_REQUIRE(h_Cs[i] >= 0, "sl_actmax_update_multi: negative component count");
      if (h_Cs[i] == 0) continue;
      SL_REQUIRE(h_d_vals[i] && h_d_ids[i] && h_d_cands[i], "sl_actmax_update_multi: null state or candidates");
      tab.vals[n] = h_d_vals[i], tab.ids[n] = h_d_ids[i], tab.cand[n] = h_d_cands[i], tab.id_base[n] = h_id_bases[i];
      tab.cstart[n] = total;
      total += h_Cs[i];
      ++n;
    }
    if (n == 0) continue;
    tab.cstart[n] = total;
    ProfScope prof(SL_PROF_MERGE, st, (double)total * ((double)B * 2 + (double)k * 10));
    const int rc = launch_wave(prof, tab, n, k, nullptr, B, st);
    if (rc) return rc;
  }
  return 0;
}

/* Host-only: the positions torch.topk's CPU kernel selects from ONE row of n bf16 values (TopKImpl.h through libstdc++, restated in
 * aten_topk_order.hpp), best first.  No device is touched: the Python side checks this against the installed torch.topk once per
 * process, so that a torch / libstdc++ pair that orders ties differently is noticed instead of silently giving other sample ids. */
SL_API int sl_aten_topk_order_host(const uint16_t* h_vals_bf16, int64_t n, int64_t k, int32_t* h_positions) {
  SL_REQUIRE(h_vals_bf16 && h_positions, "sl_aten_topk_order_host: null pointer");
  SL_REQUIRE(n >= 1 && n <= 65535 && k >= 0 && k <= n, "sl_aten_topk_order_host: need 1 <= n <= 65535 and 0 <= k <= n");
  std::vector<uint32_t> a((size_t)n);
  for (int64_t j = 0; j < n; ++j) a[(size_t)j] = (sl::bf16_order_key(h_vals_bf16[j]) << 16) | (uint32_t)j;
  uint32_t* p = a.data();
  sl::aten_order::topk_order(p, (int)n, (int)k);
  for (int64_t j = 0; j < k; ++j) h_positions[j] = (int32_t)(a[(size_t)j] & 0xFFFFu);
  return 0;
}

SL_API size_t sl_actmax_aten_ws_bytes(int64_t C, int64_t k, int64_t B) {
  (void)B;
  return (size_t)(C * k) * (sizeof(int64_t) + sizeof(uint16_t)) + 16;
}
