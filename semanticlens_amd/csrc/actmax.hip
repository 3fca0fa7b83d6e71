// K3 / K4 — streaming per-component top-k state ("ActMax").
//
// Replaces ActMax._setup_tensors / ActMax.update (component_visualization/
// activation_caching.py:101-141): `cat([state, batch]) -> torch.topk -> gather` on the host
// becomes an in-place merge of bf16 candidates into a sorted (C,k) state that lives in HBM.
//
// One wavefront owns one component.  Its state row sits in LDS (packed order-key|bits word +
// int64 id per entry).  Candidates are read 64 at a time; a wave-wide ballot against the
// current k-th entry filters them (in steady state almost nothing passes), survivors are
// inserted one by one: rank by ballot/popcount over the sorted row, shift the tail one slot
// (all lanes, LDS), drop the new entry in.
//
// Tie order (SL_TIES_TOTAL): value descending with ATen's comparator semantics (NaN first,
// -0.0 == +0.0), then sample id ascending.  Being a strict total order on (value,id) it makes
// the result independent of batch size, merge order and sharding (K4).
//
// These kernels are latency-bound and tiny next to K1/K2: per merge they read
// B*C*2 bytes of candidates and 10*C*k bytes of state.
#include "common.hpp"

namespace sl {
namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kMaxK = 2048;

struct SlotArgs {
  int nslots;
  int64_t stride;
  int64_t id_base[SL_MAX_SLOTS];
  int64_t rows[SL_MAX_SLOTS];
};

__global__ __launch_bounds__(256) void actmax_init_kernel(uint16_t* vals, int64_t* ids, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    vals[i] = 0x8000;  // -0.0  (activation_caching.py:108)
    ids[i] = -1;       //       (activation_caching.py:109)
  }
}

__device__ inline int64_t readlane_i64(int64_t v, int l) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), l);
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// Sorted row of one component, owned by one wavefront.
struct Row {
  uint32_t* kb;  // (order key << 16) | bf16 bits
  int64_t* id;
  int k;
  uint32_t thr_key;
  int64_t thr_id;
  bool dirty;

  __device__ inline void load(const uint16_t* vals, const int64_t* ids, int64_t c, int lane) {
    for (int i = lane; i < k; i += kWave) {
      uint16_t b = vals[c * k + i];
      kb[i] = (bf16_order_key(b) << 16) | b;
      id[i] = ids[c * k + i];
    }
    dirty = false;
    refresh_threshold();
  }
  __device__ inline void refresh_threshold() {
    // same address in every lane (LDS broadcast); readfirstlane makes it provably wave-uniform
    thr_key = (uint32_t)__builtin_amdgcn_readfirstlane((int)(kb[k - 1] >> 16));
    const int64_t t = id[k - 1];
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)t);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)t >> 32));
    thr_id = (int64_t)(((uint64_t)hi << 32) | lo);
  }
  __device__ inline bool beats_threshold(uint32_t key, int64_t i) const { return better(key, i, thr_key, thr_id); }

  // wave-uniform (key, bits, cid): insert if it still beats the k-th entry
  __device__ inline void insert(uint32_t key, uint32_t bits, int64_t cid, int lane) {
    if (!beats_threshold(key, cid)) return;
    int p = 0;
    for (int base = 0; base < k; base += kWave) {
      const int i = base + lane;
      const bool bt = i < k && better(kb[i] >> 16, id[i], key, cid);
      p += __popcll(__ballot(bt));
    }
    // shift [p, k-2] -> [p+1, k-1]; highest chunk first so every read sees the old value
    for (int base = ((k - 1) / kWave) * kWave; base >= 0; base -= kWave) {
      const int i = base + lane;
      const bool mv = i > p && i < k;
      uint32_t tkb = 0;
      int64_t tid = 0;
      if (mv) {
        tkb = kb[i - 1];
        tid = id[i - 1];
      }
      if (mv) {
        kb[i] = tkb;
        id[i] = tid;
      }
    }
    if (lane == 0) {
      kb[p] = (key << 16) | bits;
      id[p] = cid;
    }
    dirty = true;
    refresh_threshold();
  }

  // a per-lane candidate (key,bits,cid,valid): filter with one ballot, insert survivors in lane order
  __device__ inline void offer(uint32_t key, uint32_t bits, int64_t cid, bool valid, int lane) {
    uint64_t mask = __ballot(valid && beats_threshold(key, cid));
    while (mask) {
      const int l = __ffsll((unsigned long long)mask) - 1;
      mask &= mask - 1;
      const uint32_t kk = (uint32_t)__builtin_amdgcn_readlane((int)key, l);
      const uint32_t bb = (uint32_t)__builtin_amdgcn_readlane((int)bits, l);
      const int64_t ii = readlane_i64(cid, l);
      insert(kk, bb, ii, lane);
    }
  }

  __device__ inline void store(uint16_t* vals, int64_t* ids, int64_t c, int lane) const {
    if (!dirty) return;
    for (int i = lane; i < k; i += kWave) {
      vals[c * k + i] = (uint16_t)(kb[i] & 0xFFFFu);
      ids[c * k + i] = id[i];
    }
  }
};

__device__ inline Row make_row(unsigned char* smem, int k, int w) {
  Row r;
  r.id = reinterpret_cast<int64_t*>(smem) + (size_t)w * k;
  r.kb = reinterpret_cast<uint32_t*>(smem + (size_t)kWavesPerBlock * k * sizeof(int64_t)) + (size_t)w * k;
  r.k = k;
  return r;
}

// candidates: `nslots` (rows,C) bf16 matrices; ids = id_base[slot] + b, or sample_ids[b] (slot 0)
__global__ __launch_bounds__(256) void actmax_merge_kernel(uint16_t* __restrict__ vals, int64_t* __restrict__ ids,
                                                            int64_t C, int k, const uint16_t* __restrict__ cand,
                                                            SlotArgs sa, const int64_t* __restrict__ sample_ids) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * kWavesPerBlock + w;
  if (c >= C) return;
  Row row = make_row(smem, k, w);
  row.load(vals, ids, c, lane);
  for (int s = 0; s < sa.nslots; ++s) {
    const uint16_t* cs = cand + (int64_t)s * sa.stride;
    const int64_t rows = sa.rows[s];
    for (int64_t b0 = 0; b0 < rows; b0 += kWave) {
      const int64_t b = b0 + lane;
      const bool valid = b < rows;
      const uint32_t bits = valid ? cs[b * C + c] : 0u;
      const int64_t cid = valid ? (sample_ids ? sample_ids[b] : sa.id_base[s] + b) : 0;
      row.offer(bf16_order_key((uint16_t)bits), bits, cid, valid, lane);
    }
  }
  row.store(vals, ids, c, lane);
}

// candidates: R other states with explicit ids; state r's row c starts at ovals + r * stride_v + c * k (ids likewise with
// stride_i) — (R,C,k) tensors have stride C*k, the all-gathered packed buffer of sl_actmax_allgather_merge one rank's block.
// State `skip` (this rank's own block in an all-gathered buffer; -1: none) is not read.
__global__ __launch_bounds__(256) void actmax_merge_states_kernel(uint16_t* __restrict__ vals,
                                                                   int64_t* __restrict__ ids, int64_t C, int k,
                                                                   const uint16_t* __restrict__ ovals,
                                                                   const int64_t* __restrict__ oids, int64_t R,
                                                                   int64_t stride_v, int64_t stride_i, int64_t skip) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * kWavesPerBlock + w;
  if (c >= C) return;
  Row row = make_row(smem, k, w);
  row.load(vals, ids, c, lane);
  for (int64_t r = 0; r < R; ++r) {
    if (r == skip) continue;
    const uint16_t* ov = ovals + r * stride_v + c * k;
    const int64_t* oi = oids + r * stride_i + c * k;
    for (int j0 = 0; j0 < k; j0 += kWave) {
      const int j = j0 + lane;
      const bool valid = j < k;
      const uint32_t bits = valid ? ov[j] : 0u;
      const int64_t cid = valid ? oi[j] : 0;
      row.offer(bf16_order_key((uint16_t)bits), bits, cid, valid, lane);
    }
  }
  row.store(vals, ids, c, lane);
}

size_t row_smem_bytes(int64_t k) { return (size_t)kWavesPerBlock * (size_t)k * (sizeof(int64_t) + sizeof(uint32_t)); }
// k in (1365, 2048] needs 64-96 KiB of dynamic LDS: above the default limit the attribute must be raised first
template <class K>
int allow_smem(K kernel, size_t bytes) {
  if (bytes > 64 * 1024)
    SL_CHECK_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return 0;
}

int check_state(const char* fn, const void* vals, const void* ids, int64_t C, int64_t k) {
  SL_REQUIRE(C >= 0 && k >= 0, "%s: negative shape", fn);
  SL_REQUIRE((vals && ids) || C * k == 0, "%s: null state", fn);
  SL_REQUIRE(k <= kMaxK, "%s: k=%lld exceeds the supported maximum %d", fn, (long long)k, kMaxK);
  return 0;
}

}  // namespace
}  // namespace sl

using namespace sl;

SL_API int sl_actmax_init(uint16_t* d_vals, int64_t* d_ids, int64_t C, int64_t k, void* stream) {
  if (int rc = check_state("sl_actmax_init", d_vals, d_ids, C, k)) return rc;
  const int64_t n = C * k;
  if (n == 0) return 0;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(actmax_init_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_vals, d_ids, n);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

SL_API int sl_actmax_merge(uint16_t* d_vals, int64_t* d_ids, int64_t C, int64_t k, const uint16_t* d_cand,
                           int64_t slot_stride, const int64_t* h_slot_id_base, const int64_t* h_slot_rows, int nslots,
                           void* stream) {
  if (int rc = check_state("sl_actmax_merge", d_vals, d_ids, C, k)) return rc;
  SL_REQUIRE(nslots >= 0 && nslots <= SL_MAX_SLOTS, "sl_actmax_merge: nslots=%d not in [0,%d]", nslots, SL_MAX_SLOTS);
  if (C * k == 0 || nslots == 0) return 0;  // topk with k == 0 returns nothing (TopKImpl.h)
  SL_REQUIRE(d_cand && h_slot_id_base && h_slot_rows, "sl_actmax_merge: null candidate arguments");
  SlotArgs sa;
  sa.nslots = nslots;
  sa.stride = slot_stride;
  double bytes = 0;
  for (int s = 0; s < nslots; ++s) {
    SL_REQUIRE(h_slot_rows[s] >= 0, "sl_actmax_merge: negative row count in slot %d", s);
    sa.id_base[s] = h_slot_id_base[s];
    sa.rows[s] = h_slot_rows[s];
    bytes += (double)h_slot_rows[s] * C * 2;
  }
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof(SL_PROF_MERGE, st, bytes + (double)C * k * 10);
  const unsigned blocks = (unsigned)((C + kWavesPerBlock - 1) / kWavesPerBlock);
  if (int rc = allow_smem(actmax_merge_kernel, row_smem_bytes(k))) return rc;
  SL_LAUNCH(prof, actmax_merge_kernel, dim3(blocks), dim3(256), row_smem_bytes(k), st, d_vals, d_ids, C, (int)k, d_cand, sa,
            (const int64_t*)nullptr);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

namespace sl {
int actmax_update_aten(ProfScope& prof, uint16_t* d_vals, int64_t* d_ids, int64_t C, int64_t k, const uint16_t* d_cand,
                       const int64_t* d_sample_ids, int64_t id_base, int64_t B, void* d_ws, size_t ws_bytes,
                       hipStream_t st);
}

SL_API int sl_actmax_update(uint16_t* d_vals, int64_t* d_ids, int64_t C, int64_t k, const uint16_t* d_cand,
                            const int64_t* d_sample_ids, int64_t id_base, int64_t B, int ties, void* d_ws,
                            size_t ws_bytes, void* stream) {
  if (int rc = check_state("sl_actmax_update", d_vals, d_ids, C, k)) return rc;
  SL_REQUIRE(B >= 0, "sl_actmax_update: negative batch");
  SL_REQUIRE(ties == SL_TIES_TOTAL || ties == SL_TIES_ATEN, "sl_actmax_update: bad ties mode %d", ties);
  if (C * k == 0 || B == 0) return 0;
  SL_REQUIRE(d_cand, "sl_actmax_update: null candidates");
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof(SL_PROF_MERGE, st, (double)B * C * 2 + (double)C * k * 10);
  if (ties == SL_TIES_ATEN)
    return actmax_update_aten(prof, d_vals, d_ids, C, k, d_cand, d_sample_ids, id_base, B, d_ws, ws_bytes, st);
  SlotArgs sa;
  sa.nslots = 1;
  sa.stride = 0;
  sa.id_base[0] = id_base;
  sa.rows[0] = B;
  const unsigned blocks = (unsigned)((C + kWavesPerBlock - 1) / kWavesPerBlock);
  if (int rc = allow_smem(actmax_merge_kernel, row_smem_bytes(k))) return rc;
  SL_LAUNCH(prof, actmax_merge_kernel, dim3(blocks), dim3(256), row_smem_bytes(k), st, d_vals, d_ids, C, (int)k, d_cand, sa,
            d_sample_ids);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

namespace sl {
// shared with comm.hip (sl_actmax_allgather_merge): strides in ELEMENTS of the respective array
int merge_states_strided(const char* fn, uint16_t* d_vals, int64_t* d_ids, int64_t C, int64_t k, const uint16_t* d_other_vals,
                         const int64_t* d_other_ids, int64_t R, int64_t stride_v, int64_t stride_i, int64_t skip,
                         hipStream_t st) {
  if (int rc = check_state(fn, d_vals, d_ids, C, k)) return rc;
  SL_REQUIRE(R >= 0, "%s: negative R", fn);
  if (C * k == 0 || R == 0 || (R == 1 && skip == 0)) return 0;
  SL_REQUIRE(d_other_vals && d_other_ids, "%s: null inputs", fn);
  ProfScope prof(SL_PROF_MERGE, st, (double)(R + 1 - (skip >= 0 && skip < R ? 1 : 0)) * C * k * 10);
  const unsigned blocks = (unsigned)((C + kWavesPerBlock - 1) / kWavesPerBlock);
  if (int rc = allow_smem(actmax_merge_states_kernel, row_smem_bytes(k))) return rc;
  SL_LAUNCH(prof, actmax_merge_states_kernel, dim3(blocks), dim3(256), row_smem_bytes(k), st, d_vals, d_ids, C, (int)k,
            d_other_vals, d_other_ids, R, stride_v, stride_i, skip);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}
}  // namespace sl

SL_API int sl_actmax_merge_states(uint16_t* d_vals, int64_t* d_ids, int64_t C, int64_t k,
                                  const uint16_t* d_other_vals, const int64_t* d_other_ids, int64_t R, void* stream) {
  return merge_states_strided("sl_actmax_merge_states", d_vals, d_ids, C, k, d_other_vals, d_other_ids, R, C * k, C * k, -1,
                              (hipStream_t)stream);
}
