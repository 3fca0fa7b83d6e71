// K5 — concept_db[layer] = embeds[sample_ids] (component_visualization/activation_based.py:387-390).
// Row gather (N,D) -> (n_ids,D), HBM-bound: n_ids*D*4 bytes read + the same written.
// Negative ids wrap like torch advanced indexing, so the -1 sentinel of an unfilled top-k slot
// reads the LAST embedding (SURVEY.md finding 2).
#include "common.hpp"

namespace sl {
namespace {

// emb holds rows [row_offset, row_offset + n_local) of a table with n_total rows; ids index the
// whole table.  Rows this shard does not hold are written as zeros (plain gather: row_offset = 0,
// n_local = n_total).
template <bool VEC4>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ emb, int64_t n_local, int64_t D,
                                                           const int64_t* __restrict__ ids, int64_t n_ids,
                                                           int64_t row_offset, int64_t n_total,
                                                           float* __restrict__ out, int32_t* __restrict__ err) {
  const int64_t per_row = VEC4 ? D / 4 : D;
  const int64_t total = n_ids * per_row;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / per_row, col = i % per_row;
    int64_t src = ids[row];
    if (src < 0) src += n_total;
    if (src < 0 || src >= n_total) {  // torch raises IndexError; flag it for the host
      if (err) *err = 1;
      src = 0;
    }
    src -= row_offset;
    const bool mine = src >= 0 && src < n_local;
    if constexpr (VEC4) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (mine) v = reinterpret_cast<const float4*>(emb)[src * per_row + col];
      reinterpret_cast<float4*>(out)[row * per_row + col] = v;
    } else {
      out[row * D + col] = mine ? emb[src * D + col] : 0.f;
    }
  }
}

// Round 6: one wavefront per group of RPW output rows.  The element-indexed kernel above pays a 64-bit division and a re-read of
// the row's id per 16 bytes; here a row's id is read once (wave-uniform, scalar), the lanes walk the row in 16-byte pieces, the
// loads of all RPW rows are issued before the first store, and the output — written once, usually larger than the Infinity Cache:
// 849 MB for the configs[3] concept DB — leaves with non-temporal stores while the table rows (re-read k times, cache-resident
// for any DB whose referenced rows fit) keep the default policy.  D % 4 == 0, 16-byte aligned rows.
template <int RPW, int PPL>  // rows per wave and step; 16-byte pieces per lane and row = ceil(D / 256)
__global__ __launch_bounds__(256) void gather_rows_wave_kernel(const float* __restrict__ emb, int64_t n_local, int64_t D,
                                                                const int64_t* __restrict__ ids, int64_t n_ids,
                                                                int64_t row_offset, int64_t n_total,
                                                                float* __restrict__ out, int32_t* __restrict__ err) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwaves = (int64_t)gridDim.x * 4;
  const int per_row = (int)(D / 4);
  for (int64_t r0 = wave * RPW; r0 < n_ids; r0 += nwaves * RPW) {
    f4 v[RPW][PPL];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int64_t row = r0 + j;
      int64_t src = row < n_ids ? ids[row] : 0;  // wave-uniform address: a scalar load
      if (src < 0) src += n_total;
      if (src < 0 || src >= n_total) {  // torch raises IndexError; flag it for the host
        bad = true;
        src = 0;
      }
      src -= row_offset;
      const bool mine = row < n_ids && src >= 0 && src < n_local;
      const f4* p = reinterpret_cast<const f4*>(emb) + src * per_row;
#pragma unroll
      for (int q = 0; q < PPL; ++q) {
        const int c = q * 64 + lane;
        v[j][q] = (mine && c < per_row) ? p[c] : f4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (bad && err && lane == 0) *err = 1;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
      const int64_t row = r0 + j;
      if (row < n_ids) {
        f4* o = reinterpret_cast<f4*>(out) + row * per_row;
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
          const int c = q * 64 + lane;
          if (c < per_row) __builtin_nontemporal_store(v[j][q], o + c);
        }
      }
    }
  }
}

template <int RPW, int PPL>
void launch_gather_wave(ProfScope& prof, const float* d_emb, int64_t n_local, int64_t D, const int64_t* d_ids, int64_t n_ids,
                        int64_t row_offset, int64_t n_total, float* d_out, int32_t* d_err_flag, hipStream_t st) {
  int64_t blocks = (n_ids + 4 * RPW - 1) / (4 * RPW);
  const int64_t cap = (int64_t)num_cus() * 8;
  if (blocks > cap) blocks = cap;
  SL_LAUNCH(prof, (gather_rows_wave_kernel<RPW, PPL>), dim3((unsigned)blocks), dim3(256), 0, st, d_emb, n_local, D, d_ids, n_ids,
            row_offset, n_total, d_out, d_err_flag);
}

int launch_gather(const float* d_emb, int64_t n_local, int64_t D, const int64_t* d_ids, int64_t n_ids,
                  int64_t row_offset, int64_t n_total, float* d_out, int32_t* d_err_flag, hipStream_t st) {
  ProfScope prof(SL_PROF_GATHER, st, (double)n_ids * D * 8);
  const bool vec = (D % 4 == 0) && (((uintptr_t)d_emb | (uintptr_t)d_out) & 15) == 0;
  if (vec && D <= 2048 && n_ids >= 64) {  // rows of up to eight 16-byte pieces per lane: the wave-per-row kernel
    const int ppl = (int)((D / 4 + 63) / 64);
#define SL_GATHER_WAVE(R_, P_) launch_gather_wave<R_, P_>(prof, d_emb, n_local, D, d_ids, n_ids, row_offset, n_total, d_out, d_err_flag, st)
    switch (ppl) {
      case 1: SL_GATHER_WAVE(8, 1); break;
      case 2: SL_GATHER_WAVE(4, 2); break;
      case 3: SL_GATHER_WAVE(4, 3); break;
      case 4: SL_GATHER_WAVE(2, 4); break;
      case 5: SL_GATHER_WAVE(2, 5); break;
      case 6: SL_GATHER_WAVE(2, 6); break;
      default: SL_GATHER_WAVE(1, 8); break;
    }
#undef SL_GATHER_WAVE
    SL_CHECK_HIP(hipGetLastError());
    return 0;
  }
  const int64_t total = n_ids * (vec ? D / 4 : D);
  int64_t blocks = (total + 255) / 256;
  const int64_t cap = (int64_t)num_cus() * 16;
  if (blocks > cap) blocks = cap;
  if (vec)
    SL_LAUNCH(prof, gather_rows_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, d_emb, n_local, D, d_ids, n_ids,
              row_offset, n_total, d_out, d_err_flag);
  else
    SL_LAUNCH(prof, gather_rows_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, d_emb, n_local, D, d_ids, n_ids,
              row_offset, n_total, d_out, d_err_flag);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace
}  // namespace sl

using namespace sl;

SL_API int sl_gather_rows(const float* d_emb, int64_t N, int64_t D, const int64_t* d_ids, int64_t n_ids, float* d_out,
                          int32_t* d_err_flag, void* stream) {
  SL_REQUIRE(N >= 0 && D >= 0 && n_ids >= 0, "sl_gather_rows: negative shape");
  if (n_ids * D == 0) return 0;
  SL_REQUIRE(N > 0, "sl_gather_rows: gather from an empty embedding table");
  SL_REQUIRE(d_emb && d_ids && d_out, "sl_gather_rows: null pointer");
  return launch_gather(d_emb, N, D, d_ids, n_ids, 0, N, d_out, d_err_flag, (hipStream_t)stream);
}

SL_API int sl_gather_rows_shard(const float* d_emb_local, int64_t n_local, int64_t D, const int64_t* d_ids,
                                int64_t n_ids, int64_t row_offset, int64_t n_total, float* d_out,
                                int32_t* d_err_flag, void* stream) {
  SL_REQUIRE(n_local >= 0 && D >= 0 && n_ids >= 0 && row_offset >= 0 && n_total > 0 && row_offset + n_local <= n_total,
             "sl_gather_rows_shard: bad shard geometry");
  if (n_ids * D == 0) return 0;
  SL_REQUIRE((d_emb_local || n_local == 0) && d_ids && d_out, "sl_gather_rows_shard: null pointer");
  return launch_gather(d_emb_local, n_local, D, d_ids, n_ids, row_offset, n_total, d_out, d_err_flag,
                       (hipStream_t)stream);
}
