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

int launch_gather(const float* d_emb, int64_t n_local, int64_t D, const int64_t* d_ids, int64_t n_ids,
                  int64_t row_offset, int64_t n_total, float* d_out, int32_t* d_err_flag, hipStream_t st) {
  ProfScope prof(SL_PROF_GATHER, st, (double)n_ids * D * 8);
  const bool vec = (D % 4 == 0) && (((uintptr_t)d_emb | (uintptr_t)d_out) & 15) == 0;
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
