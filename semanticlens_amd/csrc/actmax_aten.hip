// K3, SL_TIES_ATEN — ActMax.update with torch.topk's exact CPU tie order
// (activation_caching.py:137-141; selection order restated in aten_topk_order.hpp).
//
// One lane per component runs the (inherently sequential) libstdc++ selection on the row
// [state | batch] of n = k + B packed (order-key, position) words held in LDS, element j of the
// lane's row at lds[j * rows_per_block + r].  Work is O(n) per row and data dependent; lanes of
// a wave diverge, so few rows share a wave and the rows are spread over many waves.  This mode
// exists for bit-identity with the reference; SL_TIES_TOTAL is the fast path.
#include "aten_topk_order.hpp"
#include "common.hpp"

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

constexpr size_t kLdsBudget = 64 * 1024;

}  // namespace

int actmax_update_aten(ProfScope& prof, uint16_t* d_vals, int64_t* d_ids, int64_t C, int64_t k, const uint16_t* d_cand,
                       const int64_t* d_sample_ids, int64_t id_base, int64_t B, void* d_ws, size_t ws_bytes,
                       hipStream_t st) {
  const int64_t n = k + B;
  SL_REQUIRE(n <= 16384, "sl_actmax_update(SL_TIES_ATEN): k + B = %lld exceeds 16384", (long long)n);
  SL_REQUIRE(d_ws && ws_bytes >= sl_actmax_aten_ws_bytes(C, k, B),
             "sl_actmax_update(SL_TIES_ATEN): workspace too small (%zu < %zu)", ws_bytes,
             sl_actmax_aten_ws_bytes(C, k, B));
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

SL_API size_t sl_actmax_aten_ws_bytes(int64_t C, int64_t k, int64_t B) {
  (void)B;
  return (size_t)(C * k) * (sizeof(int64_t) + sizeof(uint16_t)) + 16;
}
