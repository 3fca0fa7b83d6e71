// K9 — polysemanticity_score (semanticlens/scores.py:131-185).  Placeholder until the Gram-space
// 2-means kernel lands: reports SL_E_UNSUPPORTED so callers fail loudly (no CPU fallback).
#include "common.hpp"

using namespace sl;

SL_API size_t sl_poly2means_ws_bytes(int64_t C, int64_t n, int64_t D) {
  (void)D;
  return (size_t)C * (size_t)n * (size_t)n * 8 + 256;
}

SL_API int sl_poly2means(const float* d_V, int64_t C, int64_t n, int64_t D, const int32_t* h_first_center, int n_init,
                         const double* h_rand, int replace_empty_clusters, double* d_out, int32_t* d_min_count,
                         void* d_ws, size_t ws_bytes, void* stream) {
  set_error("sl_poly2means: not built yet");
  return SL_E_UNSUPPORTED;
}
