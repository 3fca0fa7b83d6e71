// Shared epilogue of the MFMA GEMM kernels (gemm_f32.hpp, gemm_bf16x3.hpp, gemm_8phase.hpp).
// An epilogue object `epi` provides `column(col)` (per-column value: bias, inverse norm, ...) and
// `store(row, col, acc, column_value)`.  One that ADDS A VALUE IT READS FROM MEMORY per element (the residual stream,
// positional rows: encoder.hip LinearEpi) additionally exposes
//     static constexpr bool kFetches = true;   fetch(row, col);   store_fetched(row, col, acc, column_value, fetched)
// with store(...) == store_fetched(..., fetch(row, col)) bit for bit.  The output may alias the memory that is read
// (x += W h in place), so written element by element the compiler must keep every load behind the store in front of it:
// 16 dependent round trips per MFMA tile and lane (+35 us on a 150-tile GEMM).  `store_mfma_tile` fetches the 16 values
// of a 32 x 32 tile first (16 loads in flight), then adds and stores; every element is still read and written by one lane.
#pragma once
#include "common.hpp"

namespace sl {

template <class E, class = void>
struct EpiFetches {
  static constexpr bool value = false;
};
template <class E>
struct EpiFetches<E, decltype((void)E::kFetches)> {
  static constexpr bool value = E::kFetches;
};

// One 32 x 32 MFMA accumulator tile (C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).
// row0 = first row of the tile + 4 (lane >> 5); col = this lane's column.  CHECK: rows / columns may lie past the edge.
template <bool CHECK, class Epi, class Acc16>
__device__ __forceinline__ void store_mfma_tile(const Epi& epi, int64_t row0, int64_t col, const Acc16& acc, int64_t M, int64_t N) {
  if (CHECK && col >= N) return;
  const auto cv = epi.column(col);
  if constexpr (EpiFetches<Epi>::value) {
    decltype(epi.fetch(row0, col)) pre[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = row0 + (r & 3) + 8 * (r >> 2);
      if (!CHECK || row < M) pre[r] = epi.fetch(row, col);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = row0 + (r & 3) + 8 * (r >> 2);
      if (!CHECK || row < M) epi.store_fetched(row, col, acc[r], cv, pre[r]);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = row0 + (r & 3) + 8 * (r >> 2);
      if (!CHECK || row < M) epi.store(row, col, acc[r], cv);
    }
  }
}

}  // namespace sl
