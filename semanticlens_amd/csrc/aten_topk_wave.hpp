// Data-parallel restatement of aten_topk_order.hpp's nth_element + sort path (round 5).
//
// libstdc++'s introselect / introsort are sequential, but every step they take is a deterministic function of the
// array it starts from, and the two steps that carry all the work have closed forms a wavefront can evaluate at once:
//
//  * `unguarded_partition(first, last, pivot)` (Hoare).  Let L = the positions of [first, last) whose key is <= the pivot's,
//    ascending, and R = the positions whose key is >= the pivot's, descending.  The left scan stops exactly at the
//    members of L, the right scan at the members of R, and the j-th swap exchanges L[j] and R[j] as long as
//    L[j] < R[j] — a monotone condition, so the number of swaps is m = #{j : L[j] < R[j]}.  All L[j < m] lie left of all
//    R[j < m] (no position is swapped twice), untouched positions keep their values, and the returned cut is
//    min(L[m] if it exists, R[m-1] if m > 0): the left scan continues from L[m-1] + 1 over untouched values until the
//    next member of L, unless it first reaches R[m-1], which now holds a key <= the pivot's.
//  * `__final_insertion_sort` / `__insertion_sort` move an element back only past strictly smaller-ranked ones
//    (`cmp` is strict), i.e. they are a STABLE sort of the arrangement the partition steps left behind:
//    rank(i) = #{j : key_j > key_i} + #{j < i : key_j == key_i}.
//
// Everything else (median-of-three, the range bookkeeping) is O(1) per step.  The heap fallbacks (depth limit reached,
// or torch's `k * 64 <= n` partial_sort branch) stay sequential: `aten_order::` on the same array.
//
// `lists::` below is that restatement as plain host/device code (tests/native/aten_order_check.cpp checks it against
// libstdc++ itself through aten_order::topk_order); actmax_aten.hip evaluates the same lists with ballots and LDS tables,
// one wavefront per row.
#pragma once
#include "aten_topk_order.hpp"

namespace sl {
namespace aten_order {
namespace lists {

// Hoare partition of [lo, hi) around the pivot ELEMENT `pv` (not in the range) by the L / R lists; `tab` has room for
// 2 * (hi - lo) ints.  Returns the cut.
template <class A>
SL_HD inline int partition(A& a, int lo, int hi, uint32_t pv, int* tab) {
  const uint32_t pk = pv >> 16;
  int* L = tab;
  int* Rasc = tab + (hi - lo);
  int nL = 0, nR = 0;
  for (int p = lo; p < hi; ++p) {
    const uint32_t key = a[p] >> 16;
    if (key <= pk) L[nL++] = p;
    if (key >= pk) Rasc[nR++] = p;
  }
  const int mn = nL < nR ? nL : nR;
  int m = 0;
  while (m < mn && L[m] < Rasc[nR - 1 - m]) ++m;
  for (int j = 0; j < m; ++j) swap_at(a, L[j], Rasc[nR - 1 - j]);
  const int big = 0x7FFFFFFF;
  const int cutL = m < nL ? L[m] : big;
  const int cutR = m > 0 ? Rasc[nR - m] : big;
  return cutL < cutR ? cutL : cutR;
}

// stable descending sort of [first, last) by rank; `tmp` has room for last - first words
template <class A>
SL_HD inline void stable_sort(A& a, int first, int last, uint32_t* tmp) {
  for (int i = first; i < last; ++i) {
    const uint32_t ki = a[i] >> 16;
    int rank = 0;
    for (int j = first; j < last; ++j) {
      const uint32_t kj = a[j] >> 16;
      rank += (kj > ki) || (kj == ki && j < i);
    }
    tmp[rank] = a[i];
  }
  for (int i = first; i < last; ++i) a[i] = tmp[i - first];
}

template <class A>
SL_HD inline int partition_pivot(A& a, int first, int last, int* tab) {
  const int mid = first + (last - first) / 2;
  move_median_to_first(a, first, first + 1, mid, last - 1);
  return partition(a, first + 1, last, a[first], tab);
}

// topk_order's nth_element + sort branch; `tab`: 2 n ints, `tmp`: n words
template <class A>
SL_HD inline void topk_order_nth(A& a, int n, int k, int* tab, uint32_t* tmp) {
  // std::nth_element(a, a + k - 1, a + n)
  int first = 0, last = n;
  const int nth = k - 1;
  if (nth != last) {
    int depth = floor_log2(last - first) * 2;
    bool done = false;
    while (last - first > 3) {
      if (depth == 0) {
        heap_select(a, first, nth + 1, last);
        swap_at(a, first, nth);
        done = true;
        break;
      }
      --depth;
      const int cut = partition_pivot(a, first, last, tab);
      if (cut <= nth) first = cut;
      else last = cut;
    }
    if (!done) stable_sort(a, first, last, tmp);  // __insertion_sort of <= 3 elements
  }
  // std::sort(a, a + k - 1)
  const int s_last = k - 1;
  if (s_last <= 0) return;
  constexpr int kThreshold = 16;
  int stack[48 * 3];
  int sp = 0;
  stack[0] = 0, stack[1] = s_last, stack[2] = floor_log2(s_last) * 2;
  sp = 1;
  while (sp > 0) {
    --sp;
    int f = stack[3 * sp], l = stack[3 * sp + 1], d = stack[3 * sp + 2];
    while (l - f > kThreshold) {
      if (d == 0) {
        partial_sort(a, f, l, l);
        break;
      }
      --d;
      const int cut = partition_pivot(a, f, l, tab);
      if (sp < 48) stack[3 * sp] = cut, stack[3 * sp + 1] = l, stack[3 * sp + 2] = d, ++sp;
      l = cut;
    }
  }
  stable_sort(a, 0, s_last, tmp);  // __final_insertion_sort
}

}  // namespace lists
}  // namespace aten_order
}  // namespace sl
