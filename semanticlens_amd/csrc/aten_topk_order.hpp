// Restatement of the *selection order* of torch.topk's CPU kernel, so that SL_TIES_ATEN
// reproduces the reference's `torch.topk(all_acts, k, dim=1)` (activation_caching.py:140)
// bit-for-bit, ties included.
//
// What torch does per row of n = k + B elements (aten/src/ATen/native/cpu/TopKImpl.h):
//   queue[j] = (value_j, j); cmp(x, y) = (isnan(x) && !isnan(y)) || x > y
//   if (k * 64 <= n)  std::partial_sort(q, q + k, q + n, cmp)
//   else              std::nth_element(q, q + k - 1, q + n, cmp); std::sort(q, q + k - 1, cmp)
// and the outcome on tied values is whatever libstdc++'s introselect / introsort / heap code
// does.  Those algorithms are deterministic functions of the comparison results, so they are
// written out here over an index-addressed array (no iterators, no recursion) and compiled
// for both the device (one lane per row, array in LDS) and the host (tests/native checks this
// restatement against std::nth_element / std::sort / std::partial_sort themselves).
//
// Element encoding: uint32 = (order key << 16) | position, where the order key of a bf16 value
// (common.hpp bf16_order_key) is monotone in ATen's comparator: cmp(x, y) == key(x) > key(y).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define SL_HD __host__ __device__
#else
#define SL_HD
#endif

namespace sl {
namespace aten_order {

SL_HD inline bool cmp(uint32_t x, uint32_t y) { return (x >> 16) > (y >> 16); }

SL_HD inline int floor_log2(int n) {  // std::__lg
  int r = 0;
  while (n > 1) {
    n >>= 1;
    ++r;
  }
  return r;
}

template <class A>
SL_HD inline void swap_at(A& a, int i, int j) {
  uint32_t t = a[i];
  a[i] = a[j];
  a[j] = t;
}

// ---- heap primitives (bits/stl_heap.h) ----------------------------------------------------
template <class A>
SL_HD inline void push_heap_at(A& a, int first, int hole, int top, uint32_t value) {
  int parent = (hole - 1) / 2;
  while (hole > top && cmp(a[first + parent], value)) {
    a[first + hole] = a[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  a[first + hole] = value;
}

template <class A>
SL_HD inline void adjust_heap(A& a, int first, int hole, int len, uint32_t value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (cmp(a[first + child], a[first + child - 1])) --child;
    a[first + hole] = a[first + child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    a[first + hole] = a[first + child - 1];
    hole = child - 1;
  }
  push_heap_at(a, first, hole, top, value);
}

template <class A>
SL_HD inline void make_heap(A& a, int first, int last) {
  const int len = last - first;
  if (len < 2) return;
  int parent = (len - 2) / 2;
  while (true) {
    uint32_t v = a[first + parent];
    adjust_heap(a, first, parent, len, v);
    if (parent == 0) return;
    --parent;
  }
}

template <class A>
SL_HD inline void pop_heap(A& a, int first, int last, int result) {
  uint32_t v = a[result];
  a[result] = a[first];
  adjust_heap(a, first, 0, last - first, v);
}

template <class A>
SL_HD inline void heap_select(A& a, int first, int middle, int last) {
  make_heap(a, first, middle);
  for (int i = middle; i < last; ++i)
    if (cmp(a[i], a[first])) pop_heap(a, first, middle, i);
}

template <class A>
SL_HD inline void sort_heap(A& a, int first, int last) {
  while (last - first > 1) {
    --last;
    pop_heap(a, first, last, last);
  }
}

template <class A>
SL_HD inline void partial_sort(A& a, int first, int middle, int last) {
  heap_select(a, first, middle, last);
  sort_heap(a, first, middle);
}

// ---- partition / insertion primitives (bits/stl_algo.h) -----------------------------------
template <class A>
SL_HD inline void move_median_to_first(A& a, int result, int x, int y, int z) {
  if (cmp(a[x], a[y])) {
    if (cmp(a[y], a[z])) swap_at(a, result, y);
    else if (cmp(a[x], a[z])) swap_at(a, result, z);
    else swap_at(a, result, x);
  } else if (cmp(a[x], a[z])) swap_at(a, result, x);
  else if (cmp(a[y], a[z])) swap_at(a, result, z);
  else swap_at(a, result, y);
}

template <class A>
SL_HD inline int unguarded_partition(A& a, int first, int last, int pivot) {
  while (true) {
    while (cmp(a[first], a[pivot])) ++first;
    --last;
    while (cmp(a[pivot], a[last])) --last;
    if (!(first < last)) return first;
    swap_at(a, first, last);
    ++first;
  }
}

template <class A>
SL_HD inline int unguarded_partition_pivot(A& a, int first, int last) {
  const int mid = first + (last - first) / 2;
  move_median_to_first(a, first, first + 1, mid, last - 1);
  return unguarded_partition(a, first + 1, last, first);
}

template <class A>
SL_HD inline void unguarded_linear_insert(A& a, int last) {
  uint32_t v = a[last];
  int next = last - 1;
  while (cmp(v, a[next])) {
    a[last] = a[next];
    last = next;
    --next;
  }
  a[last] = v;
}

template <class A>
SL_HD inline void insertion_sort(A& a, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (cmp(a[i], a[first])) {
      uint32_t v = a[i];
      for (int j = i; j > first; --j) a[j] = a[j - 1];  // move_backward
      a[first] = v;
    } else {
      unguarded_linear_insert(a, i);
    }
  }
}

// ---- std::nth_element ------------------------------------------------------------------------
template <class A>
SL_HD inline void nth_element(A& a, int first, int nth, int last) {
  if (first == last || nth == last) return;
  int depth = floor_log2(last - first) * 2;
  while (last - first > 3) {
    if (depth == 0) {
      heap_select(a, first, nth + 1, last);
      swap_at(a, first, nth);
      return;
    }
    --depth;
    const int cut = unguarded_partition_pivot(a, first, last);
    if (cut <= nth) first = cut;
    else last = cut;
  }
  insertion_sort(a, first, last);
}

// ---- std::sort (introsort loop with an explicit stack + final insertion sort) ------------------
// The recursion of __introsort_loop only ever descends into disjoint sub-ranges, so the order
// in which they are processed does not change the result.
template <class A>
SL_HD inline void sort(A& a, int first, int last) {
  if (first == last) return;
  constexpr int kThreshold = 16;
  struct Frame {
    int first, last, depth;
  };
  Frame stack[48];
  int sp = 0;
  stack[sp++] = Frame{first, last, floor_log2(last - first) * 2};
  while (sp > 0) {
    Frame f = stack[--sp];
    while (f.last - f.first > kThreshold) {
      if (f.depth == 0) {
        partial_sort(a, f.first, f.last, f.last);
        break;
      }
      --f.depth;
      const int cut = unguarded_partition_pivot(a, f.first, f.last);
      if (sp < 48) stack[sp++] = Frame{cut, f.last, f.depth};
      f.last = cut;
    }
  }
  // __final_insertion_sort
  if (last - first > kThreshold) {
    insertion_sort(a, first, first + kThreshold);
    for (int i = first + kThreshold; i != last; ++i) unguarded_linear_insert(a, i);
  } else {
    insertion_sort(a, first, last);
  }
}

// The whole top-k: after the call a[0..k) hold the selected elements in torch.topk's order.
template <class A>
SL_HD inline void topk_order(A& a, int n, int k) {
  if (k <= 0) return;
  if ((int64_t)k * 64 <= (int64_t)n) {
    partial_sort(a, 0, k, n);
  } else {
    nth_element(a, 0, k - 1, n);
    sort(a, 0, k - 1);
  }
}

}  // namespace aten_order
}  // namespace sl
