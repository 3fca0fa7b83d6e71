// Host-side check of semanticlens_amd/csrc/aten_topk_order.hpp against libstdc++ itself
// (std::partial_sort / std::nth_element / std::sort with ATen's comparator, the exact calls
// torch.topk's CPU kernel makes).  Built and run by tests/test_aten_order_host.py with g++.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <utility>
#include <vector>

#include "../../semanticlens_amd/csrc/aten_topk_order.hpp"
#include "../../semanticlens_amd/csrc/aten_topk_wave.hpp"

static uint32_t key_of(float f) {  // same map as common.hpp bf16_order_key, on the bf16 bits of f
  uint32_t u;
  std::memcpy(&u, &f, 4);
  uint16_t h = (uint16_t)(u >> 16);
  uint32_t mag = h & 0x7FFFu;
  if (mag > 0x7F80u) return 0xFFFFu;
  return (h & 0x8000u) ? (0x8000u - mag) : (0x8000u + mag);
}

static bool run_case(const std::vector<float>& vals, int k, long& n_partial, long& n_nth) {
  const int n = (int)vals.size();
  using elem = std::pair<float, int64_t>;
  std::vector<elem> q(n);
  for (int j = 0; j < n; ++j) q[j] = {vals[j], j};
  auto cmp = [](const elem& x, const elem& y) {
    return ((std::isnan(x.first) && !std::isnan(y.first)) || (x.first > y.first));
  };
  if (k > 0) {
    if ((int64_t)k * 64 <= n) {
      std::partial_sort(q.begin(), q.begin() + k, q.end(), cmp);
      ++n_partial;
    } else {
      std::nth_element(q.begin(), q.begin() + k - 1, q.end(), cmp);
      std::sort(q.begin(), q.begin() + k - 1, cmp);
      ++n_nth;
    }
  }
  std::vector<uint32_t> a(n);
  for (int j = 0; j < n; ++j) a[j] = (key_of(vals[j]) << 16) | (uint32_t)j;
  uint32_t* p = a.data();
  sl::aten_order::topk_order(p, n, k);
  for (int j = 0; j < k; ++j)
    if ((int64_t)(a[j] & 0xFFFF) != q[j].second) return false;
  // the data-parallel restatement (aten_topk_wave.hpp: Hoare partition by its L / R lists, insertion sorts as stable rank
  // sorts) must leave the SAME first k elements — it is what actmax_aten.hip evaluates with one wavefront per row
  if (k > 0 && (int64_t)k * 64 > n) {
    std::vector<uint32_t> b(n), tmp(n);
    std::vector<int> tab(2 * n + 2);
    for (int j = 0; j < n; ++j) b[j] = (key_of(vals[j]) << 16) | (uint32_t)j;
    uint32_t* pb = b.data();
    sl::aten_order::lists::topk_order_nth(pb, n, k, tab.data(), tmp.data());
    for (int j = 0; j < k; ++j)
      if ((b[j] & 0xFFFF) != (a[j] & 0xFFFF)) return false;
  }
  return true;
}

static float bf16_round(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
  std::memcpy(&f, &u, 4);
  return f;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 20000;
  std::mt19937 rng(12345);
  long bad = 0, n_partial = 0, n_nth = 0, total = 0;
  const int ks[] = {1, 2, 3, 5, 17, 20, 33, 100, 257};
  const int Bs[] = {1, 2, 7, 32, 64, 100, 256, 1500, 7000};
  for (int it = 0; it < iters; ++it) {
    int k = ks[rng() % 9], B = Bs[rng() % 9];
    if (it % 50 != 0 && B > 300) B = 64;  // keep the big ones rare
    int n = k + B;
    std::vector<float> v(n);
    int kind = rng() % 6;
    for (int j = 0; j < n; ++j) {
      float x;
      switch (kind) {
        case 0: x = bf16_round(std::normal_distribution<float>(0, 1)(rng)); break;          // typical
        case 1: x = (float)((int)(rng() % 9) - 2) / 4.f; break;                             // heavy ties
        case 2: x = 0.f; break;                                                             // all equal
        case 3: x = (float)j; break;                                                        // ascending
        case 4: x = (float)(n - j); break;                                                  // descending
        default: x = (rng() % 13 == 0) ? NAN : bf16_round((float)(rng() % 50) / 8.f - 1.f); // NaNs + ties
      }
      if (kind == 2 && j % 3 == 0) x = -0.f;
      v[j] = std::isnan(x) ? x : bf16_round(x);  // the device only ever sees bf16 values
    }
    if (kind == 0 && it % 7 == 0)  // sorted state prefix like the real [state | batch] layout
      std::sort(v.begin(), v.begin() + k, std::greater<float>());
    if (!run_case(v, k, n_partial, n_nth)) ++bad;
    ++total;
  }
  // median-of-3 killer style inputs push introselect/introsort into their heap fallbacks
  for (int n : {64, 200, 1000, 4096}) {
    std::vector<float> v(n);
    int half = n / 2;
    for (int i = 0; i < half; ++i) {
      v[i] = (i % 2 == 0) ? (float)(i + 1) : (float)(half + i + (half % 2 == 0 ? 0 : 1));
      v[half + i] = (float)((i + 1) * 2);
    }
    for (auto& x : v) x = bf16_round(x);
    for (int k : {2, 20, n / 3, n - 1, n}) {
      if (!run_case(v, k, n_partial, n_nth)) ++bad;
      std::vector<float> w(v.rbegin(), v.rend());
      if (!run_case(w, k, n_partial, n_nth)) ++bad;
      total += 2;
    }
  }
  std::printf("cases=%ld partial_sort=%ld nth_element=%ld mismatches=%ld\n", total, n_partial, n_nth, bad);
  return bad == 0 ? 0 : 1;
}
