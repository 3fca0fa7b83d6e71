// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of the SemanticLens concept-DB hot path (reference @ v0.2.1,
// /root/reference/semanticlens/...).  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load this library; the product
// (semanticlens_amd/) never does and fails loudly without its HIP library.
//
// Parity status: PINNED.  tests/test_oracle_golden.py checks every function
// here against tests/golden/*.npz, which tests/golden/make_golden.py produced
// by running the unmodified reference in the build container (plus the one
// known-answer vector the reference's own tests hold,
// tests/component_visualization/test_activation_caching.py:14-30).
//
// Each function cites the reference file:line it restates.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <utility>
#include <vector>

#define ORC_API extern "C" __attribute__((visibility("default")))

namespace {

inline uint32_t f32_bits(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}
inline float bits_f32(uint32_t u) {
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

// fp32 -> bf16, round-to-nearest-even, NaN -> 0x7FC0.  This is what
// `acts.T.to(torch.bfloat16)` does (activation_caching.py:133; c10::BFloat16
// round_to_nearest_even).
inline uint16_t f32_to_bf16(float f) {
  if (std::isnan(f)) return 0x7FC0;
  uint32_t u = f32_bits(f);
  uint32_t rounding_bias = ((u >> 16) & 1u) + 0x7FFFu;
  return (uint16_t)((u + rounding_bias) >> 16);
}
inline float bf16_to_f32(uint16_t h) { return bits_f32((uint32_t)h << 16); }

// ATen's top-k comparator for `largest=True` (NaN sorts first):
// aten/src/ATen/native/cpu/TopKImpl.h — called by torch.topk at
// activation_caching.py:140.
struct AtenGreater {
  bool operator()(const std::pair<float, int64_t>& x, const std::pair<float, int64_t>& y) const {
    return ((std::isnan(x.first) && !std::isnan(y.first)) || (x.first > y.first));
  }
};

// The build's deterministic total order: value descending (NaN first, -0 == +0),
// then sample id ascending (the -1 sentinel therefore wins ties).  Equals a
// *stable* top-k on the reference's concatenation [state | batch] whenever ids
// grow with position, which is how the reference numbers samples
// (activation_caching.py:410-413).
struct TotalBetter {
  bool operator()(const std::pair<float, int64_t>& x, const std::pair<float, int64_t>& y) const {
    bool xn = std::isnan(x.first), yn = std::isnan(y.first);
    if (xn != yn) return xn;
    if (!xn && x.first != y.first) return x.first > y.first;
    return x.second < y.second;
  }
};

}  // namespace

ORC_API void orc_f32_to_bf16(const float* x, uint16_t* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = f32_to_bf16(x[i]);
}

// ---------------------------------------------------------------------------
// Aggregators — aggregators.py:38-61 (conv mean), :64-87 (conv max).
// in (B,C,S) contiguous fp32, out (B,C) fp32.  agg: 0 = max (amax, NaN
// propagates), 1 = mean.  The mean is accumulated in float64 and rounded once
// (torch's fp32 cascade sum may differ by 1 fp32 ulp).
// ---------------------------------------------------------------------------
ORC_API void orc_agg_conv(const float* x, int64_t B, int64_t C, int64_t S, int agg, float* out) {
  const int64_t R = B * C;
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < R; ++r) {
    const float* p = x + r * S;
    if (agg == 0) {
      float m = -std::numeric_limits<float>::infinity();
      bool nan = false;
      for (int64_t i = 0; i < S; ++i) {
        nan |= std::isnan(p[i]);
        m = p[i] > m ? p[i] : m;
      }
      out[r] = nan ? std::numeric_limits<float>::quiet_NaN() : m;
    } else {
      // agg 1: mean; agg 2: plain sum (zennit-crp ChannelConcept.reference_sampling, max_target="sum", which the
      // reference's relevance visualizer configures: relevance_based.py:111,123-129)
      double s = 0.0;
      for (int64_t i = 0; i < S; ++i) s += (double)p[i];
      out[r] = agg == 2 ? (float)s : (float)(s / (double)S);
    }
  }
}

// crp ChannelConcept.reference_sampling with abs_norm=True: rel / (|rel|.sum(-1) + 1e-10) per sample, in fp32
// (the sum over channels in float64 then rounded: exact for the integer-valued test data).
ORC_API void orc_abs_norm_rows(float* x, int64_t B, int64_t C, float eps) {
  for (int64_t b = 0; b < B; ++b) {
    double s = 0.0;
    for (int64_t c = 0; c < C; ++c) s += std::fabs((double)x[b * C + c]);
    const float tot = (float)s + eps;
    for (int64_t c = 0; c < C; ++c) x[b * C + c] = x[b * C + c] / tot;
  }
}

// Token aggregators — aggregators.py:90-114 (mean), :117-141 (absmean),
// :144-168 (max), :171-195 (absmax), :198-244 (special token).
// in (B,T,F) contiguous fp32, out (B,F).  agg: 0 mean, 1 absmean, 2 max,
// 3 absmax, 4 token `pos` (python-style negative index allowed).
ORC_API void orc_agg_tokens(const float* x, int64_t B, int64_t T, int64_t F, int agg, int64_t pos, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; ++b) {
    const float* xb = x + b * T * F;
    for (int64_t f = 0; f < F; ++f) {
      if (agg == 4) {
        int64_t t = pos < 0 ? pos + T : pos;
        out[b * F + f] = xb[t * F + f];
        continue;
      }
      const bool is_abs = (agg == 1 || agg == 3);
      if (agg == 0 || agg == 1) {
        double s = 0.0;
        for (int64_t t = 0; t < T; ++t) {
          float v = xb[t * F + f];
          s += (double)(is_abs ? std::fabs(v) : v);
        }
        out[b * F + f] = (float)(s / (double)T);
      } else {
        float m = -std::numeric_limits<float>::infinity();
        bool nan = false;
        for (int64_t t = 0; t < T; ++t) {
          float v = xb[t * F + f];
          if (is_abs) v = std::fabs(v);
          nan |= std::isnan(v);
          m = v > m ? v : m;
        }
        out[b * F + f] = nan ? std::numeric_limits<float>::quiet_NaN() : m;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// ActMax — activation_caching.py:101-141.
// State: vals (C,k) bf16 bit patterns, ids (C,k) int64.
// ---------------------------------------------------------------------------
ORC_API void orc_actmax_init(uint16_t* vals, int64_t* ids, int64_t C, int64_t k) {
  // activation_caching.py:108-109: -zeros (= -0.0 = 0x8000) and -ones.
  for (int64_t i = 0; i < C * k; ++i) {
    vals[i] = 0x8000;
    ids[i] = -1;
  }
}

// One ActMax.update (activation_caching.py:112-141).
// acts (B,C) fp32, sample_ids (B).  mode 0: torch.topk's CPU tie order
// (libstdc++ partial_sort / nth_element + sort with ATen's comparator, on the
// concatenation [state | batch]); mode 1: the build's total order.
ORC_API void orc_actmax_update(uint16_t* vals, int64_t* ids, int64_t C, int64_t k, const float* acts,
                               const int64_t* sample_ids, int64_t B, int mode) {
  if (k == 0) return;  // TopKImpl.h: k == 0 -> empty outputs
  const int64_t n = k + B;
#pragma omp parallel
  {
    std::vector<std::pair<float, int64_t>> q(n);
    std::vector<uint16_t> all_v(n);
    std::vector<int64_t> all_i(n);
#pragma omp for schedule(static)
    for (int64_t c = 0; c < C; ++c) {
      // all_acts = cat([state, batch_acts], 1) — :137-138
      for (int64_t j = 0; j < k; ++j) {
        all_v[j] = vals[c * k + j];
        all_i[j] = ids[c * k + j];
      }
      for (int64_t b = 0; b < B; ++b) {
        all_v[k + b] = f32_to_bf16(acts[b * C + c]);  // :133
        all_i[k + b] = sample_ids[b];                 // :134
      }
      if (mode == 0) {
        for (int64_t j = 0; j < n; ++j) q[j] = {bf16_to_f32(all_v[j]), j};
        AtenGreater cmp;
        if (k * 64 <= n) {
          std::partial_sort(q.begin(), q.begin() + k, q.end(), cmp);
        } else {
          std::nth_element(q.begin(), q.begin() + (k - 1), q.end(), cmp);
          std::sort(q.begin(), q.begin() + (k - 1), cmp);
        }
        for (int64_t j = 0; j < k; ++j) {  // :140-141 (values, gather of ids)
          vals[c * k + j] = all_v[q[j].second];
          ids[c * k + j] = all_i[q[j].second];
        }
      } else {
        for (int64_t j = 0; j < n; ++j) q[j] = {bf16_to_f32(all_v[j]), all_i[j]};
        // carry the position so the original bit pattern (sign of zero) is kept
        std::vector<int64_t> order(n);
        for (int64_t j = 0; j < n; ++j) order[j] = j;
        TotalBetter better;
        std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b2) { return better(q[a], q[b2]); });
        for (int64_t j = 0; j < k; ++j) {
          vals[c * k + j] = all_v[order[j]];
          ids[c * k + j] = all_i[order[j]];
        }
      }
    }
  }
}

// Cross-rank merge of R per-rank states into `vals/ids` under the total order
// (no reference counterpart: SURVEY.md §8e, K4).  other_* are (R,C,k).
ORC_API void orc_actmax_merge_states(uint16_t* vals, int64_t* ids, int64_t C, int64_t k, const uint16_t* other_vals,
                                     const int64_t* other_ids, int64_t R) {
  if (k == 0) return;
  const int64_t n = k * (R + 1);
  std::vector<std::pair<float, int64_t>> q(n);
  std::vector<uint16_t> all_v(n);
  std::vector<int64_t> order(n);
  TotalBetter better;
  for (int64_t c = 0; c < C; ++c) {
    for (int64_t j = 0; j < k; ++j) {
      all_v[j] = vals[c * k + j];
      q[j] = {bf16_to_f32(all_v[j]), ids[c * k + j]};
    }
    for (int64_t r = 0; r < R; ++r)
      for (int64_t j = 0; j < k; ++j) {
        int64_t o = (r * C + c) * k + j;
        all_v[k * (r + 1) + j] = other_vals[o];
        q[k * (r + 1) + j] = {bf16_to_f32(other_vals[o]), other_ids[o]};
      }
    for (int64_t j = 0; j < n; ++j) order[j] = j;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return better(q[a], q[b]); });
    // real entries are unique by id; sentinels (id -1) are interchangeable
    for (int64_t j = 0; j < k; ++j) {
      vals[c * k + j] = all_v[order[j]];
      ids[c * k + j] = q[order[j]].second;
    }
  }
}

// embeds[sample_ids] — activation_based.py:387-390; negative ids wrap
// (python/torch advanced indexing), so the -1 sentinel reads row N-1.
ORC_API int orc_gather_rows(const float* emb, int64_t N, int64_t D, const int64_t* ids, int64_t n_ids, float* out) {
  for (int64_t i = 0; i < n_ids; ++i) {
    int64_t r = ids[i];
    if (r < 0) r += N;
    if (r < 0 || r >= N) return -1;  // torch raises IndexError
    std::memcpy(out + i * D, emb + r * D, (size_t)D * 4);
  }
  return 0;
}

// ---------------------------------------------------------------------------
// scores.py
// ---------------------------------------------------------------------------
namespace {
// torch.nn.functional.normalize(x, dim=-1): x / max(||x||_2, 1e-12)
inline double inv_norm(const float* v, int64_t D, double eps) {
  double s = 0.0;
  for (int64_t i = 0; i < D; ++i) s += (double)v[i] * (double)v[i];
  double nrm = std::sqrt(s);
  return 1.0 / (nrm > eps ? nrm : eps);
}
}  // namespace

// similarity_score — scores.py:84-128.  Returns the branch taken:
//   0: shapes equal -> row-wise cosine_similarity, out (xr,)            (:127)
//   1: x.shape[1] == y.shape[0] -> normalize(x) @ normalize(y), out (xr,yc)  (:122-123)
//   2: x.shape[1] == y.shape[1] -> normalize(x) @ normalize(y).T, out (xr,yr) (:124-125)
//  -1: ValueError
ORC_API int orc_similarity(const float* x, int64_t xr, int64_t xc, const float* y, int64_t yr, int64_t yc,
                           float* out) {
  if (xr == yr && xc == yc) {
    // F.cosine_similarity(x, y, dim=-1, eps=1e-8): each norm clamped to eps
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < xr; ++i) {
      double d = 0.0;
      for (int64_t j = 0; j < xc; ++j) d += (double)x[i * xc + j] * (double)y[i * xc + j];
      out[i] = (float)(d * inv_norm(x + i * xc, xc, 1e-8) * inv_norm(y + i * yc, yc, 1e-8));
    }
    return 0;
  }
  std::vector<double> rx(xr), ry(yr);
  for (int64_t i = 0; i < xr; ++i) rx[i] = inv_norm(x + i * xc, xc, 1e-12);
  for (int64_t i = 0; i < yr; ++i) ry[i] = inv_norm(y + i * yc, yc, 1e-12);
  if (xc == yr) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < xr; ++i)
      for (int64_t j = 0; j < yc; ++j) {
        double d = 0.0;
        for (int64_t t = 0; t < xc; ++t) d += (double)x[i * xc + t] * rx[i] * (double)y[t * yc + j] * ry[t];
        out[i * yc + j] = (float)d;
      }
    return 1;
  }
  if (xc == yc) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < xr; ++i)
      for (int64_t j = 0; j < yr; ++j) {
        double d = 0.0;
        const float* a = x + i * xc;
        const float* b = y + j * yc;
        for (int64_t t = 0; t < xc; ++t) d += (double)a[t] * (double)b[t];
        out[i * yr + j] = (float)(d * rx[i] * ry[j]);
      }
    return 2;
  }
  return -1;
}

// clarity_score — scores.py:18-47.  V (C,n,D) -> (C,)
//   ((mean_j normalize(v_j))^2 .sum - 1/n) / (n-1) * n
ORC_API void orc_clarity(const float* V, int64_t C, int64_t n, int64_t D, float* out) {
#pragma omp parallel
  {
    std::vector<double> m(D);
#pragma omp for schedule(static)
    for (int64_t c = 0; c < C; ++c) {
      std::fill(m.begin(), m.end(), 0.0);
      for (int64_t j = 0; j < n; ++j) {
        const float* v = V + (c * n + j) * D;
        double r = inv_norm(v, D, 1e-12);
        for (int64_t d = 0; d < D; ++d) m[d] += (double)v[d] * r;
      }
      double s = 0.0;
      for (int64_t d = 0; d < D; ++d) {
        double md = m[d] / (double)n;
        s += md * md;
      }
      out[c] = (float)((s - 1.0 / (double)n) / (double)(n - 1) * (double)n);
    }
  }
}

// redundancy_score — scores.py:50-81.  cones (Bt,C,D) -> (Bt,) ; a 2-D input is Bt = 1.
//   sims = normalize(c) @ normalize(c)^T - 2*I ; max over last dim ; mean over rows
ORC_API void orc_redundancy(const float* V, int64_t Bt, int64_t C, int64_t D, float* out) {
  for (int64_t b = 0; b < Bt; ++b) {
    const float* X = V + b * C * D;
    std::vector<double> r(C);
    for (int64_t i = 0; i < C; ++i) r[i] = inv_norm(X + i * D, D, 1e-12);
    double acc = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : acc)
    for (int64_t i = 0; i < C; ++i) {
      double best = -std::numeric_limits<double>::infinity();
      for (int64_t j = 0; j < C; ++j) {
        double d = 0.0;
        for (int64_t t = 0; t < D; ++t) d += (double)X[i * D + t] * (double)X[j * D + t];
        d *= r[i] * r[j];
        if (i == j) d -= 2.0;
        best = d > best ? d : best;
      }
      acc += best;
    }
    out[b] = (float)(acc / (double)C);
  }
}

// Template-difference averaging of text embeddings — lens.py:196-199, keeping
// the reference's grouping quirk (SURVEY.md finding 4): E is built
// template-major ([t.format(q) for t in templates for q in query], :174) but
// reshaped "(q t) d -> q t d", i.e. row q*T + t is read as (query q, template t).
ORC_API void orc_template_mean(const float* E, const float* E0, int64_t Q, int64_t T, int64_t D, float* out) {
  for (int64_t q = 0; q < Q; ++q)
    for (int64_t d = 0; d < D; ++d) {
      // torch: (E.reshape(Q,T,D) - E0[None]).mean(1) in fp32
      float s = 0.f;
      for (int64_t t = 0; t < T; ++t) s += E[(q * T + t) * D + d] - E0[t * D + d];
      out[q * D + d] = s / (float)T;
    }
}
