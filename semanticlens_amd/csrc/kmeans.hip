// K9 — polysemanticity_score (semanticlens/scores.py:131-185).
//
// The reference loops over components on the host and runs scikit-learn
// `KMeans(n_clusters=2, n_init=10, random_state=123).fit(e)` on each (n_samples, D) block
// (scores.py:167; 6.3 ms per component), then `1 - clarity_score(centres)`.
// Because it hands sklearn a torch tensor, sklearn clusters in float64.
//
// Here one launch computes every component's n x n Gram matrix H = V V^T in fp64 (HBM-bound:
// C*n*D*4 bytes read once), and a second launch runs the whole KMeans procedure per component on
// the *centred* Gram matrix G, one wavefront per component, everything in LDS:
//   * k-means centres are always means of point subsets S_k, so every quantity sklearn computes
//     from coordinates follows from G:  ||x_i - c_k||^2 = G_ii - 2/|S_k| sum_{j in S_k} G_ij + W_k,
//     W_k = 1/|S_k|^2 sum_{j,l in S_k} G_jl;  ||c_k - c'_k||^2 likewise.
//   * the random draws of k-means++ (first centre, two candidate thresholds per init) do not depend
//     on the data: the host replays numpy's RandomState for them (semanticlens_amd/scores.py
//     kmeans_draws) and passes them in.
//   * restated from sklearn/cluster/_kmeans.py (1.7.2): _kmeans_plusplus with n_local_trials = 2,
//     _kmeans_single_lloyd (E-step with strict '<' tie rule, empty-cluster relocation to the farthest
//     point, strict-convergence / tol = mean(var) * 1e-4 tests, max_iter 300, final E-step when not
//     strictly converged), best-of-n_init by inertia unless `_is_same_clustering`.
// fp64 Gram-space arithmetic is not bit-identical to sklearn's coordinate arithmetic; decisions can
// differ only on numerical near-ties (measured agreement: see DESIGN.md).
#include "common.hpp"

namespace sl {
namespace {

constexpr int kMaxN = 128;

// ---- launch 1: H[c] = V_c V_c^T (fp64 accumulation of exact fp32 products) ----------------------
// One workgroup per component; the (n x Dc) slab is staged in LDS (rows padded by one float) and each
// thread owns pairs p = t, t + 256, ... of the upper triangle.
__global__ __launch_bounds__(256) void gram_kernel(const float* __restrict__ V, int64_t C, int n, int64_t D, int Dc,
                                                    double* __restrict__ H) {
  extern __shared__ float s_v[];  // n x (Dc + 1)
  const int ld = Dc + 1;
  const int npairs = n * (n + 1) / 2;
  constexpr int kMaxPairsPerThread = (kMaxN * (kMaxN + 1) / 2 + 255) / 256;
  for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
    const float* Vc = V + c * (int64_t)n * D;
    double acc[kMaxPairsPerThread];
#pragma unroll
    for (int a = 0; a < kMaxPairsPerThread; ++a) acc[a] = 0.0;
    for (int64_t d0 = 0; d0 < D; d0 += Dc) {
      const int dc = (int)((D - d0) < Dc ? (D - d0) : Dc);
      __syncthreads();
      for (int e = threadIdx.x; e < n * dc; e += 256) {
        const int i = e / dc, d = e % dc;
        s_v[i * ld + d] = Vc[(int64_t)i * D + d0 + d];
      }
      __syncthreads();
#pragma unroll
      for (int a = 0; a < kMaxPairsPerThread; ++a) {
        const int p = threadIdx.x + a * 256;
        if (p < npairs) {
          // p -> (i, j), i <= j, row-major upper triangle
          int i = 0, rem = p;
          while (rem >= n - i) {
            rem -= n - i;
            ++i;
          }
          const int j = i + rem;
          const float* a_ = s_v + i * ld;
          const float* b_ = s_v + j * ld;
          double s = 0.0;
          for (int d = 0; d < dc; ++d) s += (double)a_[d] * (double)b_[d];
          acc[a] += s;
        }
      }
    }
#pragma unroll
    for (int a = 0; a < kMaxPairsPerThread; ++a) {
      const int p = threadIdx.x + a * 256;
      if (p < npairs) {
        int i = 0, rem = p;
        while (rem >= n - i) {
          rem -= n - i;
          ++i;
        }
        const int j = i + rem;
        H[(c * n + i) * n + j] = acc[a];
        H[(c * n + j) * n + i] = acc[a];
      }
    }
  }
}

// ---- launch 2: KMeans(2) per component, one wave per component ------------------------------------
struct Draws {
  int n_init;
  int first[32];
  double rand[32][2];
};

__device__ inline double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ inline int wave_sum_i(int v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

struct Work {
  int n;
  double* G;       // n x n centred Gram
  double* r;       // r_i = mean_j H_ij
  double* sA;      // sum_{j in S_0} G_ij
  double* sB;      // sum_{j in S_1} G_ij
  double* tA;      // scratch for the new subsets
  double* tB;
  double* closest;
  int* label;
  int* label_old;
  int* member;     // subset defining the current centres
  int* member_new;
  int* best_label;
  int* best_member;
  double cnt[2], W[2];

  // row sums and W for subsets given by m[] -> (oA, oB, ocnt, oW); all lanes return the same scalars
  __device__ inline void subset_stats(const int* m, double* oA, double* oB, double* ocnt, double* oW, int lane) {
    double wa = 0.0, wb = 0.0;
    int ca = 0, cb = 0;
    for (int i = lane; i < n; i += kWave) {
      double a = 0.0, b = 0.0;
      const double* Gi = G + (size_t)i * n;
      for (int j = 0; j < n; ++j) {
        const double g = Gi[j];
        if (m[j] == 0) a += g;
        else b += g;
      }
      oA[i] = a;
      oB[i] = b;
      if (m[i] == 0) {
        wa += a;
        ++ca;
      } else {
        wb += b;
        ++cb;
      }
    }
    ca = wave_sum_i(ca);
    cb = wave_sum_i(cb);
    wa = wave_sum(wa);
    wb = wave_sum(wb);
    ocnt[0] = (double)ca;
    ocnt[1] = (double)cb;
    oW[0] = ca ? wa / ((double)ca * (double)ca) : 0.0;
    oW[1] = cb ? wb / ((double)cb * (double)cb) : 0.0;
  }
  // squared distance of point i to current centre k, without the G_ii term (what sklearn's E-step compares)
  __device__ inline double dpart(int i, int k) const { return W[k] - 2.0 * (k == 0 ? sA[i] : sB[i]) / cnt[k]; }
};

__global__ __launch_bounds__(64) void kmeans2_kernel(const double* __restrict__ Hall, int64_t C, int n, int64_t D,
                                                      Draws dr, int replace_empty, double* __restrict__ out,
                                                      int32_t* __restrict__ min_count) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const int64_t c = blockIdx.x;
  if (c >= C) return;
  Work w;
  w.n = n;
  double* p = reinterpret_cast<double*>(smem);
  w.G = p; p += (size_t)n * n;
  w.r = p; p += n;
  w.sA = p; p += n;
  w.sB = p; p += n;
  w.tA = p; p += n;
  w.tB = p; p += n;
  w.closest = p; p += n;
  int* q = reinterpret_cast<int*>(p);
  w.label = q; q += n;
  w.label_old = q; q += n;
  w.member = q; q += n;
  w.member_new = q; q += n;
  w.best_label = q; q += n;
  w.best_member = q; q += n;

  // H -> centred G:  G_ij = H_ij - r_i - r_j + mu   (KMeans.fit subtracts X.mean(axis=0))
  const double* H = Hall + c * (int64_t)n * n;
  double musum = 0.0;
  for (int i = lane; i < n; i += kWave) {
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += H[(size_t)i * n + j];
    w.r[i] = s / n;
    musum += s / n;
  }
  const double mu = wave_sum(musum) / n;
  __syncthreads();
  double tr = 0.0;
  for (int i = lane; i < n; i += kWave) {
    for (int j = 0; j < n; ++j) w.G[(size_t)i * n + j] = H[(size_t)i * n + j] - w.r[i] - w.r[j] + mu;
    tr += H[(size_t)i * n + i] - 2.0 * w.r[i] + mu;
  }
  __syncthreads();
  const double tol = wave_sum(tr) / ((double)n * (double)D) * 1e-4;  // _tolerance: mean(var(X, axis=0)) * tol

  double best_inertia = 0.0;
  bool have_best = false;
  for (int init = 0; init < dr.n_init; ++init) {
    // ---- k-means++ (_kmeans_plusplus, n_local_trials = 2) ----
    const int c0 = dr.first[init];
    double pot = 0.0;
    for (int i = lane; i < n; i += kWave) {
      double d = w.G[(size_t)i * n + i] + w.G[(size_t)c0 * n + c0] - 2.0 * w.G[(size_t)i * n + c0];
      d = d > 0.0 ? d : 0.0;
      w.closest[i] = d;
      pot += d;
    }
    pot = wave_sum(pot);
    __syncthreads();
    // candidates = searchsorted(cumsum(closest), rand * pot), clipped to n - 1 (sequential, lane-uniform)
    int cand[2];
    {
      const double v0 = dr.rand[init][0] * pot, v1 = dr.rand[init][1] * pot;
      int k0 = n, k1 = n;
      double cs = 0.0;
      for (int i = 0; i < n; ++i) {
        cs += w.closest[i];
        if (k0 == n && cs >= v0) k0 = i;
        if (k1 == n && cs >= v1) k1 = i;
      }
      cand[0] = k0 < n ? k0 : n - 1;
      cand[1] = k1 < n ? k1 : n - 1;
    }
    double cpot[2] = {0.0, 0.0};
    for (int t = 0; t < 2; ++t) {
      const int cc = cand[t];
      double s = 0.0;
      for (int i = lane; i < n; i += kWave) {
        double d = w.G[(size_t)i * n + i] + w.G[(size_t)cc * n + cc] - 2.0 * w.G[(size_t)i * n + cc];
        d = d > 0.0 ? d : 0.0;
        s += d < w.closest[i] ? d : w.closest[i];
      }
      cpot[t] = wave_sum(s);
    }
    const int c1 = cpot[1] < cpot[0] ? cand[1] : cand[0];  // np.argmin: first minimum

    // ---- Lloyd (_kmeans_single_lloyd) ----
    for (int i = lane; i < n; i += kWave) {
      // initial centres are the points c0, c1: S_0 = {c0}, S_1 = {c1}
      w.sA[i] = w.G[(size_t)i * n + c0];
      w.sB[i] = w.G[(size_t)i * n + c1];
      w.label_old[i] = -1;
      w.member[i] = -1;
    }
    w.cnt[0] = w.cnt[1] = 1.0;
    w.W[0] = w.G[(size_t)c0 * n + c0];
    w.W[1] = w.G[(size_t)c1 * n + c1];
    __syncthreads();
    bool strict = false;
    for (int it = 0; it < 300; ++it) {
      // E-step: nearest centre, first on ties
      int n1 = 0;
      for (int i = lane; i < n; i += kWave) {
        const int l = w.dpart(i, 1) < w.dpart(i, 0) ? 1 : 0;
        w.label[i] = l;
        w.member_new[i] = l;
        n1 += l;
      }
      n1 = wave_sum_i(n1);
      __syncthreads();
      // M-step; an empty cluster takes the point farthest from its own (old) centre
      if (n1 == 0 || n1 == n) {
        const int empty = n1 == 0 ? 1 : 0;
        double bestd = -1.0;
        int besti = 0;
        for (int i = 0; i < n; ++i) {  // lane-uniform sequential scan: first maximum
          const double d = w.G[(size_t)i * n + i] + w.dpart(i, w.label[i]);
          if (d > bestd) {
            bestd = d;
            besti = i;
          }
        }
        if (lane == 0) w.member_new[besti] = empty;
        __syncthreads();
      }
      double ncnt[2], nW[2];
      w.subset_stats(w.member_new, w.tA, w.tB, ncnt, nW, lane);
      // centre shift^2 = W_old + W_new - 2 <c_old, c_new>
      double x0 = 0.0, x1 = 0.0;
      for (int i = lane; i < n; i += kWave) {
        if (w.member_new[i] == 0) x0 += w.sA[i];
        else x1 += w.sB[i];
      }
      x0 = wave_sum(x0);
      x1 = wave_sum(x1);
      const double shift = (w.W[0] + nW[0] - 2.0 * x0 / (w.cnt[0] * ncnt[0])) + (w.W[1] + nW[1] - 2.0 * x1 / (w.cnt[1] * ncnt[1]));
      int same = 1;
      __syncthreads();
      for (int i = lane; i < n; i += kWave) {
        w.sA[i] = w.tA[i];
        w.sB[i] = w.tB[i];
        w.member[i] = w.member_new[i];
        same &= (w.label[i] == w.label_old[i]);
      }
      w.cnt[0] = ncnt[0]; w.cnt[1] = ncnt[1];
      w.W[0] = nW[0]; w.W[1] = nW[1];
      same = __all(same);
      __syncthreads();
      if (same) {
        strict = true;
        break;
      }
      if (shift <= tol) break;
      for (int i = lane; i < n; i += kWave) w.label_old[i] = w.label[i];
      __syncthreads();
    }
    if (!strict) {  // rerun the E-step so labels match the final centres
      for (int i = lane; i < n; i += kWave) w.label[i] = w.dpart(i, 1) < w.dpart(i, 0) ? 1 : 0;
      __syncthreads();
    }
    double inertia = 0.0;
    for (int i = lane; i < n; i += kWave) inertia += w.G[(size_t)i * n + i] + w.dpart(i, w.label[i]);
    inertia = wave_sum(inertia);

    // ---- best of n_init (KMeans.fit): better inertia AND not the same clustering ----
    bool take = !have_best;
    if (have_best && inertia < best_inertia) {
      // _is_same_clustering: every label of run A maps to a single label of run B
      int map0 = -1, map1 = -1;
      bool same_clu = true;
      for (int i = 0; i < n && same_clu; ++i) {  // lane-uniform sequential
        const int a = w.label[i], b = w.best_label[i];
        int& m = a == 0 ? map0 : map1;
        if (m == -1) m = b;
        else if (m != b) same_clu = false;
      }
      take = !same_clu;
    }
    if (take) {
      best_inertia = inertia;
      have_best = true;
      for (int i = lane; i < n; i += kWave) {
        w.best_label[i] = w.label[i];
        w.best_member[i] = w.member[i];
      }
    }
    __syncthreads();
  }

  // ---- scores.py:168-185 on the chosen clustering ----
  // centres c_k = mean_{j in S_k} x_j;  x_i . x_j = H_ij
  double cA = 0.0, cB = 0.0, dAA = 0.0, dBB = 0.0, dAB = 0.0;
  int l0 = 0, l1 = 0;
  for (int i = lane; i < n; i += kWave) {
    const int mi = w.best_member[i];
    (mi == 0 ? cA : cB) += 1.0;
    (w.best_label[i] == 0 ? l0 : l1) += 1;
    double a = 0.0, b = 0.0;
    for (int j = 0; j < n; ++j) {
      const double h = H[(size_t)i * n + j];
      if (w.best_member[j] == 0) a += h;
      else b += h;
    }
    if (mi == 0) {
      dAA += a;
      dAB += b;
    } else {
      dBB += b;
    }
  }
  cA = wave_sum(cA); cB = wave_sum(cB);
  dAA = wave_sum(dAA); dBB = wave_sum(dBB); dAB = wave_sum(dAB);
  l0 = wave_sum_i(l0); l1 = wave_sum_i(l1);
  const int mc = l0 < l1 ? l0 : l1;
  // clarity_score of the two centres (n = 2):  2 * sum(((c1^ + c2^) / 2)^2) - 1 with c^ = c / max(||c||, 1e-12)
  const double nA2 = cA > 0 ? dAA / (cA * cA) : 0.0, nB2 = cB > 0 ? dBB / (cB * cB) : 0.0;
  const double dot = (cA > 0 && cB > 0) ? dAB / (cA * cB) : 0.0;
  const double na = sqrt(nA2 > 0 ? nA2 : 0.0), nb = sqrt(nB2 > 0 ? nB2 : 0.0);
  const double ia = 1.0 / (na > 1e-12 ? na : 1e-12), ib = 1.0 / (nb > 1e-12 ? nb : 1e-12);
  const double msq = (nA2 * ia * ia + nB2 * ib * ib + 2.0 * dot * ia * ib) / 4.0;
  double poly = 1.0 - ((msq - 0.5) / 1.0 * 2.0);
  if (replace_empty && mc < 2) {
    // fallback (scores.py:173-184): 1 - mean_{i < min(10,n)} clarity([mean_j v_j, v_i]), clarity in fp32 there
    const int ns = n < 10 ? n : 10;
    const double nm = sqrt(mu > 0 ? mu : 0.0);
    const double im = 1.0 / (nm > 1e-12 ? nm : 1e-12);
    double acc = 0.0;
    for (int i = 0; i < ns; ++i) {  // lane-uniform
      const double hii = H[(size_t)i * n + i];
      const double ni = sqrt(hii > 0 ? hii : 0.0);
      const double ii = 1.0 / (ni > 1e-12 ? ni : 1e-12);
      const double m2 = (mu * im * im + hii * ii * ii + 2.0 * w.r[i] * im * ii) / 4.0;
      acc += (m2 - 0.5) * 2.0;
    }
    poly = 1.0 - acc / ns;
  }
  if (lane == 0) {
    out[c] = poly;
    if (min_count) min_count[c] = mc;
  }
}

size_t kmeans_smem_bytes(int64_t n) { return (size_t)n * n * 8 + (size_t)n * 6 * 8 + (size_t)n * 6 * 4 + 64; }

}  // namespace
}  // namespace sl

using namespace sl;

SL_API size_t sl_poly2means_ws_bytes(int64_t C, int64_t n, int64_t D) {
  (void)D;
  return (size_t)C * (size_t)n * (size_t)n * 8 + 256;
}

SL_API int sl_poly2means(const float* d_V, int64_t C, int64_t n, int64_t D, const int32_t* h_first_center, int n_init,
                         const double* h_rand, int replace_empty_clusters, double* d_out, int32_t* d_min_count,
                         void* d_ws, size_t ws_bytes, void* stream) {
  SL_REQUIRE(C >= 0 && n >= 0 && D >= 0, "sl_poly2means: negative shape");
  if (C == 0) return 0;
  SL_REQUIRE(n >= 2, "sl_poly2means: n_samples=%lld should be >= n_clusters=2.", (long long)n);  // sklearn's ValueError
  SL_REQUIRE(n <= kMaxN, "sl_poly2means: n_samples=%lld exceeds the supported maximum %d", (long long)n, kMaxN);
  SL_REQUIRE(n_init >= 1 && n_init <= 32, "sl_poly2means: n_init=%d not in [1, 32]", n_init);
  SL_REQUIRE(d_V && d_out && h_first_center && h_rand, "sl_poly2means: null pointer");
  SL_REQUIRE(d_ws && ws_bytes >= sl_poly2means_ws_bytes(C, n, D), "sl_poly2means: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  double* H = reinterpret_cast<double*>(((uintptr_t)d_ws + 255) & ~(uintptr_t)255);
  Draws dr;
  dr.n_init = n_init;
  for (int i = 0; i < n_init; ++i) {
    SL_REQUIRE(h_first_center[i] >= 0 && h_first_center[i] < n, "sl_poly2means: first centre %d out of range", h_first_center[i]);
    dr.first[i] = h_first_center[i];
    dr.rand[i][0] = h_rand[2 * i];
    dr.rand[i][1] = h_rand[2 * i + 1];
  }
  {
    ProfScope prof(SL_PROF_SCORES, st, (double)C * n * D * 4);
    int Dc = (int)(48 * 1024 / (4 * n)) - 1;
    if (Dc > D) Dc = (int)D;
    if (Dc < 1) Dc = 1;
    int64_t blocks = C;
    const int64_t cap = (int64_t)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    SL_LAUNCH(prof, gram_kernel, dim3((unsigned)blocks), dim3(256), (size_t)n * (Dc + 1) * 4, st, d_V, C, (int)n, D, Dc, H);
  }
  const size_t smem = kmeans_smem_bytes(n);
  if (smem > 64 * 1024)
    SL_CHECK_HIP(hipFuncSetAttribute((const void*)kmeans2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipLaunchKernelGGL(kmeans2_kernel, dim3((unsigned)C), dim3(64), smem, st, (const double*)H, C, (int)n, D, dr,
                     replace_empty_clusters, d_out, d_min_count);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}
