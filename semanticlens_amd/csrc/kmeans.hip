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
// differ only on numerical near-ties.  One kind is structural, not a coincidence: two mutually nearest outliers j1, j2
// (nobody else closer to them than the centres chosen so far) have k-means++ potentials S + d(j1, j2) EACH, so which of
// the two scikit-learn takes hangs on the rounding of its BLAS distances (tools/k9_postmortem.py; ~1 in 1 500 components
// of tiny random inputs ends in another clustering for it: profiles/r05_k9_near_tie.txt).
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
__device__ inline double wave_max(double v) {
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
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

// ---- general path: any n_clusters (<= kMaxClusters), n_samples up to kMaxNGeneral ------------------------------------
// The same restatement with k member sets.  State that does not fit LDS for every (n, k) lives in a per-component
// slice of the workspace (L2-resident: one wavefront walks it): the centred Gram matrix, sum_{l in S_j} G_il for the
// current and the next member sets, the k-means++ distance arrays, labels.  LDS holds the per-cluster scalars and a
// (k x 64) pad of per-lane accumulators.  scikit-learn specifics restated (sklearn/cluster/_kmeans.py, 1.7.2):
//   * _kmeans_plusplus: n_local_trials = 2 + int(ln k) candidates per further centre, np.searchsorted on the cumulative
//     closest distances, np.argmin (first minimum) over the candidates' potentials;
//   * E-step: first minimum of ||c_j||^2 - 2 x_i.c_j; empty clusters take, in ascending cluster id, the points farthest
//     from their own centre (descending distance) and those points leave their old cluster (_relocate_empty_clusters_dense);
//     nothing moves when the largest distance is 0 (its early return).  Caveats: with two or more empty clusters sklearn
//     hands out the far points in np.argpartition's order, which is not sorted among the n_empty farthest — the SET of
//     relocated points is the same, which cluster id gets which may differ (the score is symmetric in the ids); and
//     "distance == 0" is tested in Gram space here, in feature space there (a centre of >= 3 duplicates may round off it);
//   * strict convergence (labels unchanged) / tol = mean(var) * 1e-4 on the summed squared centre shifts / 300 iterations,
//     a final E-step when not strictly converged, best of n_init by inertia unless _is_same_clustering.
// Checked against scikit-learn on the host first (a numpy transcription of this kernel: 460 components, k = 2..6, with
// duplicated points that force relocations — every clustering identical) and on the device by tests/test_gpu_parity.py.
constexpr int kMaxClusters = 16, kMaxNGeneral = 1024, kMaxTrials = 4;

struct GenWs {  // offsets (in doubles / ints) of one component's slice
  double *G, *s, *sn, *closest, *cand;  // n*n, k*n, k*n, n, trials*n
  int *label, *label_old, *member, *best_label, *best_member;
};
__host__ __device__ inline size_t gen_ws_doubles(int64_t n, int64_t k) { return (size_t)(n * n + 2 * k * n + n + kMaxTrials * n); }
__host__ __device__ inline size_t gen_ws_bytes_per_component(int64_t n, int64_t k) {
  return ((gen_ws_doubles(n, k) * 8 + (size_t)5 * n * 4) + 255) & ~(size_t)255;
}

__global__ __launch_bounds__(64) void kmeansk_kernel(const double* __restrict__ Hall, int64_t C, int n, int64_t D, int k, int trials,
                                                      int n_init, const int32_t* __restrict__ first, const double* __restrict__ rnd,
                                                      int replace_empty, unsigned char* __restrict__ ws_all, size_t ws_stride,
                                                      double* __restrict__ out, int32_t* __restrict__ min_count) {
  __shared__ double s_acc[kMaxClusters * kWave];  // per-lane accumulators, one column per lane
  __shared__ double s_cnt[kMaxClusters], s_W[kMaxClusters], s_ncnt[kMaxClusters], s_nW[kMaxClusters], s_x[kMaxClusters];
  __shared__ double s_dot[kMaxClusters * kMaxClusters];
  __shared__ int s_seed[kMaxClusters], s_count[kMaxClusters], s_map[kMaxClusters];
  const int lane = threadIdx.x;
  const int64_t c = blockIdx.x;
  if (c >= C) return;
  unsigned char* base = ws_all + c * ws_stride;
  GenWs w;
  double* p = reinterpret_cast<double*>(base);
  w.G = p; p += (size_t)n * n;
  w.s = p; p += (size_t)k * n;
  w.sn = p; p += (size_t)k * n;
  w.closest = p; p += n;
  w.cand = p; p += (size_t)kMaxTrials * n;
  int* q = reinterpret_cast<int*>(p);
  w.label = q; q += n;
  w.label_old = q; q += n;
  w.member = q; q += n;
  w.best_label = q; q += n;
  w.best_member = q; q += n;
  const double* H = Hall + c * (int64_t)n * n;

  // H -> centred G; r_i kept in w.cand[3n..4n) is not safe (cand is reused): recompute r_i where needed from H
  double musum = 0.0;
  for (int i = lane; i < n; i += kWave) {
    double t = 0.0;
    for (int j = 0; j < n; ++j) t += H[(size_t)i * n + j];
    w.closest[i] = t / n;  // r_i, parked until G is built
    musum += t / n;
  }
  const double mu = wave_sum(musum) / n;
  __syncthreads();
  double tr = 0.0;
  for (int i = lane; i < n; i += kWave) {
    const double ri = w.closest[i];
    for (int j = 0; j < n; ++j) w.G[(size_t)i * n + j] = H[(size_t)i * n + j] - ri - w.closest[j] + mu;
    tr += H[(size_t)i * n + i] - 2.0 * ri + mu;
  }
  __syncthreads();
  const double tol = wave_sum(tr) / ((double)n * (double)D) * 1e-4;
  auto Gd = [&](int i) { return w.G[(size_t)i * n + i]; };
  auto dist_to = [&](int i, int cc) {
    double d = Gd(i) + Gd(cc) - 2.0 * w.G[(size_t)i * n + cc];
    return d > 0.0 ? d : 0.0;
  };
  // squared distance of point i to current centre j without the G_ii term
  // (a cluster left empty — relocation skipped, below — is never the first minimum: sklearn parks its centre on the biggest
  // cluster's, a tie the lower-numbered of the two wins without changing the partition)
  auto dpart = [&](const double* sarr, int i, int j) {
    return s_cnt[j] > 0 ? s_W[j] - 2.0 * sarr[(size_t)j * n + i] / s_cnt[j] : __builtin_huge_val();
  };
  // member sets m[] -> sums so[j][i] = sum_{l in S_j} G_il, counts, W_j (into s_ncnt / s_nW)
  auto subset_stats = [&](const int* m, double* so) {
    for (int j = lane; j < k; j += kWave) s_count[j] = 0;
    __syncthreads();
    for (int i = lane; i < n; i += kWave) atomicAdd(&s_count[m[i]], 1);
    for (int i0 = 0; i0 < n; i0 += kWave) {
      const int i = i0 + lane;
      for (int j = 0; j < k; ++j) s_acc[j * kWave + lane] = 0.0;
      if (i < n) {
        const double* Gi = w.G + (size_t)i * n;
        for (int l = 0; l < n; ++l) s_acc[m[l] * kWave + lane] += Gi[l];
        for (int j = 0; j < k; ++j) so[(size_t)j * n + i] = s_acc[j * kWave + lane];
      }
    }
    __syncthreads();
    for (int j = 0; j < k; ++j) {  // W_j = sum_{i in S_j} so[j][i] / |S_j|^2
      double t = 0.0;
      for (int i = lane; i < n; i += kWave)
        if (m[i] == j) t += so[(size_t)j * n + i];
      t = wave_sum(t);
      if (lane == 0) {
        s_ncnt[j] = (double)s_count[j];
        s_nW[j] = s_count[j] ? t / ((double)s_count[j] * (double)s_count[j]) : 0.0;
      }
    }
    __syncthreads();
  };
  auto estep = [&](const double* sarr, int* lab) {
    for (int i = lane; i < n; i += kWave) {
      int bj = 0;
      double bd = dpart(sarr, i, 0);
      for (int j = 1; j < k; ++j) {
        const double d = dpart(sarr, i, j);
        if (d < bd) {
          bd = d;
          bj = j;
        }
      }
      lab[i] = bj;
    }
    __syncthreads();
  };

  double best_inertia = 0.0;
  bool have_best = false;
  double* s_cur = w.s;
  double* s_new = w.sn;
  for (int init = 0; init < n_init; ++init) {
    // ---- k-means++ ----
    const int c0 = first[init];
    if (lane == 0) s_seed[0] = c0;
    double pot = 0.0;
    for (int i = lane; i < n; i += kWave) {
      const double d = dist_to(i, c0);
      w.closest[i] = d;
      pot += d;
    }
    pot = wave_sum(pot);
    __syncthreads();
    for (int cc = 1; cc < k; ++cc) {
      const double* rv = rnd + ((size_t)init * (k - 1) + (cc - 1)) * trials;
      int cand[kMaxTrials];
      for (int t = 0; t < trials; ++t) cand[t] = n;
      double cs = 0.0;
      for (int i = 0; i < n; ++i) {  // lane-uniform: searchsorted(cumsum(closest), rand * pot), side='left'
        cs += w.closest[i];
        for (int t = 0; t < trials; ++t)
          if (cand[t] == n && cs >= rv[t] * pot) cand[t] = i;
      }
      double pots[kMaxTrials];
      for (int t = 0; t < trials; ++t) {
        if (cand[t] >= n) cand[t] = n - 1;
        double sum = 0.0;
        for (int i = lane; i < n; i += kWave) {
          const double d = dist_to(i, cand[t]);
          const double mn = d < w.closest[i] ? d : w.closest[i];
          w.cand[(size_t)t * n + i] = mn;
          sum += mn;
        }
        pots[t] = wave_sum(sum);
      }
      int b = 0;
      for (int t = 1; t < trials; ++t)
        if (pots[t] < pots[b]) b = t;  // np.argmin: first minimum
      pot = pots[b];
      __syncthreads();
      for (int i = lane; i < n; i += kWave) w.closest[i] = w.cand[(size_t)b * n + i];
      if (lane == 0) s_seed[cc] = cand[b];
      __syncthreads();
    }
    // ---- Lloyd ----
    for (int j = lane; j < k; j += kWave) {
      s_cnt[j] = 1.0;
      s_W[j] = Gd(s_seed[j]);
    }
    for (int i = lane; i < n; i += kWave) {
      for (int j = 0; j < k; ++j) s_cur[(size_t)j * n + i] = w.G[(size_t)i * n + s_seed[j]];
      w.label_old[i] = -1;
    }
    __syncthreads();
    bool strict = false;
    for (int it = 0; it < 300; ++it) {
      estep(s_cur, w.label);
      for (int j = lane; j < k; j += kWave) s_count[j] = 0;
      __syncthreads();
      for (int i = lane; i < n; i += kWave) {
        w.member[i] = w.label[i];
        atomicAdd(&s_count[w.label[i]], 1);
      }
      __syncthreads();
      bool any_empty = false;
      for (int j = 0; j < k; ++j) any_empty |= (s_count[j] == 0);
      if (any_empty) {  // relocation: lane-uniform and sequential (rare)
        double far = 0.0;
        for (int i = lane; i < n; i += kWave) {
          const double d = Gd(i) + dpart(s_cur, i, w.label[i]);
          w.cand[i] = d;
          far = d > far ? d : far;
        }
        far = wave_max(far);
        __syncthreads();
        // _relocate_empty_clusters_dense returns early when max(distances) == 0 (every point on its centre: fewer distinct
        // points than clusters); the empty clusters then stay empty and _average_centers parks them on the biggest one
        for (int j = 0; j < k && far > 0.0; ++j) {
          if (s_count[j] != 0) continue;
          double bestd = -__builtin_huge_val();
          int besti = 0;
          for (int i = 0; i < n; ++i) {
            const double d = w.cand[i];
            if (d > bestd) {
              bestd = d;
              besti = i;
            }
          }
          __syncthreads();
          if (lane == 0) {
            w.member[besti] = j;
            w.cand[besti] = -__builtin_huge_val();  // taken
          }
          __syncthreads();
        }
      }
      subset_stats(w.member, s_new);
      // centre shift^2 = sum_j W_old + W_new - 2 <c_old, c_new>
      for (int j = 0; j < k; ++j) {
        double x = 0.0;
        for (int i = lane; i < n; i += kWave)
          if (w.member[i] == j) x += s_cur[(size_t)j * n + i];
        x = wave_sum(x);
        if (lane == 0) s_x[j] = x;
      }
      __syncthreads();
      double shift = 0.0;
      for (int j = 0; j < k; ++j)  // a cluster empty on either side sits on another cluster's centre: no shift of its own
        if (s_cnt[j] > 0 && s_ncnt[j] > 0) shift += s_W[j] + s_nW[j] - 2.0 * s_x[j] / (s_cnt[j] * s_ncnt[j]);
      __syncthreads();
      {
        double* t = s_cur;
        s_cur = s_new;
        s_new = t;
      }
      for (int j = lane; j < k; j += kWave) {
        s_cnt[j] = s_ncnt[j];
        s_W[j] = s_nW[j];
      }
      int same = 1;
      for (int i = lane; i < n; i += kWave) same &= (w.label[i] == w.label_old[i]);
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
    if (!strict) estep(s_cur, w.label);
    double inertia = 0.0;
    for (int i = lane; i < n; i += kWave) inertia += Gd(i) + dpart(s_cur, i, w.label[i]);
    inertia = wave_sum(inertia);
    bool take = !have_best;
    if (have_best && inertia < best_inertia) {  // _is_same_clustering
      for (int j = lane; j < k; j += kWave) s_map[j] = -1;
      __syncthreads();
      bool same_clu = true;
      if (lane == 0) {
        for (int i = 0; i < n && same_clu; ++i) {
          const int a = w.label[i], b = w.best_label[i];
          if (s_map[a] == -1) s_map[a] = b;
          else if (s_map[a] != b) same_clu = false;
        }
        s_count[0] = same_clu ? 1 : 0;
      }
      __syncthreads();
      take = s_count[0] == 0;
      __syncthreads();
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

  // ---- scores.py:168-185 on the chosen clustering: 1 - clarity(centres), clarity over the k centre vectors ----
  // original-space centres c_j = centred centre + mean:  c_j.c_l = g_jl + (R_j - mu) + (R_l - mu) + mu, with
  // g_jl = sum_{a in S_j} s_l[a] / (n_j n_l) (centred) and R_j = mean_{a in S_j} r_a, r_a = mean_b H_ab
  subset_stats(w.best_member, s_cur);
  for (int j = 0; j < k; ++j) {
    double rj = 0.0;
    for (int i = lane; i < n; i += kWave) {
      if (w.best_member[i] == j) {
        double t = 0.0;
        for (int b = 0; b < n; ++b) t += H[(size_t)i * n + b];
        rj += t / n;
      }
    }
    rj = wave_sum(rj);
    if (lane == 0) s_x[j] = s_ncnt[j] > 0 ? rj / s_ncnt[j] : 0.0;
  }
  __syncthreads();
  for (int j = 0; j < k; ++j)
    for (int l = 0; l < k; ++l) {
      double t = 0.0;
      for (int i = lane; i < n; i += kWave)
        if (w.best_member[i] == j) t += s_cur[(size_t)l * n + i];
      t = wave_sum(t);
      if (lane == 0) {
        const double g = (s_ncnt[j] > 0 && s_ncnt[l] > 0) ? t / (s_ncnt[j] * s_ncnt[l]) : 0.0;
        s_dot[j * k + l] = (s_ncnt[j] > 0 && s_ncnt[l] > 0) ? g + (s_x[j] - mu) + (s_x[l] - mu) + mu : 0.0;
      }
    }
  __syncthreads();
  if (lane == 0) {  // _average_centers: a cluster without members takes the centre of the biggest one (np.argmax: the first)
    int big = 0;
    for (int j = 1; j < k; ++j)
      if (s_ncnt[j] > s_ncnt[big]) big = j;
    for (int j = 0; j < k; ++j) {
      if (s_ncnt[j] > 0) continue;
      for (int l = 0; l < k; ++l) s_dot[j * k + l] = s_dot[big * k + l];
      for (int l = 0; l < k; ++l) s_dot[l * k + j] = s_dot[l * k + big];
    }
  }
  for (int j = lane; j < k; j += kWave) s_count[j] = 0;
  __syncthreads();
  for (int i = lane; i < n; i += kWave) atomicAdd(&s_count[w.best_label[i]], 1);
  __syncthreads();
  if (lane == 0) {
    // clarity_score of the centres:  ((mean_j c_j^)^2.sum() - 1/k) / (k - 1) * k,  c^ = c / max(||c||, 1e-12)
    double inv[kMaxClusters];
    for (int j = 0; j < k; ++j) {
      const double n2 = s_dot[j * k + j] > 0 ? s_dot[j * k + j] : 0.0;
      const double nn = sqrt(n2);
      inv[j] = 1.0 / (nn > 1e-12 ? nn : 1e-12);
    }
    double msq = 0.0;
    for (int j = 0; j < k; ++j)
      for (int l = 0; l < k; ++l) msq += s_dot[j * k + l] * inv[j] * inv[l];
    msq /= (double)k * (double)k;
    double poly = 1.0 - (msq - 1.0 / k) / (double)(k - 1) * (double)k;
    int mc = n, distinct = 0;
    for (int j = 0; j < k; ++j) {
      if (s_count[j] > 0) ++distinct;
      mc = s_count[j] < mc ? s_count[j] : mc;
    }
    if (distinct < k) mc = 0;  // the reference zero-fills the counts when a label is missing (scores.py:177)
    if (replace_empty && mc < 2) {
      const int ns = n < 10 ? n : 10;
      const double nm = sqrt(mu > 0 ? mu : 0.0);
      const double im = 1.0 / (nm > 1e-12 ? nm : 1e-12);
      double acc = 0.0;
      for (int i = 0; i < ns; ++i) {
        const double hii = H[(size_t)i * n + i];
        double ri = 0.0;
        for (int b = 0; b < n; ++b) ri += H[(size_t)i * n + b];
        ri /= n;
        const double ni = sqrt(hii > 0 ? hii : 0.0);
        const double ii = 1.0 / (ni > 1e-12 ? ni : 1e-12);
        const double m2 = (mu * im * im + hii * ii * ii + 2.0 * ri * im * ii) / 4.0;
        acc += (m2 - 0.5) * 2.0;
      }
      poly = 1.0 - acc / ns;
    }
    out[c] = poly;
    if (min_count) min_count[c] = mc;
  }
}

// Gram matrices for any n (the staged kernel above keeps a register array sized for n <= 128): one thread per pair,
// rows read from L2.  H[c] = V_c V_c^T in fp64.
__global__ __launch_bounds__(256) void gram_general_kernel(const float* __restrict__ V, int64_t C, int n, int64_t D,
                                                            double* __restrict__ H) {
  const int64_t c = blockIdx.y;
  const int64_t npairs = (int64_t)n * n;
  const float* Vc = V + c * (int64_t)n * D;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npairs; p += (int64_t)gridDim.x * 256) {
    const int i = (int)(p / n), j = (int)(p % n);
    if (j < i) continue;
    const float* a = Vc + (int64_t)i * D;
    const float* b = Vc + (int64_t)j * D;
    double s = 0.0;
    for (int64_t d = 0; d < D; ++d) s += (double)a[d] * (double)b[d];
    H[(c * n + i) * n + j] = s;
    H[(c * n + j) * n + i] = s;
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

// ---- any n_clusters / larger n: the general kernel ---------------------------------------------------------------------
SL_API int sl_kmeans_trials(int n_clusters) { return n_clusters >= 1 ? 2 + (int)log((double)n_clusters) : 0; }

SL_API size_t sl_polykmeans_ws_bytes(int64_t C, int64_t n, int64_t D, int n_clusters, int n_init) {
  (void)D;
  if (C < 0 || n < 0 || n_clusters < 1 || n_init < 1) return 0;
  const size_t draws = (size_t)n_init * 4 + (size_t)n_init * (size_t)(n_clusters > 1 ? n_clusters - 1 : 1) * kMaxTrials * 8;
  return (size_t)C * (size_t)n * (size_t)n * 8 + (size_t)C * gen_ws_bytes_per_component(n, n_clusters) + ((draws + 255) & ~(size_t)255) + 512;
}

SL_API int sl_polykmeans(const float* d_V, int64_t C, int64_t n, int64_t D, int n_clusters, const int32_t* h_first_center,
                         int n_init, const double* h_rand, int replace_empty_clusters, double* d_out, int32_t* d_min_count,
                         void* d_ws, size_t ws_bytes, void* stream) {
  SL_REQUIRE(C >= 0 && n >= 0 && D >= 0, "sl_polykmeans: negative shape");
  if (C == 0) return 0;
  SL_REQUIRE(n_clusters >= 2 && n_clusters <= kMaxClusters, "sl_polykmeans: n_clusters=%d not in [2, %d]", n_clusters, kMaxClusters);
  SL_REQUIRE(n >= n_clusters, "sl_polykmeans: n_samples=%lld should be >= n_clusters=%d.", (long long)n, n_clusters);  // sklearn's ValueError
  SL_REQUIRE(n <= kMaxNGeneral, "sl_polykmeans: n_samples=%lld exceeds the supported maximum %d", (long long)n, kMaxNGeneral);
  SL_REQUIRE(n_init >= 1 && n_init <= 32, "sl_polykmeans: n_init=%d not in [1, 32]", n_init);
  SL_REQUIRE(C <= 65535, "sl_polykmeans: %lld components in one call (the Gram kernel's grid holds 65535: split the call)", (long long)C);
  SL_REQUIRE(d_V && d_out && h_first_center && h_rand, "sl_polykmeans: null pointer");
  SL_REQUIRE(d_ws && ws_bytes >= sl_polykmeans_ws_bytes(C, n, D, n_clusters, n_init), "sl_polykmeans: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int trials = sl_kmeans_trials(n_clusters);
  for (int i = 0; i < n_init; ++i)
    SL_REQUIRE(h_first_center[i] >= 0 && h_first_center[i] < n, "sl_polykmeans: first centre %d out of range", h_first_center[i]);
  unsigned char* p = reinterpret_cast<unsigned char*>(((uintptr_t)d_ws + 255) & ~(uintptr_t)255);
  double* H = reinterpret_cast<double*>(p);
  p += (size_t)C * n * n * 8;
  unsigned char* slices = p;
  const size_t stride = gen_ws_bytes_per_component(n, n_clusters);
  p += (size_t)C * stride;
  double* d_rand = reinterpret_cast<double*>(p);
  const size_t rand_bytes = (size_t)n_init * (n_clusters - 1) * trials * 8;
  int32_t* d_first = reinterpret_cast<int32_t*>(p + ((rand_bytes + 255) & ~(size_t)255));
  // the data-independent draws of numpy's RandomState (host) -> device; pageable source: the copy is staged by the runtime
  SL_CHECK_HIP(hipMemcpyAsync(d_rand, h_rand, rand_bytes, hipMemcpyHostToDevice, st));
  SL_CHECK_HIP(hipMemcpyAsync(d_first, h_first_center, (size_t)n_init * 4, hipMemcpyHostToDevice, st));
  {
    ProfScope prof(SL_PROF_SCORES, st, (double)C * n * D * 4);
    int64_t bx = ((int64_t)n * n + 255) / 256;
    if (bx > 64) bx = 64;
    SL_LAUNCH(prof, gram_general_kernel, dim3((unsigned)bx, (unsigned)C), dim3(256), 0, st, d_V, C, (int)n, D, H);
  }
  hipLaunchKernelGGL(kmeansk_kernel, dim3((unsigned)C), dim3(64), 0, st, (const double*)H, C, (int)n, D, n_clusters, trials, n_init,
                     (const int32_t*)d_first, (const double*)d_rand, replace_empty_clusters, slices, stride, d_out, d_min_count);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}
