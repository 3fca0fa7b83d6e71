// K11 — primitives of a native CLIP-family ViT tower.
//
// The reference's OpenClip.encode_image / encode_text (foundation_models/clip.py:103-135) only forward
// to the third-party open_clip model; its arithmetic is a standard pre-LN transformer:
//   tokens -> ln_pre -> L x [ x += out_proj(MHA(ln_1(x))) ; x += c_proj(gelu(c_fc(ln_2(x)))) ] -> ln_post -> proj
// These kernels run that tower in fp32 on the device: every linear layer is the fp32-input MFMA GEMM of
// gemm_f32.hpp with bias / GELU / residual (and, for the patch embedding, the token scatter + positional
// add) fused into its epilogue; LayerNorm, attention and patch extraction are small bandwidth-bound kernels.
// The orchestration (weights, layer loop) lives in semanticlens_amd/foundation_models/native_clip.py.
#include <cstdlib>
#include <cstring>

#include "gemm_bf16x3.hpp"
#include "gemm_f32.hpp"

namespace sl {
namespace {

// ---- linear: out = act(x W^T + b) (+ residual), optional row scatter for the patch embedding ----------
using gemm3::split_kp;
using gemm3::store_split;   // (v, row, col, Kp, split matrix): see gemm_bf16x3.hpp for the layout
using gemm3::store_split4;

// erf for the GELU epilogue: Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7) with the hardware exp2 and reciprocal —
// 15 instructions instead of libm erff's ~50.  The epilogue of a 256 x 256 tile evaluates it 128 times per lane, which
// with erff cost 25 K cycles per tile next to a k loop of 52-78 K (K = 512 / 768): fc1 ran 29 % slower per tile than the
// plain GEMM.  max |gelu - exact| over [-8, 8] stays 4.7e-7, the same as with a correctly rounded fp32 erf (the final
// products round at that level).
// Every multiply-add below is written as an explicit fma and contraction is off: left to the compiler, which mul + add pairs
// fuse depends on how the surrounding (unrolled) epilogue code was vectorised, so the SAME value came out 1 ulp apart
// depending on the row's position inside the 256-row tile — and an embedding depended on where its sample sat in the batch
// (found when a batch was cut in two: rows 4800.. of a 9600-row GEMM vs rows 0.. of a 4800-row one).
__device__ inline float erf_as(float z) {
#pragma clang fp contract(off)
  const float a = __builtin_fabsf(z);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, a, 1.f));
  float p = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
  p = __builtin_fmaf(t, p, 1.421413741f);
  p = __builtin_fmaf(t, p, -0.284496736f);
  p = __builtin_fmaf(t, p, 0.254829592f);
  const float poly = t * p;
  return __builtin_copysignf(__builtin_fmaf(-poly, __expf(-(a * a)), 1.f), z);
}

template <int ACT, bool RES, bool REMAP, bool SPLIT = false>
struct LinearEpi {
  const float* bias;  // (N) or nullptr
  const float* res;   // (M, ldo) or nullptr; may alias out (same element read then written by one lane)
  float* out;
  uint16_t* out_sp;   // SPLIT: the result goes out as a split matrix (M x N, gemm_bf16x3.hpp) instead of fp32
  int64_t out_kp;     // its padded row length split_kp(N)
  int64_t ldo;
  int64_t rpg, gstride, roff;  // REMAP: out row = (r / rpg) * gstride + roff + r % rpg
  const float* rowadd;         // REMAP: (roff + r % rpg, col) of this (T, N) table is added (positional embedding)
  int64_t N;
  uint32_t rpg_inv;            // REMAP: floor(2^32 / rpg) (2^32 - 1 for rpg = 1), for row / rpg without a division
  // row -> (row / rpg, row % rpg), rows and rpg below 2^32: the high product underestimates the quotient by at most one
  __device__ inline void divmod_rpg(int64_t row, uint32_t& q, uint32_t& rem) const {
    const uint32_t r = (uint32_t)row, g = (uint32_t)rpg;
    q = __umulhi(r, rpg_inv);
    rem = r - q * g;
    if (rem >= g) { q += 1; rem -= g; }
    if (rem >= g) { q += 1; rem -= g; }
  }
  __device__ inline float column(int64_t col) const { return bias ? bias[col] : 0.f; }
  // the same epilogue for a GEMM over columns [c0, ...) of the output (c0 a multiple of 32; gemm_bf16x3.hpp: column-strip split)
  LinearEpi shifted(int64_t c0) const {
    LinearEpi e = *this;
    if (e.bias) e.bias += c0;
    if (e.res) e.res += c0;
    if (e.out) e.out += c0;
    if (e.out_sp) e.out_sp += (c0 >> 5) * 64;  // split_pos: 64 elements per 32-column tile of a row
    if (e.rowadd) e.rowadd += c0;
    return e;
  }
  // what `store` adds from memory, fetched ahead of the stores by kernels that batch their epilogue (gemm_8phase.hpp);
  // `store_fetched(..., fetch(row, col))` == `store(...)` bit for bit (same operands, same order of additions)
  static constexpr bool kFetches = RES || REMAP;
  __device__ inline int64_t out_row(int64_t row) const {
    if constexpr (REMAP) {
      // (two 64-bit divisions per element cost the patch-embedding GEMM 10 us of 176)
      uint32_t q, rem;
      divmod_rpg(row, q, rem);
      return (int64_t)q * gstride + roff + (int64_t)rem;
    }
    return row;
  }
  __device__ inline float2 fetch(int64_t row, int64_t col) const {  // (positional-table value, residual value)
    float2 f = make_float2(0.f, 0.f);
    if constexpr (REMAP) {
      if (rowadd) {
        uint32_t q, rem;
        divmod_rpg(row, q, rem);
        f.x = rowadd[(roff + (int64_t)rem) * N + col];
      }
    }
    if constexpr (RES) f.y = res[out_row(row) * ldo + col];
    return f;
  }
  __device__ inline void store_fetched(int64_t row, int64_t col, float acc, float b, float2 f) const {
#pragma clang fp contract(off)  // see erf_as: one rounding sequence per element, whatever code surrounds it
    float v = acc + b;
    if constexpr (ACT == SL_ACT_GELU) {
      const float hv = 0.5f * v;
      v = __builtin_fmaf(hv, erf_as(v * 0.70710678118654752440f), hv);  // 0.5 v (1 + erf(v / sqrt 2))
    }
    if constexpr (ACT == SL_ACT_QUICKGELU) v = v * __builtin_amdgcn_rcpf(1.f + __expf(-1.702f * v));
    if constexpr (ACT == SL_ACT_GELU_TANH) {  // torch gelu(approximate="tanh"): SigLIP's "gelu_pytorch_tanh"
      const float u = 0.79788456080286535588f * __builtin_fmaf(0.044715f * v, v * v, v);
      // 0.5 v (1 + tanh u) = v - v / (1 + exp(2 u)); exp overflow -> v, underflow -> 0
      v = __builtin_fmaf(-v, __builtin_amdgcn_rcpf(1.f + __expf(2.f * u)), v);
    }
    const int64_t orow = out_row(row);
    if constexpr (REMAP) {
      if (rowadd) v += f.x;
    }
    if constexpr (RES) v += f.y;
    if constexpr (SPLIT) store_split(v, orow, col, out_kp, out_sp);
    else out[orow * ldo + col] = v;
  }
  __device__ inline void store(int64_t row, int64_t col, float acc, float b) const { store_fetched(row, col, acc, b, fetch(row, col)); }
};

template <int ACT, bool RES, bool REMAP>
int run_linear(ProfScope& prof, const float* x, int64_t M, int64_t K, const float* w, int64_t N, const float* bias,
               const float* res, float* out, int64_t ldo, int64_t rpg, int64_t gstride, int64_t roff,
               const float* rowadd, hipStream_t st) {
  LinearEpi<ACT, RES, REMAP> epi{bias, res, out, nullptr, 0, ldo, rpg, gstride, roff, rowadd, N,
                                 rpg > 1 ? (uint32_t)((1ull << 32) / (uint64_t)rpg) : 0xFFFFFFFFu};
  return gemm::launch_gemm_nt(prof, x, M, w, N, K, epi, st);
}

template <int ACT, bool RES, bool REMAP, bool SPLIT>
int run_linear3(ProfScope& prof, const uint16_t* xs, int64_t M, int64_t K, const uint16_t* ws, int64_t N, const float* bias,
                const float* res, float* out, uint16_t* osp, int64_t ldo, int64_t rpg, int64_t gstride, int64_t roff,
                const float* rowadd, hipStream_t st) {
  LinearEpi<ACT, RES, REMAP, SPLIT> epi{bias, res, out, osp, split_kp(N), ldo, rpg, gstride, roff, rowadd, N,
                                        rpg > 1 ? (uint32_t)((1ull << 32) / (uint64_t)rpg) : 0xFFFFFFFFu};
  return gemm3::launch_gemm3_nt(prof, xs, M, ws, N, K, epi, st);
}

// ---- LayerNorm over the last dim: one wave per row -------------------------------------------------------------
// Fast path (J > 0: cols % 4 == 0, cols <= 256 J, 16-byte aligned rows): the row is read once into registers (up to J
// float4 per lane: J = 4 up to 1024 columns, J = 8 up to 2048 — SigLIP-so400m's 1152 took the three-pass path until round 4:
// 227 us per LayerNorm at 65 536 rows, 2.7 TB/s), mean and variance are two wave reductions, the result leaves as 16-byte
// fp32 stores or as 8-byte packed bf16 hi / lo stores.  J = 0: any shape, three passes over the (L1-resident) row.
template <int J>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t rows, int cols,
                                                         int64_t xs, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps,
                                                         float* __restrict__ out, int64_t os, uint16_t* __restrict__ osp) {
  const int64_t okp = split_kp(cols);  // split output: a (rows x cols) split matrix
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  if constexpr (J > 0) {
    const int nq = cols >> 2;  // float4 chunks per row, <= 64 J
    // gamma / beta: in registers for J = 4 (32 of 83); for J = 8 they would be 64 of 147 registers (three waves per SIMD, 55 KB
    // of loads in flight per CU: 4.2 TB/s at so400m's 1152 columns) — there the workgroup keeps them in LDS instead
    constexpr bool kLds = J > 4;
    __shared__ float4 s_g[kLds ? 64 * J : 1], s_b[kLds ? 64 * J : 1];
    float4 g[kLds ? 1 : J], bt[kLds ? 1 : J];
    if constexpr (kLds) {
      for (int q = threadIdx.x; q < 64 * J; q += 256) {
        s_g[q] = q < nq ? reinterpret_cast<const float4*>(gamma)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        s_b[q] = q < nq ? reinterpret_cast<const float4*>(beta)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      __syncthreads();
    } else {
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int q = lane + 64 * j;
        g[j] = q < nq ? reinterpret_cast<const float4*>(gamma)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        bt[j] = q < nq ? reinterpret_cast<const float4*>(beta)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    for (int64_t r = wave; r < rows; r += nw) {
      const float4* p = reinterpret_cast<const float4*>(x + r * xs);
      float4 v[J];
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int q = lane + 64 * j;
        v[j] = q < nq ? p[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
      }
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      const float mean = s / (float)cols;
      float var = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        if (lane + 64 * j < nq) {
          const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
          var += (a * a + b * b) + (c * c + d * d);
        }
      }
      for (int off = 32; off > 0; off >>= 1) var += __shfl_xor(var, off, 64);
      const float rstd = 1.f / sqrtf(var / (float)cols + eps);
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int q = lane + 64 * j;
        if (q < nq) {
          const float4 gj = kLds ? s_g[q] : g[kLds ? 0 : j], bj = kLds ? s_b[q] : bt[kLds ? 0 : j];
          const float4 y = make_float4((v[j].x - mean) * rstd * gj.x + bj.x, (v[j].y - mean) * rstd * gj.y + bj.y,
                                       (v[j].z - mean) * rstd * gj.z + bj.z, (v[j].w - mean) * rstd * gj.w + bj.w);
          if (out) *reinterpret_cast<float4*>(out + r * os + q * 4) = y;
          if (osp) store_split4(y, r, q * 4, okp, osp);
        }
      }
    }
    return;
  }
  for (int64_t r = wave; r < rows; r += nw) {
    const float* p = x + r * xs;
    float s = 0.f;
    for (int i = lane; i < cols; i += 64) s += p[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)cols;
    float v = 0.f;
    for (int i = lane; i < cols; i += 64) {
      const float d = p[i] - mean;
      v += d * d;
    }
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const float rstd = 1.f / sqrtf(v / (float)cols + eps);
    for (int i = lane; i < cols; i += 64) {
      const float y = (p[i] - mean) * rstd * gamma[i] + beta[i];
      if (out) out[r * os + i] = y;
      if (osp) store_split(y, r, i, okp, osp);
    }
  }
}

// (The first attention kernel of the tower — VALU, four lanes per query row, head_dim 64 — was superseded by the two matrix-core
// kernels below in round 2 and left the product library in round 6: tools/native/attention_valu_lab.hpp.)
// ---- the same attention on the fp32-input matrix cores ---------------------------------------------------------
// The VALU kernel re-read K and V from LDS for every group of 16 query rows (1.6 MB of LDS traffic per head at
// T = 50) and was LDS-bandwidth bound.  Here a wave owns 32 query rows and works on 32-key tiles with
// v_mfma_f32_32x32x2_f32 (exact fp32 products and accumulation):
//   S^T (keys x queries) = K Q^T : A = K rows from LDS (lane l: key l % 32, dims 32 (l / 32) + s), B = Q^T from
//                                  registers (lane l: query l % 32, the same 32 dims, pre-scaled by 1/8)
//   O^T (dims x queries) += V^T P^T : computing the TRANSPOSED scores puts P^T in the accumulator exactly where the B
//                                  operand of this product wants it: k-step s takes keys (s&3) + 8 (s>>2) + 4 (l/32),
//                                  the rows accumulator register s holds, so P never moves between lanes or through LDS
// Online softmax per query = per lane (its two half-waves combine max and sum with one cross-half exchange each).
// K / V rows are padded to 68 floats: the 16-byte K fragment reads are conflict-free, V is read one float per lane.
typedef float floatx16_t __attribute__((ext_vector_type(16)));

__device__ inline float xhalf(float v) {  // value held by the lane 32 away
  return __shfl_xor(v, 32, 64);
}

// head_dim D (multiple of 8, <= 128): each half-wave owns D/2 of the dims in the score product; the output has
// NT = ceil(D / 32) tiles of 32 dims (V columns past D are zero).  K / V rows hold 32 NT + 4 floats.
__host__ __device__ constexpr int attn_ld(int D) { return ((D + 31) / 32) * 32 + 4; }
__host__ __device__ constexpr int attn_chunk(int D) { return D <= 64 ? 256 : (D <= 96 ? 192 : 128); }  // keys in LDS at a time

template <int D>
__global__ __launch_bounds__(256) void attention_mfma_kernel(const float* __restrict__ qkv, int T, int H, int causal, float scale,
                                                              float* __restrict__ out, uint16_t* __restrict__ osp) {
  constexpr int kDh = D;
  constexpr int NT = (D + 31) / 32;
  constexpr int HD = D / 2;  // dims per half-wave in the score product (multiple of 4)
  constexpr int kKvLd = attn_ld(D);
  constexpr int kAttnChunk = attn_chunk(D);
  extern __shared__ __align__(16) float smem[];
  const int Tp = (T + 31) & ~31;
  const int KC = Tp < kAttnChunk ? Tp : kAttnChunk;  // keys per LDS chunk (multiple of 32)
  float* sK = smem;                       // KC x 68
  float* sV = smem + (size_t)KC * kKvLd;  // KC x 68
  const int tid = threadIdx.x;
  const int nwaves = blockDim.x >> 6;
  const int w = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t b = blockIdx.x / H;
  const int h = blockIdx.x % H;
  const int64_t ld = 3ll * H * kDh;
  const float* base = qkv + b * T * ld + h * kDh;
  const int nqt = Tp / 32, kct = KC / 32;
  int loaded = -1;  // first key tile of the chunk in LDS (workgroup-uniform)
  // rounds of `nwaves` query tiles; within a round the waves walk the key chunks together (longer sequences than one
  // chunk re-stream K / V from L2 once per round)
  for (int qt0 = 0; qt0 < nqt; qt0 += nwaves) {
    const int qt = qt0 + w;
    const bool active = qt < nqt;
    const int q = qt * 32 + li;  // this lane's query row (both half-waves)
    float qf[HD];
    {
      const float* qp = base + (int64_t)(q < T ? q : T - 1) * ld + lh * HD;
#pragma unroll
      for (int c = 0; c < HD / 4; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(qp + c * 4);
        qf[4 * c] = v.x * scale; qf[4 * c + 1] = v.y * scale; qf[4 * c + 2] = v.z * scale; qf[4 * c + 3] = v.w * scale;
      }
    }
    floatx16_t o[NT];  // O^T: tile t = dims 32 t .. 32 t + 31 (rows) x this lane's query
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[t][e] = 0.f;
    float m = -__builtin_huge_valf(), l = 0.f;
    const int round_last = qt0 + nwaves < nqt ? qt0 + nwaves : nqt;  // key tiles any wave of this round may need
    const int nkt_round = causal ? round_last : nqt;
    const int nkt = !active ? 0 : (causal ? qt + 1 : nqt);
    for (int kc0 = 0; kc0 < nkt_round; kc0 += kct) {
      if (kc0 != loaded) {
        if (loaded >= 0) __syncthreads();  // every wave is done with the previous chunk
        for (int e = tid; e < KC * (NT * 8); e += blockDim.x) {  // NT * 8 float4 per row, zero past D and past T
          const int tl = e / (NT * 8), c = e % (NT * 8);
          const int t = kc0 * 32 + tl;
          float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
          if (t < T && c * 4 < D) {
            kv = *reinterpret_cast<const float4*>(base + t * ld + (int64_t)H * kDh + c * 4);
            vv = *reinterpret_cast<const float4*>(base + t * ld + 2ll * H * kDh + c * 4);
          }
          *reinterpret_cast<float4*>(sK + (size_t)tl * kKvLd + c * 4) = kv;
          *reinterpret_cast<float4*>(sV + (size_t)tl * kKvLd + c * 4) = vv;
        }
        loaded = kc0;
        __syncthreads();
      }
      const int kt_end = kc0 + kct < nkt ? kc0 + kct : nkt;
      for (int kt = kc0; kt < kt_end; ++kt) {
        const int kl = (kt - kc0) * 32;  // first row of this key tile inside the chunk
        floatx16_t st;
#pragma unroll
        for (int e = 0; e < 16; ++e) st[e] = 0.f;
        const float* kp = sK + (size_t)(kl + li) * kKvLd + lh * HD;
#pragma unroll
        for (int c = 0; c < HD / 4; ++c) {
          const float4 kv = *reinterpret_cast<const float4*>(kp + c * 4);
          st = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.x, qf[4 * c], st, 0, 0, 0);
          st = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.y, qf[4 * c + 1], st, 0, 0, 0);
          st = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.z, qf[4 * c + 2], st, 0, 0, 0);
          st = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.w, qf[4 * c + 3], st, 0, 0, 0);
        }
        // st[r] = score of key kt*32 + (r&3) + 8 (r>>2) + 4 lh against query q
        float mx = -__builtin_huge_valf();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const bool masked = key >= T || (causal && key > q);
          st[r] = masked ? -__builtin_huge_valf() : st[r];
          mx = fmaxf(mx, st[r]);
        }
        mx = fmaxf(mx, xhalf(mx));
        const float mn = fmaxf(m, mx);  // key 0 is never masked for a valid query, so mn is finite from the first tile on
        const float alpha = expf(m - mn);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          st[r] = expf(st[r] - mn);  // exp(-inf) = 0 for masked keys
          ps += st[r];
        }
        ps += xhalf(ps);
        l = l * alpha + ps;
        m = mn;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[t][e] *= alpha;
        // O^T += V^T P^T: k-step s covers keys (s&3) + 8 (s>>2) + 4 (lane/32), whose probabilities are st[s]
        const float* vp = sV + (size_t)(kl + 4 * lh) * kKvLd + li;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
          const float* vr = vp + (size_t)((s2 & 3) + 8 * (s2 >> 2)) * kKvLd;
#pragma unroll
          for (int t = 0; t < NT; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[32 * t], st[s2], o[t], 0, 0, 0);
        }
      }
    }
    if (active && q < T) {
      const float inv = 1.f / l;
      const int64_t obase = (b * T + q) * (int64_t)H * kDh + h * kDh;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // accumulator registers 4g..4g+3 of tile t are dims 32 t + 8 g + 4 lh + (0..3)
          const int d = 32 * t + 8 * g + 4 * lh;
          if (d < D) {
            const float4 v = make_float4(o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
            if (out) *reinterpret_cast<float4*>(out + obase + d) = v;
            if (osp) store_split4(v, b * T + q, h * kDh + d, split_kp((int64_t)H * kDh), osp);
          }
        }
    }
  }
}

// ---- the same attention with split-bf16 x3 products on the bf16 matrix cores (round 3) ---------------------------------
// attention_mfma_kernel issues fp32-input MFMAs: 16 K matrix-pipe cycles per (image, head) at T = 50 — a 24 µs floor per
// ViT-B/32 block, 69 µs measured, 0.85 ms per encode.  In the bf16x3 arithmetic mode of the towers (every GEMM around this
// kernel already computes a*b as a_lo*b_hi + a_hi*b_lo + a_hi*b_hi) the two products of attention take the same form on
// v_mfma_f32_32x32x16_bf16: 96 cycles per 16 k instead of 512.
//   S^T (keys x queries) = K Q^T : A = K rows from LDS (lane l: key l % 32, dims 16 s + 8 (l / 32) + [0, 8), hi and lo planes),
//                                  B = Q^T from registers (lane l: query l % 32, the same dims, pre-scaled, split once)
//   O^T (dims x queries) += V^T P^T : B = P^T straight from the score accumulator: k-step s2 of a 32-key tile takes, for the
//                                  lane half h, the keys (r & 3) + 8 (r >> 2) + 4 h of registers r = 8 s2 .. 8 s2 + 7 — the rows
//                                  those registers hold — so P never moves between lanes; A = V^T from LDS, stored
//                                  TRANSPOSED ([dim][key]) with the keys of a tile permuted into that order, 16 bytes per read.
// Online softmax, masking, chunking over the keys and the output layout are those of the fp32 kernel.  K rows and V^T rows
// carry 16 bytes of padding: a 16-lane group's ds_read_b128 then covers the 64 banks once.
typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));
typedef float afloatx16 __attribute__((ext_vector_type(16)));

__host__ __device__ constexpr int attn3_dp(int D) { return (D + 15) / 16 * 16; }   // dims per K row (k-steps of 16)
__host__ __device__ constexpr int attn3_kc() { return 128; }                       // keys in LDS at a time

// position of key `kk` (0..31 of its tile) in a V^T row: block (2 s2 + h) of eight, j-th of the block
__device__ inline int attn3_vpos(int kk) {
  const int r = (kk & 3) + 4 * (kk >> 3), h = (kk >> 2) & 1;
  return (2 * (r >> 3) + h) * 8 + (r & 7);
}

// e^x for the softmax of attention_bf16x3_kernel: 2^(x log2 e) on the hardware exp2 with the rounding error of the product and the
// low half of the constant folded back in (2^(t + d) = 2^t (1 + d ln 2)): ~2e-7 relative — the library expf's accuracy class — in
// 7 instructions instead of ~15.  Round 4: with two waves per SIMD the kernel's VALU work (17 exponentials and 48 rescales per
// lane and key tile) outweighed its 33 MFMAs.  x <= 0 here; -inf (masked keys, the first tile's running maximum) gives 0.
__device__ __forceinline__ float exp_neg(float x) {
  x = fmaxf(x, -104.f);  // 2^-150 flushes to 0; keeps -inf out of the error term (inf - inf)
  const float t = x * 1.44269502e+00f;
  float d = __builtin_fmaf(x, 1.44269502e+00f, -t);
  d = __builtin_fmaf(x, 1.92596299e-08f, d);
  const float r = __builtin_amdgcn_exp2f(t);
  return __builtin_fmaf(r, d * 6.93147182e-01f, r);
}

// MAXW waves per (image, head), one 32-query tile each per round.  MAXW = 8 (round 4) for sequences of more than four tiles:
// SigLIP-so400m's 256 tokens ran as two rounds of four waves, which staged (converted, transposed) K and V twice per
// workgroup and left each SIMD with ONE wave whose MFMAs and softmax VALU work serialise (253 us per layer at B = 64: 63 us per
// workgroup, one workgroup per CU by its 97 KB of LDS); eight waves stage once and pair two waves per SIMD.
#ifndef SL_ATTN_EXP
#define SL_ATTN_EXP 0  // lab only (garbage results): 1 = no key-tile loop (staging + output only), 2 = no K / V global loads (compute only)
#endif
// Where a workgroup's life goes (so400m shape, B = 256, `tools/attn_exp_probe.py`, profiles/r04_attention_ablation.txt): 592 us per
// layer as shipped, 251 with the key-tile loop compiled out (staging + output), 417 with the K / V global loads compiled out.
// Staging and MFMAs do not overlap: the 97 KB of LDS and 216 registers keep ONE workgroup per CU.  Tried and dropped: pulling the
// next chunk's (and the next workgroup's first chunk's) lines into L2 with LDS-DMA touches into a dummy area — 605 -> 648 us.
template <int D, int MAXW>
__global__ __launch_bounds__(64 * MAXW, (D <= 96 ? 2 : 1)) void attention_bf16x3_kernel(const float* __restrict__ qkv, int T, int H, int causal, float scale,
                                                                float* __restrict__ out, uint16_t* __restrict__ osp) {
  constexpr int kDh = D;
  constexpr int DP = attn3_dp(D), NS = DP / 16;  // k-steps of the score product
  constexpr int NT = (D + 31) / 32;              // output tiles of 32 dims
  constexpr int KROW = DP * 2 + 16;              // bytes per K row (one plane)
  extern __shared__ __align__(16) unsigned char smem3[];
  const int Tp = (T + 31) & ~31;
  const int KC = Tp < attn3_kc() ? Tp : attn3_kc();
  const int VROW = KC * 2 + 16;  // bytes per V^T row (one plane)
  unsigned char* sKh = smem3;
  unsigned char* sKl = sKh + (size_t)KC * KROW;
  unsigned char* sVh = sKl + (size_t)KC * KROW;
  unsigned char* sVl = sVh + (size_t)(NT * 32) * VROW;
  const int tid = threadIdx.x;
  const int nwaves = blockDim.x >> 6;
  const int w = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  const int64_t b = blockIdx.x / H;
  const int h = blockIdx.x % H;
  const int64_t ld = 3ll * H * kDh;
  const float* base = qkv + b * T * ld + h * kDh;
  const int nqt = Tp / 32, kct = KC / 32;
  int loaded = -1;
  auto split8 = [](const float* v, abf16x8& hi, abf16x8& lo) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const __bf16 hb = (__bf16)v[j];
      hi[j] = hb;
      lo[j] = (__bf16)(v[j] - (float)hb);
    }
  };
  for (int qt0 = 0; qt0 < nqt; qt0 += nwaves) {
    const int qt = qt0 + w;
    const bool active = qt < nqt;
    const int q = qt * 32 + li;
    // the query's dims are only REQUESTED here; they are scaled and split after the first K / V chunk has been staged, so
    // that the wave does not sit out one memory round trip for Q before it asks for K and V
    float4 qraw[NS][2];
    {
      const float* qp = base + (int64_t)(q < T ? q : T - 1) * ld;
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int d0 = 16 * s + 8 * lh + 4 * c;
          qraw[s][c] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (d0 < D) qraw[s][c] = *reinterpret_cast<const float4*>(qp + d0);  // D % 8 == 0: whole float4s
        }
    }
    abf16x8 qh[NS], ql[NS];
    bool q_ready = false;
    afloatx16 o[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[t][e] = 0.f;
    float m = -__builtin_huge_valf(), l = 0.f;
    const int round_last = qt0 + nwaves < nqt ? qt0 + nwaves : nqt;
    const int nkt_round = causal ? round_last : nqt;
    const int nkt = !active ? 0 : (causal ? qt + 1 : nqt);
    for (int kc0 = 0; kc0 < nkt_round; kc0 += kct) {
      if (kc0 != loaded) {
        if (loaded >= 0) __syncthreads();
        // K rows: [hi(DP) | pad] and [lo(DP) | pad]; V^T rows per dim: keys of the chunk in `attn3_vpos` order.  Zero past D / T.
        // A thread takes FOUR consecutive keys x four dims: eight 16-byte global loads in flight, and — four aligned keys
        // are four consecutive V^T positions — the 4 x 4 transpose leaves as 8-byte LDS stores (one key x four dims per
        // thread meant eight serial round trips per thread and 2-byte stores: 50 -> 3x µs per ViT-B/32 block).
        constexpr int C4 = NT * 8;  // float4 chunks per source row that any plane needs (dims < NT * 32 >= DP)
        for (int e = tid; e < (KC / 4) * C4; e += blockDim.x) {
          const int tl0 = (e / C4) * 4, c = e % C4;
          float4 kv[4], vv[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int t = kc0 * 32 + tl0 + i;
            kv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            vv[i] = kv[i];
            if (SL_ATTN_EXP != 2 && t < T && c * 4 < D) {
              kv[i] = *reinterpret_cast<const float4*>(base + t * ld + (int64_t)H * kDh + c * 4);
              vv[i] = *reinterpret_cast<const float4*>(base + t * ld + 2ll * H * kDh + c * 4);
            }
          }
          auto pack4 = [](float a, float b2, float c2, float d2, uint2& hi, uint2& lo) __attribute__((always_inline)) {
            const float f[4] = {a, b2, c2, d2};
            uint16_t hb[4], lb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const __bf16 hh = (__bf16)f[j];
              hb[j] = __builtin_bit_cast(uint16_t, hh);
              lb[j] = __builtin_bit_cast(uint16_t, (__bf16)(f[j] - (float)hh));
            }
            hi = make_uint2(hb[0] | ((uint32_t)hb[1] << 16), hb[2] | ((uint32_t)hb[3] << 16));
            lo = make_uint2(lb[0] | ((uint32_t)lb[1] << 16), lb[2] | ((uint32_t)lb[3] << 16));
          };
          if (c * 4 < DP) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              uint2 hi, lo;
              pack4(kv[i].x, kv[i].y, kv[i].z, kv[i].w, hi, lo);
              *reinterpret_cast<uint2*>(sKh + (size_t)(tl0 + i) * KROW + c * 8) = hi;
              *reinterpret_cast<uint2*>(sKl + (size_t)(tl0 + i) * KROW + c * 8) = lo;
            }
          }
          const int pos0 = (tl0 & ~31) + attn3_vpos(tl0 & 31);  // keys tl0 .. tl0 + 3 sit at pos0 .. pos0 + 3
          const float vt[4][4] = {{vv[0].x, vv[1].x, vv[2].x, vv[3].x}, {vv[0].y, vv[1].y, vv[2].y, vv[3].y},
                                  {vv[0].z, vv[1].z, vv[2].z, vv[3].z}, {vv[0].w, vv[1].w, vv[2].w, vv[3].w}};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            uint2 hi, lo;
            pack4(vt[j][0], vt[j][1], vt[j][2], vt[j][3], hi, lo);
            const size_t off = (size_t)(c * 4 + j) * VROW + pos0 * 2;
            *reinterpret_cast<uint2*>(sVh + off) = hi;
            *reinterpret_cast<uint2*>(sVl + off) = lo;
          }
        }
        loaded = kc0;
        __syncthreads();
      }
      if (!q_ready) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const float v[8] = {qraw[s][0].x * scale, qraw[s][0].y * scale, qraw[s][0].z * scale, qraw[s][0].w * scale,
                              qraw[s][1].x * scale, qraw[s][1].y * scale, qraw[s][1].z * scale, qraw[s][1].w * scale};
          split8(v, qh[s], ql[s]);
        }
        q_ready = true;
      }
      const int kt_end = SL_ATTN_EXP == 1 ? kc0 : (kc0 + kct < nkt ? kc0 + kct : nkt);
      for (int kt = kc0; kt < kt_end; ++kt) {
        const int kl = (kt - kc0) * 32;
        afloatx16 st;
#pragma unroll
        for (int e = 0; e < 16; ++e) st[e] = 0.f;
        const unsigned char* kph = sKh + (size_t)(kl + li) * KROW + lh * 16;
        const unsigned char* kpl = sKl + (size_t)(kl + li) * KROW + lh * 16;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const abf16x8 kh = *reinterpret_cast<const abf16x8*>(kph + s * 32);
          const abf16x8 klo = *reinterpret_cast<const abf16x8*>(kpl + s * 32);
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(klo, qh[s], st, 0, 0, 0);  // small terms first, as in the GEMMs
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[s], st, 0, 0, 0);
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[s], st, 0, 0, 0);
        }
        float mx = -__builtin_huge_valf();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const bool masked = key >= T || (causal && key > q);
          st[r] = masked ? -__builtin_huge_valf() : st[r];
          mx = fmaxf(mx, st[r]);
        }
        mx = fmaxf(mx, xhalf(mx));
        const float mn = fmaxf(m, mx);
        const float alpha = exp_neg(m - mn);
        float ps = 0.f;
        float pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          pv[r] = exp_neg(st[r] - mn);
          ps += pv[r];
        }
        ps += xhalf(ps);
        l = l * alpha + ps;
        // once the running maxima have settled (after the first key tiles of most rows) alpha is exactly 1 in every lane and
        // the 16 NT rescales of the output accumulators are skipped for the whole wave
        if (!__all(mn == m)) {
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[t][e] *= alpha;
        }
        m = mn;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          abf16x8 ph, pl;
          split8(pv + 8 * s2, ph, pl);
          const size_t col = (size_t)(kl + (2 * s2 + lh) * 8) * 2;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const abf16x8 vh = *reinterpret_cast<const abf16x8*>(sVh + (size_t)(32 * t + li) * VROW + col);
            const abf16x8 vl = *reinterpret_cast<const abf16x8*>(sVl + (size_t)(32 * t + li) * VROW + col);
            o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, o[t], 0, 0, 0);
            o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, o[t], 0, 0, 0);
            o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, o[t], 0, 0, 0);
          }
        }
      }
    }
    // Output.  In the accumulator a lane owns ONE query (row of the output) and 4-dim groups of it, so stored directly every
    // store instruction scatters 16-byte fragments over 32 different rows.  The tile goes through LDS instead (the K planes
    // are free once every wave has left the key loop): [query][dim] fp32 with padded rows, read back so that the lanes of a
    // 16-lane group (D = 64) cover one token's dims in order — whole 64-byte runs of the split layout / 256-byte fp32 rows.
    __syncthreads();
    {
      constexpr int SROW = D * 4 + 16;  // staging row: D floats + 16 bytes (bank spread)
      unsigned char* stage = smem3 + (size_t)w * 32 * SROW;
      const float inv = 1.f / l;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = 32 * t + 8 * g + 4 * lh;
          if (d < D)
            *reinterpret_cast<float4*>(stage + (size_t)li * SROW + d * 4) =
                make_float4(o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave reads only what it wrote itself: no barrier needed
      constexpr int CPR = D / 4;  // 16-byte chunks per row
      if (active) {
        for (int e = lane; e < 32 * CPR; e += 64) {
          const int row = e / CPR, c = e % CPR;
          const int qq = qt * 32 + row;
          if (qq < T) {
            const float4 v = *reinterpret_cast<const float4*>(stage + (size_t)row * SROW + c * 16);
            if (out) *reinterpret_cast<float4*>(out + (b * T + qq) * (int64_t)H * kDh + h * kDh + c * 4) = v;
            if (osp) store_split4(v, b * T + qq, h * kDh + c * 4, split_kp((int64_t)H * kDh), osp);
          }
        }
      }
    }
    __syncthreads();  // the next round's K / V fill may overwrite the staging area
    loaded = -1;
  }
}

// ---- patch extraction: (B, C, Hi, Wi) -> (B * gh * gw, C * P * P), k = c*P*P + py*P + px (conv weight order) --
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, int64_t B, int C, int Hi, int Wi,
                                                        int P, float* __restrict__ out, uint16_t* __restrict__ osp) {
  const int gh = Hi / P, gw = Wi / P;
  const int64_t kdim = (int64_t)C * P * P;
  const int64_t total4 = B * gh * gw * kdim / 4;  // P % 4 == 0
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t idx = e * 4;
    const int64_t row = idx / kdim;
    const int k = (int)(idx % kdim);
    const int c = k / (P * P), py = (k / P) % P, px = k % P;
    const int64_t bb = row / (gh * gw);
    const int pr = (int)(row % (gh * gw));
    const int gy = pr / gw, gx = pr % gw;
    const float* src = img + ((bb * C + c) * Hi + gy * P + py) * (int64_t)Wi + gx * P + px;
    const float4 v = *reinterpret_cast<const float4*>(src);
    if (out) reinterpret_cast<float4*>(out)[e] = v;
    if (osp) store_split4(v, row, k, split_kp(kdim), osp);
  }
}

// any patch size (14 x 14 of ViT-L/14 and SigLIP-so400m): one element per thread
__global__ __launch_bounds__(256) void patchify_scalar_kernel(const float* __restrict__ img, int64_t B, int C, int Hi, int Wi,
                                                               int P, float* __restrict__ out, uint16_t* __restrict__ osp) {
  const int gh = Hi / P, gw = Wi / P;
  const int64_t kdim = (int64_t)C * P * P;
  const int64_t total = B * gh * gw * kdim;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx / kdim;
    const int k = (int)(idx % kdim);
    const int c = k / (P * P), py = (k / P) % P, px = k % P;
    const int64_t bb = row / (gh * gw);
    const int pr = (int)(row % (gh * gw));
    const int gy = pr / gw, gx = pr % gw;
    const float v = img[((bb * C + c) * Hi + gy * P + py) * (int64_t)Wi + gx * P + px];
    if (out) out[idx] = v;
    if (osp) store_split(v, row, k, split_kp(kdim), osp);
  }
}

// out[g * gstride + row] = v[:] (+ add[:]) for every group g: the class token row of every image
__global__ __launch_bounds__(256) void broadcast_row_kernel(const float* __restrict__ v, const float* __restrict__ add,
                                                             int64_t G, int64_t gstride_elems, int N,
                                                             float* __restrict__ out) {
  const int64_t total = G * N;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t g = e / N;
    const int c = (int)(e % N);
    out[g * gstride_elems + c] = v[c] + (add ? add[c] : 0.f);
  }
}

// token embedding lookup + positional embedding: out[b][t][:] = table[ids[b][t]][:] + pos[t][:]
__global__ __launch_bounds__(256) void embed_tokens_kernel(const float* __restrict__ table, int64_t vocab,
                                                            const int64_t* __restrict__ ids, int64_t B, int T, int W,
                                                            const float* __restrict__ pos, float* __restrict__ out) {
  const int64_t total = B * T * (int64_t)W;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % W);
    const int64_t bt = e / W;
    const int t = (int)(bt % T);
    int64_t id = ids[bt];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    out[e] = table[id * W + c] + pos[(int64_t)t * W + c];
  }
}

// ---- attention pooling with ONE query per head (SigLIP's MAP head: softmax(q k^T / sqrt(d)) v with q = the learned
// probe, the same for every image).  One wave per (image, head): lanes stride over the T keys, each lane keeps an online
// softmax (running max, sum, weighted V) of its keys, then the 64 partial states are combined.  K / V rows are read once:
// HBM/L2-bound, B*T*2*W*4 bytes.
constexpr int kPoolMaxHd = 128;
__global__ __launch_bounds__(64) void attention_pool_kernel(const float* __restrict__ q, int64_t q_stride, const float* __restrict__ kv,
                                                            int64_t ld, int64_t voff, int T, int H, int hd, float scale,
                                                            float* __restrict__ out) {
  __shared__ float s_q[kPoolMaxHd];
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x / H;
  const int h = (int)(blockIdx.x % H);
  for (int d = lane; d < hd; d += 64) s_q[d] = q[b * q_stride + h * hd + d] * scale;  // q_stride 0: one probe for every image
  __syncthreads();
  float m = -__builtin_huge_valf(), l = 0.f;
  float acc[kPoolMaxHd];
#pragma unroll
  for (int d = 0; d < kPoolMaxHd; ++d) acc[d] = 0.f;
  for (int t = lane; t < T; t += 64) {
    const float* kr = kv + (b * T + t) * ld + h * hd;
    const float* vr = kr + voff;
    // The score is accumulated in double and the exponentials are the library's (1 ulp): behind a CLIP-ResNet trunk the
    // logits are in the hundreds, where the hardware exp2 path (`__expf`: the product x * log2(e) rounded to fp32) is off by
    // |x| * 6e-8 in the exponent, i.e. ~1e-5 relative in the weights — 3x torch's own fp32 distance from float64 on the seeded
    // head test (round 5).  hd MACs and two exps per key beside 2 * hd * 4 bytes of K / V: free for this HBM-bound kernel.
    double sd = 0.0;
    for (int d = 0; d < hd; d += 4) {
      const float4 k4 = *reinterpret_cast<const float4*>(kr + d);
      sd += (double)s_q[d] * k4.x + (double)s_q[d + 1] * k4.y + (double)s_q[d + 2] * k4.z + (double)s_q[d + 3] * k4.w;
    }
    const float s = (float)sd;
    const float mn = fmaxf(m, s);
    const float corr = expf(m - mn), p = expf(s - mn);  // first key: m = -inf -> corr = 0
    l = l * corr + p;
#pragma unroll
    for (int d = 0; d < kPoolMaxHd; d += 4) {
      if (d < hd) {
        const float4 v4 = *reinterpret_cast<const float4*>(vr + d);
        acc[d] = acc[d] * corr + p * v4.x;
        acc[d + 1] = acc[d + 1] * corr + p * v4.y;
        acc[d + 2] = acc[d + 2] * corr + p * v4.z;
        acc[d + 3] = acc[d + 3] * corr + p * v4.w;
      }
    }
    m = mn;
  }
  float M = m;
  for (int off = 32; off > 0; off >>= 1) M = fmaxf(M, __shfl_xor(M, off, 64));
  const float f = (m == -__builtin_huge_valf()) ? 0.f : expf(m - M);  // lanes without a key contribute nothing
  l *= f;
  for (int off = 32; off > 0; off >>= 1) l += __shfl_xor(l, off, 64);
  const float inv = 1.f / l;
#pragma unroll
  for (int d = 0; d < kPoolMaxHd; ++d) {
    if (d < hd) {
      float a = acc[d] * f;
      for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
      if (lane == 0) out[b * (int64_t)H * hd + h * hd + d] = a * inv;
    }
  }
}

// CLIP-ResNet attention pool input (open_clip ModifiedResNet.attnpool, foundation_models/clip.py:52-62 `OpenClip("RN50", ...)`):
// tokens[b][0][c] = mean_s map[b][c][s] + pos[0][c], tokens[b][1 + s][c] = map[b][c][s] + pos[1 + s][c] — the NCHW trunk output
// turned into (B, S + 1, C) token rows.  One workgroup per (b, 64-channel block): the (64 x S) tile goes through LDS so that both
// the reads (s contiguous) and the writes (c contiguous) are coalesced.
__global__ __launch_bounds__(256) void tokens_from_map_kernel(const float* __restrict__ map, const float* __restrict__ pos, int C, int S,
                                                              float* __restrict__ out) {
  extern __shared__ float s_tile[];  // 64 x (S + 1)
  const int ld = S + 1;
  const int64_t b = blockIdx.y;
  const int c0 = blockIdx.x * 64;
  const int nc = C - c0 < 64 ? C - c0 : 64;
  const float* src = map + (b * C + c0) * (int64_t)S;
  for (int i = threadIdx.x; i < nc * S; i += 256) s_tile[(i / S) * ld + (i % S)] = src[i];
  __syncthreads();
  if ((int)threadIdx.x < nc) {  // the mean token: fp32 sum in index order, then / S (torch: x.mean(dim=0))
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc += s_tile[threadIdx.x * ld + s];
    s_tile[threadIdx.x * ld + S] = acc / (float)S;
  }
  __syncthreads();
  float* dst = out + b * (int64_t)(S + 1) * C + c0;
  for (int i = threadIdx.x; i < 64 * (S + 1); i += 256) {
    const int t = i / 64, c = i % 64;  // t = 0: mean token
    if (c < nc) dst[(int64_t)t * C + c] = s_tile[c * ld + (t == 0 ? S : t - 1)] + pos[(int64_t)t * C + c0 + c];
  }
}

int64_t grid_for(int64_t items) {
  int64_t blocks = (items + 255) / 256;
  const int64_t cap = (int64_t)num_cus() * 16;
  if (blocks > cap) blocks = cap;
  return blocks < 1 ? 1 : blocks;
}

}  // namespace
}  // namespace sl

using namespace sl;

SL_API int sl_linear(const float* d_x, int64_t M, int64_t K, const float* d_w, int64_t N, const float* d_bias, int act,
                     const float* d_residual, float* d_out, int64_t ldo, int64_t rows_per_group, int64_t group_stride,
                     int64_t row_offset, const float* d_rowadd, void* stream) {
  SL_REQUIRE(M >= 0 && K >= 0 && N >= 0 && ldo >= N, "sl_linear: bad shape");
  SL_REQUIRE(act >= SL_ACT_NONE && act <= SL_ACT_GELU_TANH, "sl_linear: bad activation %d", act);
  if (M * N == 0) return 0;
  SL_REQUIRE(d_x && d_w && d_out, "sl_linear: null pointer");
  const bool remap = rows_per_group > 0;
  SL_REQUIRE(!(remap && (act != SL_ACT_NONE || d_residual)), "sl_linear: row scatter supports neither activation nor residual");
  SL_REQUIRE(!remap || (M < (1ll << 32) && rows_per_group < (1ll << 32)), "sl_linear: row scatter indexes rows with 32 bits");
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof(SL_PROF_GEMM, st, 2.0 * (double)M * (double)N * (double)K);
#define SL_RUN(A_, R_, P_) \
  return run_linear<A_, R_, P_>(prof, d_x, M, K, d_w, N, d_bias, d_residual, d_out, ldo, rows_per_group, group_stride, row_offset, d_rowadd, st)
  if (remap) SL_RUN(SL_ACT_NONE, false, true);
  if (d_residual) {
    if (act == SL_ACT_NONE) SL_RUN(SL_ACT_NONE, true, false);
    if (act == SL_ACT_GELU) SL_RUN(SL_ACT_GELU, true, false);
    if (act == SL_ACT_GELU_TANH) SL_RUN(SL_ACT_GELU_TANH, true, false);
    SL_RUN(SL_ACT_QUICKGELU, true, false);
  }
  if (act == SL_ACT_NONE) SL_RUN(SL_ACT_NONE, false, false);
  if (act == SL_ACT_GELU) SL_RUN(SL_ACT_GELU, false, false);
  if (act == SL_ACT_GELU_TANH) SL_RUN(SL_ACT_GELU_TANH, false, false);
  SL_RUN(SL_ACT_QUICKGELU, false, false);
#undef SL_RUN
}

SL_API int sl_layernorm(const float* d_x, int64_t rows, int64_t cols, int64_t x_row_stride, const float* d_gamma,
                        const float* d_beta, float eps, float* d_out, uint16_t* d_out_split, int64_t out_row_stride,
                        void* stream) {
  SL_REQUIRE(rows >= 0 && cols >= 1 && cols < (1 << 30), "sl_layernorm: bad shape");
  if (rows == 0) return 0;
  SL_REQUIRE(d_x && d_gamma && d_beta && (d_out || d_out_split), "sl_layernorm: null pointer");
  int64_t blocks = (rows + 3) / 4;
  const int64_t cap = (int64_t)num_cus() * 8;
  if (blocks > cap) blocks = cap;
  const uintptr_t ptrs = (uintptr_t)d_x | (uintptr_t)d_gamma | (uintptr_t)d_beta | (uintptr_t)d_out | (uintptr_t)d_out_split;
  const bool fast = cols % 4 == 0 && cols <= 2048 && x_row_stride % 4 == 0 && out_row_stride % 4 == 0 && (ptrs & 15) == 0;
  if (fast && cols <= 1024)
    hipLaunchKernelGGL(layernorm_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_x, rows, (int)cols,
                       x_row_stride, d_gamma, d_beta, eps, d_out, out_row_stride, d_out_split);
  else if (fast)
    hipLaunchKernelGGL(layernorm_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_x, rows, (int)cols,
                       x_row_stride, d_gamma, d_beta, eps, d_out, out_row_stride, d_out_split);
  else
    hipLaunchKernelGGL(layernorm_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_x, rows, (int)cols,
                       x_row_stride, d_gamma, d_beta, eps, d_out, out_row_stride, d_out_split);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

template <int D>
static int launch_attention_mfma(const float* qkv, int64_t B, int64_t T, int64_t H, int causal, float* out, uint16_t* osp,
                                 hipStream_t st) {
  const int64_t Tp = (T + 31) & ~(int64_t)31;
  const int64_t kc = Tp < attn_chunk(D) ? Tp : attn_chunk(D);
  const size_t smem = (size_t)kc * attn_ld(D) * 4 * 2;
  const int waves = (int)(Tp / 32 < 4 ? Tp / 32 : 4);
  if (smem > 64 * 1024)
    SL_CHECK_HIP(hipFuncSetAttribute((const void*)attention_mfma_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const float scale = (float)(1.0 / sqrt((double)D));  // torch: q * head_dim ** -0.5
  hipLaunchKernelGGL(attention_mfma_kernel<D>, dim3((unsigned)(B * H)), dim3(64 * waves), smem, st, qkv, (int)T, (int)H, causal,
                     scale, out, osp);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

SL_API int sl_attention(const float* d_qkv, int64_t B, int64_t T, int64_t H, int64_t head_dim, int causal, float* d_out,
                        uint16_t* d_out_split, void* stream) {
  SL_REQUIRE(B >= 0 && T >= 1 && H >= 1, "sl_attention: bad shape");
  SL_REQUIRE(T < (1 << 24), "sl_attention: sequence length %lld too long", (long long)T);
  if (B == 0) return 0;
  SL_REQUIRE(d_qkv && (d_out || d_out_split), "sl_attention: null pointer");
  SL_REQUIRE(B * H < (1ll << 31), "sl_attention: too many heads");
  hipStream_t st = (hipStream_t)stream;
  switch (head_dim) {
    case 32: return launch_attention_mfma<32>(d_qkv, B, T, H, causal, d_out, d_out_split, st);
    case 64: return launch_attention_mfma<64>(d_qkv, B, T, H, causal, d_out, d_out_split, st);
    case 72: return launch_attention_mfma<72>(d_qkv, B, T, H, causal, d_out, d_out_split, st);
    case 80: return launch_attention_mfma<80>(d_qkv, B, T, H, causal, d_out, d_out_split, st);
    case 88: return launch_attention_mfma<88>(d_qkv, B, T, H, causal, d_out, d_out_split, st);
    case 96: return launch_attention_mfma<96>(d_qkv, B, T, H, causal, d_out, d_out_split, st);
    case 104: return launch_attention_mfma<104>(d_qkv, B, T, H, causal, d_out, d_out_split, st);
    case 128: return launch_attention_mfma<128>(d_qkv, B, T, H, causal, d_out, d_out_split, st);
    default: break;
  }
  SL_REQUIRE(false, "sl_attention: head_dim=%lld (built: 32, 64, 72, 80, 88, 96, 104, 128)", (long long)head_dim);
}

template <int D>
static int launch_attention_bf16x3(const float* qkv, int64_t B, int64_t T, int64_t H, int causal, float* out, uint16_t* osp,
                                   hipStream_t st) {
  const int64_t Tp = (T + 31) & ~(int64_t)31;
  const int64_t kc = Tp < attn3_kc() ? Tp : attn3_kc();
  const int NT = (D + 31) / 32;
  const size_t smem = 2 * (size_t)kc * (attn3_dp(D) * 2 + 16) + 2 * (size_t)(NT * 32) * (kc * 2 + 16);
  const float scale = (float)(1.0 / sqrt((double)D));
  if (Tp / 32 > 4 && D <= 96) {  // head_dim 104 / 128 would spill at 256 registers per wave
    const int waves = (int)(Tp / 32 < 8 ? Tp / 32 : 8);
    if (smem > 64 * 1024)
      SL_CHECK_HIP(hipFuncSetAttribute((const void*)attention_bf16x3_kernel<D, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL((attention_bf16x3_kernel<D, 8>), dim3((unsigned)(B * H)), dim3(64 * waves), smem, st, qkv, (int)T, (int)H, causal,
                       scale, out, osp);
  } else {
    const int waves = (int)(Tp / 32 < 4 ? Tp / 32 : 4);
    if (smem > 64 * 1024)
      SL_CHECK_HIP(hipFuncSetAttribute((const void*)attention_bf16x3_kernel<D, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL((attention_bf16x3_kernel<D, 4>), dim3((unsigned)(B * H)), dim3(64 * waves), smem, st, qkv, (int)T, (int)H, causal,
                       scale, out, osp);
  }
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

SL_API int sl_attention_bf16x3(const float* d_qkv, int64_t B, int64_t T, int64_t H, int64_t head_dim, int causal, float* d_out,
                               uint16_t* d_out_split, void* stream) {
  SL_REQUIRE(B >= 0 && T >= 1 && H >= 1, "sl_attention_bf16x3: bad shape");
  SL_REQUIRE(T < (1 << 24), "sl_attention_bf16x3: sequence length %lld too long", (long long)T);
  if (B == 0) return 0;
  SL_REQUIRE(d_qkv && (d_out || d_out_split), "sl_attention_bf16x3: null pointer");
  SL_REQUIRE(B * H < (1ll << 31), "sl_attention_bf16x3: too many heads");
  hipStream_t st = (hipStream_t)stream;
  switch (head_dim) {
    case 32: return launch_attention_bf16x3<32>(d_qkv, B, T, H, causal, d_out, d_out_split, st);
    case 64: return launch_attention_bf16x3<64>(d_qkv, B, T, H, causal, d_out, d_out_split, st);
    case 72: return launch_attention_bf16x3<72>(d_qkv, B, T, H, causal, d_out, d_out_split, st);
    case 80: return launch_attention_bf16x3<80>(d_qkv, B, T, H, causal, d_out, d_out_split, st);
    case 88: return launch_attention_bf16x3<88>(d_qkv, B, T, H, causal, d_out, d_out_split, st);
    case 96: return launch_attention_bf16x3<96>(d_qkv, B, T, H, causal, d_out, d_out_split, st);
    case 104: return launch_attention_bf16x3<104>(d_qkv, B, T, H, causal, d_out, d_out_split, st);
    case 128: return launch_attention_bf16x3<128>(d_qkv, B, T, H, causal, d_out, d_out_split, st);
    default: break;
  }
  SL_REQUIRE(false, "sl_attention_bf16x3: head_dim=%lld (built: 32, 64, 72, 80, 88, 96, 104, 128)", (long long)head_dim);
}

static int attention_pool_impl(const char* fn, const float* d_q, int64_t q_batch_stride, const float* d_kv, int64_t kv_row_stride,
                               int64_t v_offset, int64_t B, int64_t T, int64_t H, int64_t head_dim, float* d_out, void* stream) {
  SL_REQUIRE(B >= 0 && T >= 1 && H >= 1, "%s: bad shape", fn);
  SL_REQUIRE(head_dim >= 4 && head_dim <= kPoolMaxHd && head_dim % 4 == 0, "%s: head_dim=%lld not a multiple of 4 up to %d", fn,
             (long long)head_dim, kPoolMaxHd);
  if (B == 0) return 0;
  SL_REQUIRE(d_q && d_kv && d_out, "%s: null pointer", fn);
  SL_REQUIRE(kv_row_stride % 4 == 0 && v_offset % 4 == 0 && (((uintptr_t)d_kv) & 15) == 0, "%s: rows must be 16-byte aligned", fn);
  SL_REQUIRE(B * H < (1ll << 31) && q_batch_stride >= 0, "%s: too many heads", fn);
  SL_REQUIRE(T < (1ll << 31), "%s: T=%lld exceeds the 32-bit key index", fn, (long long)T);
  hipLaunchKernelGGL(attention_pool_kernel, dim3((unsigned)(B * H)), dim3(64), 0, (hipStream_t)stream, d_q, q_batch_stride, d_kv,
                     kv_row_stride, v_offset, (int)T, (int)H, (int)head_dim, 1.f / sqrtf((float)head_dim), d_out);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

SL_API int sl_attention_pool(const float* d_q, const float* d_kv, int64_t kv_row_stride, int64_t v_offset, int64_t B, int64_t T,
                             int64_t H, int64_t head_dim, float* d_out, void* stream) {
  return attention_pool_impl("sl_attention_pool", d_q, 0, d_kv, kv_row_stride, v_offset, B, T, H, head_dim, d_out, stream);
}

SL_API int sl_attention_pool_q(const float* d_q, int64_t q_batch_stride, const float* d_kv, int64_t kv_row_stride, int64_t v_offset,
                               int64_t B, int64_t T, int64_t H, int64_t head_dim, float* d_out, void* stream) {
  return attention_pool_impl("sl_attention_pool_q", d_q, q_batch_stride, d_kv, kv_row_stride, v_offset, B, T, H, head_dim, d_out,
                             stream);
}

SL_API int sl_tokens_from_map(const float* d_map, int64_t B, int64_t C, int64_t S, const float* d_pos, float* d_out, void* stream) {
  SL_REQUIRE(B >= 0 && C >= 1 && S >= 1, "sl_tokens_from_map: bad shape");
  if (B == 0) return 0;
  SL_REQUIRE(d_map && d_pos && d_out, "sl_tokens_from_map: null pointer");
  const size_t lds = (size_t)64 * (S + 1) * sizeof(float);
  // the (64 channels x S positions) tile lives in LDS: S is bounded by the 160 KiB a gfx950 workgroup may declare (S <= 639)
  SL_REQUIRE(B < 65536 && lds <= 160 * 1024, "sl_tokens_from_map: B=%lld S=%lld (limits 65535 / %d: the 64 x (S + 1) fp32 tile must fit the LDS)",
             (long long)B, (long long)S, (int)(160 * 1024 / (64 * sizeof(float)) - 1));
  if (lds > 64 * 1024)
    SL_CHECK_HIP(hipFuncSetAttribute((const void*)tokens_from_map_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(tokens_from_map_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)B), dim3(256), lds, (hipStream_t)stream, d_map, d_pos,
                     (int)C, (int)S, d_out);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

SL_API int sl_patchify(const float* d_img, int64_t B, int64_t C, int64_t Hi, int64_t Wi, int64_t P, float* d_out,
                       uint16_t* d_out_split, void* stream) {
  SL_REQUIRE(B >= 0 && C >= 1 && P >= 1 && Hi % P == 0 && Wi % P == 0, "sl_patchify: bad geometry");
  if (B == 0) return 0;
  SL_REQUIRE(d_img && (d_out || d_out_split), "sl_patchify: null pointer");
  if (P % 4 == 0 && (((uintptr_t)d_img | (uintptr_t)d_out) & 15) == 0 && Wi % 4 == 0) {  // 16-byte pieces
    const int64_t total4 = B * C * Hi * Wi / 4;
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)grid_for(total4)), dim3(256), 0, (hipStream_t)stream, d_img, B, (int)C,
                       (int)Hi, (int)Wi, (int)P, d_out, d_out_split);
  } else {
    hipLaunchKernelGGL(patchify_scalar_kernel, dim3((unsigned)grid_for(B * C * Hi * Wi)), dim3(256), 0, (hipStream_t)stream, d_img,
                       B, (int)C, (int)Hi, (int)Wi, (int)P, d_out, d_out_split);
  }
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

SL_API int sl_broadcast_row(const float* d_v, const float* d_add, int64_t G, int64_t group_stride_elems, int64_t N,
                            float* d_out, void* stream) {
  SL_REQUIRE(G >= 0 && N >= 1, "sl_broadcast_row: bad shape");
  if (G == 0) return 0;
  SL_REQUIRE(d_v && d_out, "sl_broadcast_row: null pointer");
  hipLaunchKernelGGL(broadcast_row_kernel, dim3((unsigned)grid_for(G * N)), dim3(256), 0, (hipStream_t)stream, d_v, d_add, G,
                     group_stride_elems, (int)N, d_out);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

SL_API int sl_embed_tokens(const float* d_table, int64_t vocab, const int64_t* d_ids, int64_t B, int64_t T, int64_t W,
                           const float* d_pos, float* d_out, void* stream) {
  SL_REQUIRE(B >= 0 && T >= 1 && W >= 1 && vocab >= 1, "sl_embed_tokens: bad shape");
  if (B == 0) return 0;
  SL_REQUIRE(d_table && d_ids && d_pos && d_out, "sl_embed_tokens: null pointer");
  hipLaunchKernelGGL(embed_tokens_kernel, dim3((unsigned)grid_for(B * T * W)), dim3(256), 0, (hipStream_t)stream, d_table,
                     vocab, d_ids, B, (int)T, (int)W, d_pos, d_out);
  SL_CHECK_HIP(hipGetLastError());
  return 0;
}

SL_API size_t sl_split_elems(int64_t R, int64_t K) { return (R < 0 || K < 0) ? 0 : gemm3::split_elems(R, K); }

SL_API int sl_split_bf16(const float* d_x, const float* d_row_scale, int64_t R, int64_t K, uint16_t* d_split, void* stream) {
  SL_REQUIRE(R >= 0 && K >= 0, "sl_split_bf16: negative shape");
  if (R * K == 0) return 0;
  SL_REQUIRE(d_x && d_split && ((uintptr_t)d_split & 127) == 0, "sl_split_bf16: null or unaligned pointer (128-byte lines)");
  return gemm3::launch_split(d_x, d_row_scale, R, K, d_split, (hipStream_t)stream);
}

SL_API int sl_linear_bf16x3(const uint16_t* d_x_split, int64_t M, int64_t K, const uint16_t* d_w_split, int64_t N,
                            const float* d_bias, int act, const float* d_residual, float* d_out, uint16_t* d_out_split,
                            int64_t ldo, int64_t rows_per_group, int64_t group_stride, int64_t row_offset,
                            const float* d_rowadd, void* stream) {
  SL_REQUIRE(M >= 0 && K >= 0 && N >= 0 && ldo >= N, "sl_linear_bf16x3: bad shape");
  SL_REQUIRE(act >= SL_ACT_NONE && act <= SL_ACT_GELU_TANH, "sl_linear_bf16x3: bad activation %d", act);
  if (M * N == 0) return 0;
  const bool split = d_out_split != nullptr;
  SL_REQUIRE(d_x_split && d_w_split && (d_out || split) && !(d_out && split), "sl_linear_bf16x3: null / ambiguous pointers");
  SL_REQUIRE((((uintptr_t)d_x_split | (uintptr_t)d_w_split | (uintptr_t)d_out_split) & 127) == 0,
             "sl_linear_bf16x3: split matrices must be 128-byte aligned");
  const bool remap = rows_per_group > 0;
  SL_REQUIRE(!(remap && (act != SL_ACT_NONE || d_residual || split)), "sl_linear_bf16x3: row scatter is plain fp32 only");
  SL_REQUIRE(!remap || (M < (1ll << 32) && rows_per_group < (1ll << 32)), "sl_linear_bf16x3: row scatter indexes rows with 32 bits");
  SL_REQUIRE(!(split && d_residual), "sl_linear_bf16x3: split output takes no residual");
  hipStream_t st = (hipStream_t)stream;
  ProfScope prof(SL_PROF_GEMM, st, 2.0 * (double)M * (double)N * (double)K);
#define SL_RUN3(A_, R_, P_, S_) \
  return run_linear3<A_, R_, P_, S_>(prof, d_x_split, M, K, d_w_split, N, d_bias, d_residual, d_out, d_out_split, ldo, rows_per_group, group_stride, row_offset, d_rowadd, st)
  if (remap) SL_RUN3(SL_ACT_NONE, false, true, false);
  if (split) {
    if (act == SL_ACT_NONE) SL_RUN3(SL_ACT_NONE, false, false, true);
    if (act == SL_ACT_GELU) SL_RUN3(SL_ACT_GELU, false, false, true);
    if (act == SL_ACT_GELU_TANH) SL_RUN3(SL_ACT_GELU_TANH, false, false, true);
    SL_RUN3(SL_ACT_QUICKGELU, false, false, true);
  }
  if (d_residual) {
    if (act == SL_ACT_NONE) SL_RUN3(SL_ACT_NONE, true, false, false);
    if (act == SL_ACT_GELU) SL_RUN3(SL_ACT_GELU, true, false, false);
    if (act == SL_ACT_GELU_TANH) SL_RUN3(SL_ACT_GELU_TANH, true, false, false);
    SL_RUN3(SL_ACT_QUICKGELU, true, false, false);
  }
  if (act == SL_ACT_NONE) SL_RUN3(SL_ACT_NONE, false, false, false);
  if (act == SL_ACT_GELU) SL_RUN3(SL_ACT_GELU, false, false, false);
  if (act == SL_ACT_GELU_TANH) SL_RUN3(SL_ACT_GELU_TANH, false, false, false);
  SL_RUN3(SL_ACT_QUICKGELU, false, false, false);
#undef SL_RUN3
}
