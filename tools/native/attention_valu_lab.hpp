// LAB ONLY — not part of libsemanticlens_hip.so.  The tower's first attention kernel (round 1: VALU, four lanes per query row,
// head_dim 64, T <= 256), superseded by attention_mfma_kernel / attention_bf16x3_kernel (encoder.hip) and retired in round 6.
// Kept as source for A/B history; include it after encoder.hip's helpers (bits_f32, f32_bits, store_split, split_kp) to build it.
#pragma once

// ---- multi-head attention, head_dim 64 --------------------------------------------------------------------
// qkv: (B*T, 3*H*64) rows [q | k | v], head h at columns h*64.  softmax(q k^T / sqrt(64)) v, optional causal mask.
// One 256-thread workgroup per (batch, head): K and V of the head sit in LDS (T x 64 floats each); a query row is
// owned by 4 adjacent lanes, each holding 16 of the 64 dims of q and of the output accumulator, so a wave covers 16
// rows and the workgroup 64 rows per pass.  Scores are reduced over the 4 lanes with two quad DPP adds and fed
// to an online softmax (running max / sum, accumulator rescaled per key), so no T x T score buffer exists and
// ~6 waves per SIMD are resident (the first version, one wave per head with 64-float q and o arrays per lane
// and a score buffer, ran at one wave per SIMD and took 39 % of the encoder's time).
constexpr int kDh = 64;
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, int T, int H, int causal,
                                                         float* __restrict__ out, uint16_t* __restrict__ osp) {
  extern __shared__ __align__(16) float smem[];
  float* sK = smem;                    // T x 64
  float* sV = smem + (size_t)T * kDh;  // T x 64
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x / H;
  const int h = blockIdx.x % H;
  const int64_t ld = 3ll * H * kDh;
  const float* base = qkv + b * T * ld + h * kDh;
  for (int e = tid; e < T * (kDh / 4); e += 256) {
    const int t = e / (kDh / 4), c = e % (kDh / 4);
    reinterpret_cast<float4*>(sK)[e] = *reinterpret_cast<const float4*>(base + t * ld + (int64_t)H * kDh + c * 4);
    reinterpret_cast<float4*>(sV)[e] = *reinterpret_cast<const float4*>(base + t * ld + 2ll * H * kDh + c * 4);
  }
  __syncthreads();
  const int part = tid & 3;  // which 16 dims of the head this lane owns
  for (int r0 = 0; r0 < T; r0 += 64) {
    const int i = r0 + (tid >> 2);
    const bool active = i < T;
    float q[16], o[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (active) v = *reinterpret_cast<const float4*>(base + (int64_t)i * ld + part * 16 + c * 4);
      q[4 * c + 0] = v.x * 0.125f; q[4 * c + 1] = v.y * 0.125f; q[4 * c + 2] = v.z * 0.125f; q[4 * c + 3] = v.w * 0.125f;
      o[4 * c + 0] = 0.f; o[4 * c + 1] = 0.f; o[4 * c + 2] = 0.f; o[4 * c + 3] = 0.f;
    }
    // keys any row of this wave may need: rows of a wave are r0 + 16*w .. + 15
    const int wave_last_row = r0 + (tid >> 6) * 16 + 15;
    const int jmax = causal ? (wave_last_row + 1 < T ? wave_last_row + 1 : T) : T;
    float m = -__builtin_huge_valf(), l = 0.f;
    for (int j = 0; j < jmax; ++j) {
      const float4* kj = reinterpret_cast<const float4*>(sK + (size_t)j * kDh + part * 16);
      float sp = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 kv = kj[c];
        sp += q[4 * c] * kv.x + q[4 * c + 1] * kv.y + q[4 * c + 2] * kv.z + q[4 * c + 3] * kv.w;
      }
      // sum over the 4 lanes of the row (quad_perm [1,0,3,2] then [2,3,0,1])
      sp += bits_f32((uint32_t)__builtin_amdgcn_update_dpp(0, (int)f32_bits(sp), 0xB1, 0xF, 0xF, false));
      sp += bits_f32((uint32_t)__builtin_amdgcn_update_dpp(0, (int)f32_bits(sp), 0x4E, 0xF, 0xF, false));
      const bool masked = causal && j > i;
      const float s = masked ? -__builtin_huge_valf() : sp;
      const float mn = fmaxf(m, s);
      const float alpha = expf(m - mn);          // first key: exp(-inf) = 0
      const float pj = masked ? 0.f : expf(s - mn);
      l = l * alpha + pj;
      m = mn;
      const float4* vj = reinterpret_cast<const float4*>(sV + (size_t)j * kDh + part * 16);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 vv = vj[c];
        o[4 * c + 0] = o[4 * c + 0] * alpha + pj * vv.x;
        o[4 * c + 1] = o[4 * c + 1] * alpha + pj * vv.y;
        o[4 * c + 2] = o[4 * c + 2] * alpha + pj * vv.z;
        o[4 * c + 3] = o[4 * c + 3] * alpha + pj * vv.w;
      }
    }
    if (active) {
      const float inv = 1.f / l;
      const int64_t o0 = (b * T + i) * (int64_t)H * kDh + h * kDh + part * 16;
      if (out) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          *reinterpret_cast<float4*>(out + o0 + c * 4) = make_float4(o[4 * c] * inv, o[4 * c + 1] * inv, o[4 * c + 2] * inv, o[4 * c + 3] * inv);
      }
      if (osp) {
#pragma unroll
        for (int d = 0; d < 16; ++d) store_split(o[d] * inv, b * T + i, h * kDh + part * 16 + d, split_kp((int64_t)H * kDh), osp);
      }
    }
  }
}

