// Timing experiment (not product code): the 256 x 128 LDS-DMA staged bf16x3 kernel with operands stored so that the hi
// and lo halves of a 32-wide k-tile share one 128-byte line ([row][k-tile][hi 64 B | lo 64 B]); one DMA instruction
// then covers 8 rows x 128 B = whole lines.  Compared against the production layout (separate hi / lo matrices, 64 B of
// each line per k-tile).  Results are not checked; only the cycle count per workgroup matters.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../semanticlens_amd/csrc fullline_probe.hip -o fullline_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 256, BN = 128;
constexpr int IMGA = BM * 128, IMGB = BN * 128;  // rows of 128 B: [hi 64 | lo 64]

__global__ __launch_bounds__(256, 2) void k_fullline(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, int64_t M,
                                                     int64_t N, int64_t K, int tiles_n, float* __restrict__ out,
                                                     unsigned long long* __restrict__ probe) {
  __shared__ __align__(1024) unsigned char smem[IMGA + IMGB];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1, li = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * BM, n0 = (int64_t)(blockIdx.x % tiles_n) * BN;
  floatx16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // DMA: 8 rows x 128 B per instruction; wave w: A rows [64 w, +64) = 8 instructions, B rows [32 w, +32) = 4
  const int lrow = lane >> 3;
  int64_t a_src[8], b_src[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = w * 64 + i * 8 + lrow;
    const int gch = (lane & 7) ^ ((row >> 1) & 7);
    a_src[i] = (m0 + row < M ? m0 + row : M - 1) * 2 * K + gch * 8;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = w * 32 + i * 8 + lrow;
    const int gch = (lane & 7) ^ ((row >> 1) & 7);
    b_src[i] = (n0 + row < N ? n0 + row : N - 1) * 2 * K + gch * 8;
  }
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  int a_off[4], a_sw[4], b_off[2], b_sw[2];
#pragma unroll
  for (int t = 0; t < 4; ++t) { const int r = wm * 128 + t * 32 + li; a_off[t] = r * 128; a_sw[t] = (r >> 1) & 7; }
#pragma unroll
  for (int t = 0; t < 2; ++t) { const int r = wn * 64 + t * 32 + li; b_off[t] = IMGA + r * 128; b_sw[t] = (r >> 1) & 7; }
  const int nt = (int)(K / 32);
  for (int kt = 0; kt < nt; ++kt) {
    const int64_t k0 = (int64_t)kt * 64;  // 64 uint16 per k-tile per row (hi 32 + lo 32)
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((glb_void*)(A + a_src[i] + k0), (lds_void*)(smem + (w * 64 + i * 8) * 128), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((glb_void*)(B + b_src[i] + k0), (lds_void*)(smem + IMGA + (w * 32 + i * 8) * 128), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = ks * 2 + lh;
      bf16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bh[t] = *reinterpret_cast<const bf16x8*>(smem + b_off[t] + ((c ^ b_sw[t]) << 4));
        bl[t] = *reinterpret_cast<const bf16x8*>(smem + b_off[t] + (((4 + c) ^ b_sw[t]) << 4));
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        ah[t] = *reinterpret_cast<const bf16x8*>(smem + a_off[t] + ((c ^ a_sw[t]) << 4));
        al[t] = *reinterpret_cast<const bf16x8*>(smem + a_off[t] + (((4 + c) ^ a_sw[t]) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t col = n0 + wn * 64 + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < M && col < N) out[row * N + col] = acc[i][j][r];
      }
    }
  if (tid == 0) {
    probe[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
    probe[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

int main() {
  const int64_t M = 10000, N = 9216, K = 1152;
  std::vector<uint16_t> h((size_t)M * K * 2);
  srand(1);
  for (auto& v : h) v = (uint16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  uint16_t *A, *B; float* out; unsigned long long* st;
  hipMalloc(&A, M * K * 4 + 256); hipMalloc(&B, N * K * 4 + 256); hipMalloc(&out, M * N * 4); hipMalloc(&st, 16 * 65536);
  hipMemcpy(A, h.data(), M * K * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), N * K * 4, hipMemcpyHostToDevice);
  const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_fullline, dim3(tm * tn), dim3(256), 0, nullptr, A, B, M, N, K, tn, out, st);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 40;
  hipEventRecord(e0, nullptr);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_fullline, dim3(tm * tn), dim3(256), 0, nullptr, A, B, M, N, K, tn, out, st);
  hipEventRecord(e1, nullptr);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> hs(2 * tm * tn); hipMemcpy(hs.data(), st, 16 * tm * tn, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int b = 0; b < tm * tn; ++b) { cyc += (double)hs[2 * b]; rt += (double)hs[2 * b + 1]; }
  printf("full-line layout, 256x128 DMA kernel: per-workgroup %.0f shader cycles, %.3f ms/launch, %.1f TFLOP/s algorithmic, clock %.0f MHz\n",
         cyc / (tm * tn), ms / reps, 2.0 * M * N * K * reps / (ms * 1e-3) / 1e12, cyc / (rt / 100e6) / 1e6);
  return 0;
}
