// Timing experiment (not product code): how long does the LDS-DMA of one 48 KB k-tile (256 A rows + 128 B rows of
// 128-byte [hi | lo] lines) take per workgroup when NW waves share the 48 one-KiB global_load_lds instructions, with
// no MFMA work at all?  Answers whether the ~4 900-cycle round trip seen in gemm3_nt_dma256_kernel is per-wave
// serialisation of the DMA instructions or memory-side latency.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 feed_probe.hip -o feed_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int IMGA = 256 * 128, IMGB = 128 * 128;

template <int NW>  // waves per workgroup: 4, 8 or 16
__global__ __launch_bounds__(NW * 64) void k_feed(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, int64_t M, int64_t N,
                                                  int64_t Kp, int tiles_n, unsigned long long* __restrict__ probe) {
  __shared__ __align__(1024) unsigned char smem[IMGA + IMGB];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * 256, n0 = (int64_t)(blockIdx.x % tiles_n) * 128;
  constexpr int NA = 32 / NW, NB = 16 / NW;  // 1-KiB loads per wave for A (32 in total) and B (16)
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int lrow = lane >> 3;
  int64_t a_src[NA], b_src[NB > 0 ? NB : 1];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = (w * NA + i) * 8 + lrow;
    a_src[i] = (m0 + row < M ? m0 + row : M - 1) * 2 * Kp + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int row = (w * NB + i) * 8 + lrow;
    b_src[i] = (n0 + row < N ? n0 + row : N - 1) * 2 * Kp + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
  }
  const int nt = (int)(Kp / 32);
  for (int kt = 0; kt < nt; ++kt) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
      __builtin_amdgcn_global_load_lds((glb_void*)(A + a_src[i] + kt * 64), (lds_void*)(smem + (w * NA + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < NB; ++i)
      __builtin_amdgcn_global_load_lds((glb_void*)(B + b_src[i] + kt * 64), (lds_void*)(smem + IMGA + (w * NB + i) * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
  }
  if (tid == 0) probe[blockIdx.x] = __builtin_amdgcn_s_memtime() - c0 + smem[lane];
}

template <int NW>
static void run(const uint16_t* A, const uint16_t* B, int64_t M, int64_t N, int64_t Kp, unsigned long long* st, int blocks_per_cu_hint) {
  const int tm = (M + 255) / 256, tn = (N + 127) / 128;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_feed<NW>, dim3(tm * tn), dim3(NW * 64), 0, nullptr, A, B, M, N, Kp, tn, st);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, nullptr);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_feed<NW>, dim3(tm * tn), dim3(NW * 64), 0, nullptr, A, B, M, N, Kp, tn, st);
  hipEventRecord(e1, nullptr);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> hs(tm * tn); hipMemcpy(hs.data(), st, 8 * tm * tn, hipMemcpyDeviceToHost);
  double cyc = 0; for (auto v : hs) cyc += (double)v;
  const double bytes = (double)tm * tn * (Kp / 32) * 49152.0;
  printf("%2d waves per workgroup: %.0f cycles per k-tile and workgroup, %.3f ms per launch, %.1f TB/s L2 -> LDS\n", NW,
         cyc / (tm * tn) / (Kp / 32), ms / 20, bytes / (ms / 20 * 1e-3) / 1e12);
  (void)blocks_per_cu_hint;
}

int main() {
  const int64_t M = 10000, N = 9216, K = 1152;
  uint16_t *A, *B; unsigned long long* st;
  hipMalloc(&A, M * K * 4); hipMalloc(&B, N * K * 4); hipMalloc(&st, 8 * 65536);
  hipMemset(A, 1, M * K * 4); hipMemset(B, 1, N * K * 4);
  run<4>(A, B, M, N, K, st, 3);
  run<8>(A, B, M, N, K, st, 3);
  run<16>(A, B, M, N, K, st, 3);
  return 0;
}
