// Measures the shader clock a GEMM kernel actually runs at: s_memtime (shader cycles) against s_memrealtime
// (100 MHz constant) per workgroup, for the split-bf16 x3 kernel and the fp32-MFMA kernel, with random and with
// all-zero operands (data-dependent power).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../semanticlens_amd/csrc
#define SL_GEMM_CLOCKPROBE 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "gemm_f32.hpp"
#include "gemm_bf16x3.hpp"

namespace sl {
void set_error(const char*, ...) {}
int hip_fail(hipError_t e, const char* what) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return -1; }
ProfScope::ProfScope(int, hipStream_t, double) : start(nullptr), stop(nullptr) {}
}

struct ProbeEpi {
  float* out; int64_t ldc; unsigned long long* probe;
  __device__ float column(int64_t) const { return 0.f; }
  __device__ void store(int64_t r, int64_t c, float acc, float) const { out[r * ldc + c] = acc; }
};


static double run(bool zero, bool f32mode, int64_t M, int64_t N, int64_t K) {
  // split matrices: rows of 2 K bf16 ([hi32 | lo32] per k-tile); random bit patterns stand in for real operands
  std::vector<uint16_t> h((size_t)std::max(M, N) * K * 2);
  std::vector<float> hf((size_t)std::max(M, N) * K);
  srand(1);
  for (auto& v : h) v = zero ? 0 : (uint16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  for (auto& v : hf) v = zero ? 0.f : (float)rand() / RAND_MAX - 0.5f;
  uint16_t *As, *Bs; float *Af, *Bf, *out; unsigned long long* st;
  hipMalloc(&As, M * K * 4); hipMalloc(&Bs, N * K * 4);
  hipMalloc(&Af, M * K * 4); hipMalloc(&Bf, N * K * 4);
  hipMalloc(&out, M * N * 4); hipMalloc(&st, 16 * 65536);
  hipMemcpy(As, h.data(), M * K * 4, hipMemcpyHostToDevice); hipMemcpy(Bs, h.data(), N * K * 4, hipMemcpyHostToDevice);
  hipMemcpy(Af, hf.data(), M * K * 4, hipMemcpyHostToDevice); hipMemcpy(Bf, hf.data(), N * K * 4, hipMemcpyHostToDevice);
  ProbeEpi epi{out, N, st};
  sl::ProfScope prof(-1, nullptr, 0.0);
  auto go = [&]() {
    if (f32mode) sl::gemm::launch_gemm_nt(prof, Af, M, Bf, N, K, epi, nullptr);
    else sl::gemm3::launch_gemm3_nt(prof, As, M, Bs, N, K, epi, nullptr);
  };
  for (int i = 0; i < 5; ++i) go();
  hipDeviceSynchronize();
  const int reps = 40;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, nullptr);
  for (int i = 0; i < reps; ++i) go();
  hipEventRecord(e1, nullptr);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const char* tile_env = getenv("SL_G3_TILE");
  const int tile_sel = (!f32mode && tile_env) ? atoi(tile_env) : 128;
  const int64_t nblk = tile_sel == 512 ? ((M + 255) / 256) * ((N + 255) / 256) : tile_sel == 256 ? ((M + 255) / 256) * ((N + 127) / 128) : ((M + 127) / 128) * ((N + 127) / 128);
  std::vector<unsigned long long> hs(2 * nblk); hipMemcpy(hs.data(), st, 16 * nblk, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int64_t b = 0; b < nblk; ++b) { cyc += (double)hs[2 * b]; rt += (double)hs[2 * b + 1]; }
  const double real = ms * 1e-3;
  const double tf = 2.0 * M * N * K * reps / real / 1e12;
  printf("  per-workgroup: %.0f shader cycles, %.2f us  ", cyc / nblk, rt / nblk / 100.0);
  rt = rt / 100e6;
  printf("%s %s: %.3f ms/launch, %.1f TFLOP/s algorithmic, shader clock %.0f MHz\n", f32mode ? "f32-mfma" : "bf16x3  ",
         zero ? "zeros " : "random", real / reps * 1e3, tf, cyc / rt / 1e6);
  hipFree(As); hipFree(Bs); hipFree(Af); hipFree(Bf); hipFree(out); hipFree(st);
  return tf;
}

int main() {
  const int64_t M = 10000, N = 9216, K = 1152;
  run(false, false, M, N, K); run(true, false, M, N, K);
  run(false, true, M, N, K); run(true, true, M, N, K);
  return 0;
}
