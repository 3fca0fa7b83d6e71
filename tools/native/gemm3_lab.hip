// A/B bench + bit-identity check of the split-bf16 x3 GEMM kernels on random operands (cdna_hip_programming.md §5.4
// rules 24/25: interleaved rounds in ONE process, random data, median and min reported).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../semanticlens_amd/csrc gemm3_lab.hip -o gemm3_lab
// Run:   ./gemm3_lab [M N K] [rounds]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "gemm_bf16x3.hpp"
#include "../../tools/native/gemm3_pingpong_lab.hpp"
#include "gemm_f32.hpp"

namespace sl {
void set_error(const char*, ...) {}
int hip_fail(hipError_t e, const char* what) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return -1; }
ProfScope::ProfScope(int, hipStream_t, double) : start(nullptr), stop(nullptr) {}
}

struct PlainEpi {
  float* out; int64_t N;
#ifdef SL_GEMM_CLOCKPROBE
  unsigned long long* probe;
#endif
  __device__ float column(int64_t) const { return 0.f; }
  __device__ void store(int64_t r, int64_t c, float acc, float) const { out[r * N + c] = acc; }
};

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
  int64_t M = 10000, N = 9216, K = 1152;
  int rounds = 7;
  if (argc >= 4) { M = atoll(argv[1]); N = atoll(argv[2]); K = atoll(argv[3]); }
  if (argc >= 5) rounds = atoi(argv[4]);
  using namespace sl::gemm3;
  const int64_t Kp = split_kp(K);
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> ha((size_t)M * K), hb((size_t)N * K);
  for (auto& v : ha) v = nd(rng);
  for (auto& v : hb) v = nd(rng);
  // normalised rows, like the cosine GEMM's operands
  auto normalise = [&](std::vector<float>& x, int64_t R) {
    for (int64_t r = 0; r < R; ++r) {
      double s = 0; for (int64_t k = 0; k < K; ++k) s += (double)x[r * K + k] * x[r * K + k];
      const float inv = (float)(1.0 / std::sqrt(s));
      for (int64_t k = 0; k < K; ++k) x[r * K + k] *= inv;
    }
  };
  normalise(ha, M); normalise(hb, N);
  float *dA, *dB, *out0, *out1; uint16_t *sA, *sB;
  CK(hipMalloc(&dA, M * K * 4)); CK(hipMalloc(&dB, N * K * 4));
  CK(hipMalloc(&sA, split_elems(M, K) * 2)); CK(hipMalloc(&sB, split_elems(N, K) * 2));
  CK(hipMalloc(&out0, M * N * 4)); CK(hipMalloc(&out1, M * N * 4));
  CK(hipMemcpy(dA, ha.data(), M * K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hb.data(), N * K * 4, hipMemcpyHostToDevice));
  launch_split(dA, nullptr, M, K, sA, nullptr); launch_split(dB, nullptr, N, K, sB, nullptr);
  CK(hipDeviceSynchronize());

  const int64_t tm3 = (M + 255) / 256, tn = (N + 127) / 128, tn5 = (N + 255) / 256;
  struct Variant { const char* name; int id; };
  std::vector<Variant> vs = {{"bf16x3 dma256 (k2)", 0}, {"bf16x3 pingpong (k3)", 1}, {"bf16x3 8phase", 2}, {"bf16x3 8phase no-dma", 3},
                             {"f32 128x128", 4}, {"f32 8phase", 5}};
  const size_t n_checked = 3;  // bf16x3 variants compared bit for bit against kernel 2 (f32: compared with each other below)
#ifdef SL_GEMM_CLOCKPROBE
  unsigned long long* probe; CK(hipMalloc(&probe, 8 * 4 * 131072)); CK(hipMemset(probe, 0, 8 * 4 * 131072));
#endif
  const int64_t tm1 = (M + 127) / 128, tn1 = (N + 127) / 128;
  const int64_t grid8 = tm3 * tn5;
  auto launch = [&](int id, float* out) {
#ifdef SL_GEMM_CLOCKPROBE
    PlainEpi epi{out, N, probe};
#else
    PlainEpi epi{out, N};
#endif
    const unsigned char *bA = (const unsigned char*)sA, *bB = (const unsigned char*)sB;
    switch (id) {
      case 0: hipLaunchKernelGGL((gemm3_nt_dma256_kernel<PlainEpi>), dim3((unsigned)(tm3 * tn)), dim3(256), 0, nullptr, sA, sB, M, N, Kp, (int)tn, epi); break;
      case 1: hipLaunchKernelGGL((gemm3_nt_pingpong_kernel<PlainEpi>), dim3((unsigned)(tm3 * tn5)), dim3(512), 0, nullptr, sA, sB, M, N, Kp, (int)tn5, epi); break;
      case 2: hipLaunchKernelGGL((sl::gemm8::gemm_nt_8phase_kernel<0, PlainEpi>), dim3((unsigned)grid8), dim3(512), 0, nullptr, bA, bB, M, N, 4 * Kp, (int)(Kp / 32), (int)tm3, (int)tn5, epi); break;
      case 3: hipLaunchKernelGGL((sl::gemm8::gemm_nt_8phase_kernel<0, PlainEpi, true>), dim3((unsigned)grid8), dim3(512), 0, nullptr, bA, bB, M, N, 4 * Kp, (int)(Kp / 32), (int)tm3, (int)tn5, epi); break;
      case 4: hipLaunchKernelGGL((sl::gemm::gemm_nt_kernel<true, PlainEpi>), dim3((unsigned)(tm1 * tn1)), dim3(256), 0, nullptr, dA, dB, M, N, K, (int)tn1, epi); break;
      case 5: hipLaunchKernelGGL((sl::gemm8::gemm_nt_8phase_kernel<1, PlainEpi>), dim3((unsigned)grid8), dim3(512), 0, nullptr, (const unsigned char*)dA, (const unsigned char*)dB, M, N, K * 4, (int)(K / 32), (int)tm3, (int)tn5, epi); break;
    }
    CK(hipGetLastError());
  };
  // ---- bit identity vs kernel 2 -------------------------------------------------------------------------------------
  CK(hipMemset(out0, 0xff, M * N * 4));
  launch(0, out0);
  CK(hipDeviceSynchronize());
  std::vector<float> h0((size_t)M * N), h1((size_t)M * N);
  CK(hipMemcpy(h0.data(), out0, M * N * 4, hipMemcpyDeviceToHost));
  // fp64 spot check of kernel 2 itself
  double maxerr = 0;
  for (int t = 0; t < 2000; ++t) {
    const int64_t r = (int64_t)(rng() % M), c = (int64_t)(rng() % N);
    double s = 0; for (int64_t k = 0; k < K; ++k) s += (double)ha[r * K + k] * hb[c * K + k];
    maxerr = std::max(maxerr, std::fabs(s - (double)h0[r * N + c]));
  }
  printf("kernel 2 vs fp64 (2000 samples): max |err| %.3e\n", maxerr);
  for (size_t v = 1; v < n_checked; ++v) {
    for (int rep = 0; rep < 3; ++rep) {  // repeated: a race shows up as a run-to-run difference
      CK(hipMemset(out1, 0xff, M * N * 4));
      launch(vs[v].id, out1);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h1.data(), out1, M * N * 4, hipMemcpyDeviceToHost));
      size_t bad = 0; size_t first = 0;
      for (size_t i = 0; i < h0.size(); ++i)
        if (memcmp(&h0[i], &h1[i], 4) != 0) { if (!bad) first = i; ++bad; }
      printf("%-22s run %d: %zu / %zu elements differ from kernel 2", vs[v].name, rep, bad, h0.size());
      if (bad) printf("  (first at row %zu col %zu: %.9g vs %.9g)", first / N, first % N, h1[first], h0[first]);
      printf("\n");
    }
  }
  if (K % 32 == 0) {
    launch(4, out0);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h0.data(), out0, M * N * 4, hipMemcpyDeviceToHost));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(out1, 0xff, M * N * 4));
      launch(5, out1);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h1.data(), out1, M * N * 4, hipMemcpyDeviceToHost));
      size_t bad = 0;
      for (size_t i = 0; i < h0.size(); ++i) bad += memcmp(&h0[i], &h1[i], 4) != 0;
      printf("f32 8phase             run %d: %zu / %zu elements differ from the f32 128x128 kernel\n", rep, bad, h0.size());
    }
  }
  // ---- timing: interleaved rounds --------------------------------------------------------------------------------------
  const int reps = 20;
  std::vector<std::vector<double>> tf(vs.size());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (size_t v = 0; v < vs.size(); ++v) { launch(vs[v].id, out1); }
  CK(hipDeviceSynchronize());
  for (int round = 0; round < rounds; ++round)
    for (size_t v = 0; v < vs.size(); ++v) {
      CK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < reps; ++i) launch(vs[v].id, out1);
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      tf[v].push_back(2.0 * M * N * K * reps / (ms * 1e-3) / 1e12);
#ifdef SL_GEMM_CLOCKPROBE
      if (round == rounds - 1) {  // per-workgroup shader cycles (s_memtime) against the 100 MHz constant clock of the last launch
        if (vs[v].id == 4) continue;
        const bool k8 = vs[v].id == 2 || vs[v].id == 3 || vs[v].id == 5;
        const int64_t nblk = vs[v].id == 0 ? tm3 * tn : (k8 ? grid8 : tm3 * tn5);
        const int64_t ntile = vs[v].id == 0 ? tm3 * tn : tm3 * tn5;
        std::vector<unsigned long long> hs(2 * nblk); CK(hipMemcpy(hs.data(), probe, 16 * nblk, hipMemcpyDeviceToHost));
        double cyc = 0, rt = 0, cmax = 0; for (int64_t b = 0; b < nblk; ++b) { cyc += (double)hs[2 * b]; rt += (double)hs[2 * b + 1]; cmax = std::max(cmax, (double)hs[2 * b]); }
        // matrix-pipe cycles per SIMD one tile needs (a 256 x 128 tile of kernel 2 shares its CU with a second workgroup)
        const double mfma_cycles = (double)(Kp / 32) * 8 * (vs[v].id == 5 ? 2048 : 384) * (vs[v].id == 0 ? 0.25 : 1.0);
        printf("%-22s %lld workgroups, %lld tiles: %.0f shader cycles per tile (longest workgroup %.0f), clock %.0f MHz, in-tile matrix-pipe duty %.3f\n",
               vs[v].name, (long long)nblk, (long long)ntile, cyc / ntile, cmax, cyc / rt * 100.0, mfma_cycles * ntile / cyc);
        if (k8) {
          std::vector<unsigned long long> hp(2 * nblk); CK(hipMemcpy(hp.data(), probe + 131072, 16 * nblk, hipMemcpyDeviceToHost));
          double pro = 0, loop = 0; for (int64_t b = 0; b < nblk; ++b) { pro += (double)hp[2 * b]; loop += (double)hp[2 * b + 1]; }
          printf("%-22s   first tile of a workgroup: prologue %.0f cycles, k loop %.0f (%.1f per slot; %d = matrix pipe never idle)\n", vs[v].name,
                 pro / nblk, loop / nblk, loop / nblk / ((double)(Kp / 32) * 8), vs[v].id == 5 ? 2048 : 384);
        }
      }
#endif
    }
  printf("M=%lld N=%lld K=%lld, %d rounds x %d launches, normalised random operands\n", (long long)M, (long long)N, (long long)K, rounds, reps);
  for (size_t v = 0; v < vs.size(); ++v) {
    std::sort(tf[v].begin(), tf[v].end());
    const bool f32 = vs[v].id >= 4;
    printf("%-22s TFLOP/s algorithmic: median %.1f  min %.1f  max %.1f   (frac of %s: %.3f)\n", vs[v].name, tf[v][tf[v].size() / 2],
           tf[v].front(), tf[v].back(), f32 ? "157.3" : "833.3", tf[v][tf[v].size() / 2] / (f32 ? 157.3 : 833.3));
  }
  return 0;
}
