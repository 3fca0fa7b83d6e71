// Cold-input bandwidth of K1 (sl_reduce_conv) without Python: inputs rotated through > 1.5 GB.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../semanticlens_amd/csrc -I../../include [-DSL_REDUCE_LAB=n] reduce_lab.hip ../../semanticlens_amd/csrc/runtime.hip -o reduce_lab
#include "../../semanticlens_amd/csrc/reduce.hip"

#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
  struct Shape { int64_t B, C, S; };
  const Shape shapes[] = {{256, 512, 784}, {256, 1024, 196}, {256, 2048, 49}, {256, 192, 3136}, {256, 1536, 49}};
  const int64_t maxbytes = 256ll * 512 * 784 * 4;
  const int ncopy = 4;
  std::vector<float*> bufs(ncopy);
  std::vector<float> h(maxbytes / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(((i * 2654435761u) >> 8) & 0xffff) / 65536.f - 0.25f;
  for (auto& b : bufs) { CK(hipMalloc(&b, maxbytes)); CK(hipMemcpy(b, h.data(), maxbytes, hipMemcpyHostToDevice)); }
  uint16_t* cand; CK(hipMalloc(&cand, 64 << 20));
  for (const Shape& s : shapes) {
    const int64_t bytes = s.B * s.C * s.S * 4;
    // rotate over disjoint windows of the four 411-MB buffers: consecutive launches never touch the same bytes within
    // 1.6 GB of traffic
    const int per = (int)(maxbytes / bytes);
    std::vector<const float*> views;
    for (int c = 0; c < ncopy; ++c)
      for (int k = 0; k < per && (int)views.size() < 64; ++k) views.push_back(bufs[c] + k * (bytes / 4));
    auto run = [&](int i) {
      int rc = sl_reduce_conv(views[i % views.size()], SL_F32, s.B, s.C, s.S, s.C * s.S, s.S, 1, SL_CONV_MAX, cand, nullptr, nullptr);
      if (rc) { fprintf(stderr, "sl_reduce_conv: %s\n", sl_last_error()); exit(1); }
    };
    for (int i = 0; i < (int)views.size(); ++i) run(i);
    CK(hipDeviceSynchronize());
    std::vector<double> g;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int round = 0; round < 5; ++round) {
      const int iters = 2 * (int)views.size();
      CK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < iters; ++i) run(i);
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      g.push_back((double)bytes * iters / (ms * 1e-3) / 1e9);
    }
    std::sort(g.begin(), g.end());
    printf("B=%lld C=%lld S=%lld (%.0f MB): median %.0f GB/s (%.3f of 8 TB/s), min %.0f max %.0f, %.1f us\n", (long long)s.B, (long long)s.C,
           (long long)s.S, bytes / 1e6, g[2], g[2] / 8000, g.front(), g.back(), bytes / g[2] / 1e3);
  }
  return 0;
}
