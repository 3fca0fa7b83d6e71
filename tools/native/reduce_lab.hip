// K1 (sl_reduce_conv) without Python, in the two regimes the bench reports:
//   reduce_lab            cold inputs: rotated through > 1.5 GB, back-to-back launches timed with events
//   reduce_lab pipe       in-pipeline: an in-place ReLU over the input (what torchvision's Bottleneck ends with) runs on
//                         the same stream right before every reduce; the reduce's own dispatch time comes from sl_prof;
//                         sweeps sl_set_reduce_policy(nt_min_bytes, tail_bytes)
//   reduce_lab pipe1      the same with the SHIPPED policy only (48 launches per shape): the run to put under rocprofv3
// Build: tools/native/build_reduce_lab.sh [-DSL_REDUCE_LAB=n] [-DSL_REDUCE_LAB_HEAD_AUX=a -DSL_REDUCE_LAB_TAIL_AUX=b -DSL_REDUCE_LAB_TAIL_FIRST=1]
#include "../../semanticlens_amd/csrc/reduce.hip"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void relu_inplace_kernel(float4* x, int64_t n4) {
  // torch's vectorised elementwise kernel: a block owns a contiguous span, 4 x float4 per thread
  const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t i = base + j * 256;
    if (i < n4) {
      float4 v = x[i];
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      x[i] = v;
    }
  }
}

struct Shape { int64_t B, C, S; };

__global__ __launch_bounds__(256) void null_kernel(float* p) {
  extern __shared__ unsigned char sm[];
  if (p && threadIdx.x == 999) p[0] = sm[0];
}

// fixed cost of a launch: dispatch duration (start/stop events of hipExtLaunchKernelGGL, what sl_prof and rocprofv3 see)
// of an empty kernel with K1's grid, and K1 over growing slices of one shape -> intercept of t(bytes)
static void fixed_mode(const std::vector<float*>& bufs, uint16_t* cand) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void*)null_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10));
  for (int blocks : {256, 1024, 1536, 2048}) {
    for (int lds : {0, 26112}) {
      double tot = 0;
      for (int i = 0; i < 20; ++i) {
        hipExtLaunchKernelGGL(null_kernel, dim3(blocks), dim3(256), lds, nullptr, e0, e1, 0, (float*)nullptr);
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 4) tot += ms;
      }
      printf("null kernel %4d blocks x 256 threads, %5d B LDS: %.2f us\n", blocks, lds, tot / 16 * 1e3);
    }
  }
  sl_set_reduce_policy(0, 0);
  for (const Shape& base : {Shape{256, 2048, 49}, Shape{256, 1024, 196}, Shape{256, 512, 784}}) {
    for (int64_t B : {32, 64, 128, 256}) {
      const int64_t bytes = B * base.C * base.S * 4;
      sl_prof_enable(1); sl_prof_reset();
      for (int i = 0; i < 16; ++i) {
        const float* x = bufs[i % bufs.size()] + (i / bufs.size() % 1) * (bytes / 4);
        sl_reduce_conv(x, SL_F32, B, base.C, base.S, base.C * base.S, base.S, 1, SL_CONV_MAX, cand, nullptr, nullptr);
      }
      double ms = 0, work = 0; int64_t nl = 0;
      sl_prof_read(SL_PROF_REDUCE, &ms, &nl, &work);
      sl_prof_enable(0);
      printf("S=%4lld B=%3lld %6.1f MB: %6.2f us  %5.0f GB/s\n", (long long)base.S, (long long)B, bytes / 1e6, ms / nl * 1e3, bytes * (double)nl / ms / 1e6);
    }
  }
}

// producer variants behind which K1 runs with the DEFAULT policy: what the ~1.8 us in-pipeline penalty of the two
// cache-resident inputs belongs to
static void producer_mode(const std::vector<float*>& bufs, uint16_t* cand) {
  const Shape shapes[] = {{256, 1024, 196}, {256, 2048, 49}};
  const char* names[] = {"no producer (cold, policy all-nt)", "relu on the SAME buffer (the pipeline)", "relu on ANOTHER buffer (dirty caches, cold input)",
                         "relu on the same buffer, then an empty kernel"};
  for (int variant = 0; variant < 4; ++variant) {
    sl_set_reduce_policy(variant == 0 || variant == 2 ? 0 : -1, variant == 0 || variant == 2 ? 0 : -1);
    printf("%-52s", names[variant]);
    for (const Shape& s : shapes) {
      const int64_t n = s.B * s.C * s.S, bytes = n * 4;
      sl_prof_enable(1);
      sl_prof_reset();
      for (int i = 0; i < 16; ++i) {
        float* x = bufs[i % bufs.size()];
        float* other = bufs[(i + 2) % bufs.size()];
        const dim3 grid((unsigned)((n / 4 + 1023) / 1024));
        if (variant == 1 || variant == 3) hipLaunchKernelGGL(relu_inplace_kernel, grid, dim3(256), 0, nullptr, (float4*)x, n / 4);
        if (variant == 2) hipLaunchKernelGGL(relu_inplace_kernel, grid, dim3(256), 0, nullptr, (float4*)other, n / 4);
        if (variant == 3) hipLaunchKernelGGL(null_kernel, dim3(256), dim3(256), 0, nullptr, (float*)nullptr);
        sl_reduce_conv(x, SL_F32, s.B, s.C, s.S, s.C * s.S, s.S, 1, SL_CONV_MAX, cand, nullptr, nullptr);
      }
      double ms = 0, work = 0; int64_t nl = 0;
      sl_prof_read(SL_PROF_REDUCE, &ms, &nl, &work);
      sl_prof_enable(0);
      printf("  %4.0f MB: %6.2f us %5.0f GB/s", bytes / 1e6, ms / nl * 1e3, bytes * (double)nl / ms / 1e6);
    }
    printf("\n");
  }
  sl_set_reduce_policy(-1, -1);
}

static void pipe_mode(const std::vector<float*>& bufs, uint16_t* cand, bool shipped_only = false) {
  const Shape shapes[] = {{256, 512, 784}, {256, 1024, 196}, {256, 2048, 49}};
  struct Pol { int64_t nt_min, tail; const char* name; };
  const Pol pols[] = {{-1, -1, "default (256 MiB / 240 MiB)"}, {0, 0, "all nt"}, {1ll << 40, 0, "all default-policy"},
                      {256ll << 20, 160ll << 20, "tail 160"}, {256ll << 20, 200ll << 20, "tail 200"},
                      {256ll << 20, 256ll << 20, "tail 256"}, {256ll << 20, 300ll << 20, "tail 300"},
                      {64ll << 20, 240ll << 20, "nt_min 64 MiB (layer3/4 head nt too)"},
                      {64ll << 20, 64ll << 20, "nt_min 64, tail 64"}};
  CK(hipDeviceSynchronize());
  for (const Pol& p : pols) {
    if (shipped_only && &p != &pols[0]) break;  // `pipe1`: the shipped policy only, so that a kernel trace of the run holds one regime
    sl_set_reduce_policy(p.nt_min, p.tail);
    double tot_b = 0, tot_ms = 0;
    printf("policy %-40s", p.name);
    for (const Shape& s : shapes) {
      const int64_t n = s.B * s.C * s.S, bytes = n * 4;
      sl_prof_enable(1);
      sl_prof_reset();
      const int iters = shipped_only ? 48 : 12;
      for (int i = 0; i < iters; ++i) {
        float* x = bufs[i % bufs.size()];
        hipLaunchKernelGGL(relu_inplace_kernel, dim3((unsigned)((n / 4 + 1023) / 1024)), dim3(256), 0, nullptr, (float4*)x, n / 4);
        int rc = sl_reduce_conv(x, SL_F32, s.B, s.C, s.S, s.C * s.S, s.S, 1, SL_CONV_MAX, cand, nullptr, nullptr);
        if (rc) { fprintf(stderr, "sl_reduce_conv: %s\n", sl_last_error()); exit(1); }
      }
      double ms = 0, work = 0; int64_t nl = 0;
      sl_prof_read(SL_PROF_REDUCE, &ms, &nl, &work);
      sl_prof_enable(0);
      printf("  %4.0f MB: %5.0f GB/s %.3f", bytes / 1e6, bytes * (double)nl / ms / 1e6, bytes * (double)nl / ms / 1e6 / 8000);
      tot_b += (double)bytes * nl; tot_ms += ms;
    }
    printf("  | all %5.0f GB/s %.3f\n", tot_b / tot_ms / 1e6, tot_b / tot_ms / 1e6 / 8000);
  }
  sl_set_reduce_policy(-1, -1);
}

int main(int argc, char** argv) {
  const Shape shapes[] = {{256, 512, 784}, {256, 1024, 196}, {256, 2048, 49}, {256, 1536, 49}, {256, 768, 196}};
  const int64_t maxbytes = 256ll * 512 * 784 * 4;
  const int ncopy = 4;
  std::vector<float*> bufs(ncopy);
  std::vector<float> h(maxbytes / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(((i * 2654435761u) >> 8) & 0xffff) / 65536.f - 0.25f;
  for (auto& b : bufs) { CK(hipMalloc(&b, maxbytes)); CK(hipMemcpy(b, h.data(), maxbytes, hipMemcpyHostToDevice)); }
  uint16_t* cand; CK(hipMalloc(&cand, 64 << 20));
  if (argc > 1 && !strcmp(argv[1], "pipe")) { pipe_mode(bufs, cand); return 0; }
  if (argc > 1 && !strcmp(argv[1], "pipe1")) { pipe_mode(bufs, cand, true); return 0; }
  if (argc > 1 && !strcmp(argv[1], "fixed")) { fixed_mode(bufs, cand); return 0; }
  if (argc > 1 && !strcmp(argv[1], "producer")) { producer_mode(bufs, cand); return 0; }
  sl_set_reduce_policy(0, 0);  // cold inputs: every byte with the streaming policy
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
