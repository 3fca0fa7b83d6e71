// Do 16-byte loads from 4-byte-aligned addresses work on this part (global and raw-buffer flavours), and what do they cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* x, float* out, int shift) {
  const float* p = x + shift + threadIdx.x * 4;
  float4 v;
  asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, 4096, 0x00020000);
  const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((shift + threadIdx.x * 4) * 4), 0, 0);
  out[threadIdx.x * 8 + 0] = v.x; out[threadIdx.x * 8 + 1] = v.y; out[threadIdx.x * 8 + 2] = v.z; out[threadIdx.x * 8 + 3] = v.w;
  out[threadIdx.x * 8 + 4] = __builtin_bit_cast(float, w[0]); out[threadIdx.x * 8 + 5] = __builtin_bit_cast(float, w[1]);
  out[threadIdx.x * 8 + 6] = __builtin_bit_cast(float, w[2]); out[threadIdx.x * 8 + 7] = __builtin_bit_cast(float, w[3]);
}

__global__ void stream(const float* __restrict__ x, size_t n4, int shift, float* out) {
  float acc = 0.f;
  const float* base = x + shift;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v;
    const float* p = base + i * 4;
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) out[0] = acc;
}

// reference read stream: 32 waves per CU, four independent 16-byte streaming loads per lane and trip
typedef float vf4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream4(const vf4* __restrict__ x, size_t n4, float* out) {
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const vf4 a = __builtin_nontemporal_load(x + i), b = __builtin_nontemporal_load(x + i + stride);
    const vf4 c = __builtin_nontemporal_load(x + i + 2 * stride), d = __builtin_nontemporal_load(x + i + 3 * stride);
    acc += (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) + (d.x + d.y + d.z + d.w);
  }
  for (; i < n4; i += stride) {
    const vf4 a = __builtin_nontemporal_load(x + i);
    acc += a.x + a.y + a.z + a.w;
  }
  if (acc == 123.456f) out[0] = acc;
}

int main() {
  float h[1024], *d, *o, ho[64 * 8];
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  CHECK(hipMalloc(&d, 4096)); CHECK(hipMalloc(&o, sizeof ho));
  CHECK(hipMemcpy(d, h, 4096, hipMemcpyHostToDevice));
  for (int shift = 0; shift < 4; ++shift) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, shift);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost));
    int bad_g = 0, bad_b = 0;
    for (int t = 0; t < 64; ++t)
      for (int e = 0; e < 4; ++e) {
        bad_g += ho[t * 8 + e] != (float)(shift + t * 4 + e);
        bad_b += ho[t * 8 + 4 + e] != (float)(shift + t * 4 + e);
      }
    printf("shift %d floats: global_load_dwordx4 %s, raw_buffer_load_b128 %s  (lane 1 got %g %g | %g %g)\n", shift, bad_g ? "WRONG" : "ok",
           bad_b ? "WRONG" : "ok", ho[8], ho[9], ho[12], ho[13]);
  }
  const size_t n4 = (size_t)1 << 26;
  float* big;
  CHECK(hipMalloc(&big, n4 * 16 + 64));
  CHECK(hipMemset(big, 0, n4 * 16 + 64));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int shift = 0; shift < 4; ++shift) {
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      CHECK(hipEventRecord(a, 0));
      hipLaunchKernelGGL(stream, dim3(256 * 16), dim3(256), 0, 0, big, n4, shift, o);
      CHECK(hipEventRecord(b, 0));
      CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, a, b));
      if (r && ms < best) best = ms;
    }
    printf("1 GiB stream of 16-byte loads shifted by %d floats: %.0f GB/s\n", shift, n4 * 16.0 / best / 1e6);
  }
  // the same stream at the byte counts of K1's launches, cold (rotating through the 1 GiB allocation)
  for (int variant : {0, 8, 16, 32})
  for (size_t mb : {51, 103, 205, 411, 1024}) {
    const size_t bytes = mb == 1024 ? n4 * 16 : mb * 1000 * 1000 / 16 * 16, n = bytes / 16;
    const int nrot = (int)(n4 * 16 / bytes) > 0 ? (int)(n4 * 16 / bytes) : 1;
    float tot = 0.f;
    int cnt = 0;
    for (int r = 0; r < 3 * nrot + 2; ++r) {
      const vf4* src = reinterpret_cast<const vf4*>(big) + (size_t)(r % nrot) * n;
      CHECK(hipEventRecord(a, 0));
      if (variant == 0) hipLaunchKernelGGL(stream4, dim3(256 * 8), dim3(256), 0, 0, src, n, o);
      else hipLaunchKernelGGL(stream, dim3(256 * variant), dim3(256), 0, 0, reinterpret_cast<const float*>(src), n, 0, o);
      CHECK(hipEventRecord(b, 0));
      CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, a, b));
      if (r >= 2) { tot += ms; ++cnt; }
    }
    printf("%s x%-2d %4zu MB per launch (mean of %2d): %6.1f us  %.0f GB/s = %.3f of 8 TB/s\n",
           variant ? "one load per lane and trip, blocks per CU" : "four loads 8 MB apart per lane and trip      ", variant ? variant : 8, mb, cnt,
           tot / cnt * 1e3, bytes / (tot / cnt) / 1e6, bytes / (tot / cnt) / 8e9);
  }
  return 0;
}
