// Which CUs does bit i of a hipExtStreamCreateWithCUMask mask enable?  For each probed mask: launch 4096 one-wave
// workgroups that spin briefly and record (XCC_ID, HW_ID); print the set of (xcc, se, cu) seen.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void where_kernel(uint32_t* out) {
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  uint64_t t0 = clock64();
  while (clock64() - t0 < 20000) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

__global__ void read_kernel(const float4* __restrict__ x, size_t n, float* out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = x[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) out[0] = acc;
}

// a 1 GB read stream under a mask: GB/s (best of 5)
static int bandwidth(const std::vector<int>& bits, const char* label, const float4* d_x, size_t n, float* d_o) {
  hipStream_t st;
  if (bits.empty()) {
    CHECK(hipStreamCreate(&st));
  } else {
    uint32_t words[8] = {0};
    for (int b : bits) words[b >> 5] |= 1u << (b & 31);
    CHECK(hipExtStreamCreateWithCUMask(&st, 8, words));
  }
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  float best = 1e30f;
  for (int r = 0; r < 6; ++r) {
    CHECK(hipEventRecord(a, st));
    hipLaunchKernelGGL(read_kernel, dim3(256 * 8), dim3(256), 0, st, d_x, n, d_o);
    CHECK(hipEventRecord(b, st));
    CHECK(hipStreamSynchronize(st));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    if (r && ms < best) best = ms;
  }
  printf("%-28s read stream %7.0f GB/s\n", label, n * 16.0 / best / 1e6);
  CHECK(hipStreamDestroy(st));
  return 0;
}

static int probe(const std::vector<int>& bits, const char* label, uint32_t* d_out) {
  uint32_t words[8] = {0};
  for (int b : bits) words[b >> 5] |= 1u << (b & 31);
  hipStream_t st;
  CHECK(hipExtStreamCreateWithCUMask(&st, 8, words));
  const int nb = 4096;
  hipLaunchKernelGGL(where_kernel, dim3(nb), dim3(64), 0, st, d_out);
  CHECK(hipStreamSynchronize(st));
  std::vector<uint32_t> h(2 * nb);
  CHECK(hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost));
  std::set<uint32_t> cus;
  int per_xcc[16] = {0};
  for (int i = 0; i < nb; ++i) {
    const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 15;
    const uint32_t cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    cus.insert((xcc << 12) | (se << 8) | (sh << 4) | cu);
  }
  for (uint32_t c : cus) per_xcc[c >> 12]++;
  printf("%-28s -> %3zu CUs; per XCC:", label, cus.size());
  for (int x = 0; x < 8; ++x) printf(" %2d", per_xcc[x]);
  if (cus.size() <= 8) {
    printf("  [");
    for (uint32_t c : cus) printf(" x%u.se%u.sh%u.cu%u", c >> 12, (c >> 8) & 15, (c >> 4) & 15, c & 15);
    printf(" ]");
  }
  printf("\n");
  CHECK(hipStreamDestroy(st));
  return 0;
}

int main() {
  uint32_t* d_out;
  CHECK(hipMalloc(&d_out, 2 * 4096 * 4));
  std::vector<int> all;
  for (int i = 0; i < 256; ++i) all.push_back(i);
  probe(all, "all 256 bits", d_out);
  for (int i : {0, 1, 2, 7, 8, 9, 31, 32, 33, 63, 64, 128, 255}) {
    char l[64];
    snprintf(l, sizeof l, "bit %d", i);
    probe({i}, l, d_out);
  }
  auto range = [](int a, int b) { std::vector<int> v; for (int i = a; i < b; ++i) v.push_back(i); return v; };
  probe(range(0, 32), "bits 0..31", d_out);
  probe(range(0, 128), "bits 0..127", d_out);
  probe(range(0, 224), "bits 0..223", d_out);
  probe(range(224, 256), "bits 224..255", d_out);
  probe(range(208, 256), "bits 208..255", d_out);
  std::vector<int> ev;
  for (int i = 7; i < 256; i += 8) ev.push_back(i);
  probe(ev, "bits 7,15,..,255", d_out);
  std::vector<int> rest;
  for (int i = 0; i < 256; ++i) if (i % 8 != 7) rest.push_back(i);
  probe(rest, "all but 7,15,..", d_out);
  {
    const size_t n = (size_t)1 << 26;  // float4s = 1 GiB
    float4* d_x;
    float* d_o;
    CHECK(hipMalloc(&d_x, n * 16));
    CHECK(hipMalloc(&d_o, 4));
    CHECK(hipMemset(d_x, 0, n * 16));
    bandwidth({}, "plain stream", d_x, n, d_o);
    bandwidth(all, "mask: all 256 bits", d_x, n, d_o);
    bandwidth(range(0, 224), "mask: bits 0..223", d_x, n, d_o);
    bandwidth(range(0, 192), "mask: bits 0..191", d_x, n, d_o);
    bandwidth(range(0, 128), "mask: bits 0..127", d_x, n, d_o);
    bandwidth(range(224, 256), "mask: bits 224..255", d_x, n, d_o);
  }
  return 0;
}
