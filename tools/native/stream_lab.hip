// What can a read-once stream reach on this MI355X, and by which load form?  (K1's roofline question.)
// Every variant reads a buffer once and keeps a running max per lane (result written at the end, so nothing is dead):
//   vgpr<AUX>        buffer_load_dwordx4 to VGPRs, 8 in flight per wave, 32 waves per CU (the form reduce.hip uses)
//   dma_private<AUX> each wave owns an LDS ring (SLOTS x 1 KiB per wave-instruction group) filled by LDS-DMA
//                    (global_load_lds_dwordx4) and consumed by ds_read_b128 after a counted vmcnt
// Buffers are rotated through > 1.5 GB so the 256 MiB Infinity Cache cannot serve them.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 stream_lab.hip -o stream_lab
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// ---- VGPR loads ------------------------------------------------------------------------------------------------------
template <int AUX, int U>
__global__ __launch_bounds__(256) void vgpr_kernel(const float* __restrict__ x, int64_t nbytes, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int64_t batch_bytes = (int64_t)U * 1024;
  const int64_t nbatch = nbytes / batch_bytes;
  float m = -1e30f;
  for (int64_t b = wave; b < nbatch; b += nwaves) {
    const uint64_t bptr = (uint64_t)(reinterpret_cast<const char*>(x) + b * batch_bytes);
    const uint32_t blo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bptr);
    const uint32_t bhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bptr >> 32));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)bhi << 32) | blo), 0, (int)batch_bytes, 0x00020000);
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, u * 1024, AUX);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const f32x4 f = __builtin_bit_cast(f32x4, v[u]);
      m = max3(max3(m, f[0], f[1]), f[2], f[3]);
    }
  }
  if (m == 12345.f) out[wave * 64 + lane] = m;  // never true for the test data; keeps the loads alive
}

// ---- LDS-DMA into a wave-private ring --------------------------------------------------------------------------------
// A wave's ring holds DEPTH groups of GRP KiB; group g of the wave's stream is DMA'd GRP x 1 KiB instructions at a time.
// SHIFT: every group starts SHIFT bytes past a 1-KiB boundary (a 1-KiB instruction then touches 9 lines of 128 B, 2 partly)
template <int AUX, int GRP, int DEPTH, int WAVES, int SHIFT = 0>
__global__ __launch_bounds__(WAVES * 64) void dma_private_kernel(const float* __restrict__ x, int64_t nbytes, float* __restrict__ out) {
  extern __shared__ __align__(1024) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t wave = (int64_t)blockIdx.x * WAVES + wib;
  const int64_t nwaves = (int64_t)gridDim.x * WAVES;
  constexpr int64_t group_bytes = (int64_t)GRP * 1024;
  const int64_t ngroup = nbytes / group_bytes - (SHIFT ? 1 : 0);
  unsigned char* ring = smem + wib * (DEPTH * GRP * 1024);
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const char* base = reinterpret_cast<const char*>(x);
  auto issue = [&](int64_t gidx, int slot) __attribute__((always_inline)) {
    const char* src = base + gidx * group_bytes + lane * 16 + SHIFT;
#pragma unroll
    for (int i = 0; i < GRP; ++i)
      __builtin_amdgcn_global_load_lds((glb_void*)(src + i * 1024), (lds_void*)(ring + (slot * GRP + i) * 1024), 16, 0, AUX);
  };
  float m = -1e30f;
  // groups of this wave: wave, wave + nwaves, ...
  int64_t nmine = ngroup > wave ? (ngroup - wave + nwaves - 1) / nwaves : 0;
  for (int d = 0; d < DEPTH - 1 && d < nmine; ++d) issue(wave + d * nwaves, d);
  for (int64_t i = 0; i < nmine; ++i) {
    const int slot = (int)(i % DEPTH);
    if (i + DEPTH - 1 < nmine) {
      issue(wave + (i + DEPTH - 1) * nwaves, (int)((i + DEPTH - 1) % DEPTH));
      // group i done when at most (DEPTH - 1) * GRP younger loads are in flight
      if constexpr ((DEPTH - 1) * GRP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if constexpr ((DEPTH - 1) * GRP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr ((DEPTH - 1) * GRP == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if constexpr ((DEPTH - 1) * GRP == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if constexpr ((DEPTH - 1) * GRP == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else if constexpr ((DEPTH - 1) * GRP == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else if constexpr ((DEPTH - 1) * GRP == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int k = 0; k < GRP; ++k) {
      const f32x4 f = *reinterpret_cast<const f32x4*>(ring + (slot * GRP + k) * 1024 + lane * 16);
      m = max3(max3(m, f[0], f[1]), f[2], f[3]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot is refilled by the next iteration's issue
  }
  if (m == 12345.f) out[wave * 64 + lane] = m;
}

// groups of GB bytes (GB <= GRP KiB, multiple of 16): the last instruction of a group is partly masked, as in K1 when a batch of
// rows is not a multiple of 1 KiB
template <int AUX, int GRP, int DEPTH, int WAVES, int GB>
__global__ __launch_bounds__(WAVES * 64) void dma_masked_kernel(const float* __restrict__ x, int64_t nbytes, float* __restrict__ out) {
  extern __shared__ __align__(1024) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t wave = (int64_t)blockIdx.x * WAVES + wib;
  const int64_t nwaves = (int64_t)gridDim.x * WAVES;
  const int64_t ngroup = nbytes / GB;
  unsigned char* ring = smem + wib * (DEPTH * GRP * 1024);
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const char* base = reinterpret_cast<const char*>(x);
  auto issue = [&](int64_t gidx, int slot) __attribute__((always_inline)) {
    const char* src = base + gidx * GB + lane * 16;
#pragma unroll
    for (int i = 0; i < GRP; ++i)
      if (i * 1024 + lane * 16 < GB)
        __builtin_amdgcn_global_load_lds((glb_void*)(src + i * 1024), (lds_void*)(ring + (slot * GRP + i) * 1024), 16, 0, AUX);
  };
  float m = -1e30f;
  int64_t nmine = ngroup > wave ? (ngroup - wave + nwaves - 1) / nwaves : 0;
  for (int d = 0; d < DEPTH - 1 && d < nmine; ++d) issue(wave + d * nwaves, d);
  for (int64_t i = 0; i < nmine; ++i) {
    const int slot = (int)(i % DEPTH);
    if (i + DEPTH - 1 < nmine) {
      issue(wave + (i + DEPTH - 1) * nwaves, (int)((i + DEPTH - 1) % DEPTH));
      if constexpr ((DEPTH - 1) * GRP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if constexpr ((DEPTH - 1) * GRP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr ((DEPTH - 1) * GRP == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int k = 0; k < GRP; ++k) {
      const int byte = k * 1024 + lane * 16;
      const f32x4 f = *reinterpret_cast<const f32x4*>(ring + slot * GRP * 1024 + (byte < GB ? byte : 0));
      m = max3(max3(m, f[0], f[1]), f[2], f[3]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (m == 12345.f) out[wave * 64 + lane] = m;
}

template <class F>
double time_gbps(F launch, int ncopy, int64_t nbytes) {
  for (int i = 0; i < ncopy; ++i) launch(i);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 6 * ncopy;
  CK(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters; ++i) launch(i % ncopy);
  CK(hipEventRecord(e1, nullptr));
  CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  return (double)nbytes * iters / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
  const int64_t nbytes = argc > 1 ? atoll(argv[1]) : (int64_t)256 * 512 * 784 * 4;  // layer2 of ResNet-50 at B = 256: 411 MB
  const int ncopy = (int)std::max<int64_t>(2, (int64_t)1600 * 1024 * 1024 / nbytes + 1);
  std::vector<float*> bufs(ncopy);
  std::vector<float> h(nbytes / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f;
  for (auto& b : bufs) { CK(hipMalloc(&b, nbytes)); CK(hipMemcpy(b, h.data(), nbytes, hipMemcpyHostToDevice)); }
  float* out; CK(hipMalloc(&out, 64 << 20));
  int cus = 256; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  printf("%lld bytes x %d rotating copies, %d CUs\n", (long long)nbytes, ncopy, cus);
  struct R { const char* name; double g; };
  std::vector<R> res;
  for (int round = 0; round < 3; ++round) {
    res.clear();
#define RUN(name, expr) res.push_back({name, time_gbps([&](int i) { expr; }, ncopy, nbytes)})
#define VG(AUX, U, BLK) hipLaunchKernelGGL((vgpr_kernel<AUX, U>), dim3(cus * BLK), dim3(256), 0, nullptr, bufs[i], nbytes, out)
    RUN("vgpr nt   U=8  8 blk/CU", VG(2, 8, 8));
    RUN("vgpr def  U=8  8 blk/CU", VG(0, 8, 8));
    RUN("vgpr nt   U=16 4 blk/CU", VG(2, 16, 4));
    RUN("vgpr nt   U=4  8 blk/CU", VG(2, 4, 8));
#define DMA(AUX, GRP, DEPTH, WAVES, BLK)                                                                                              \
    do {                                                                                                                              \
      CK(hipFuncSetAttribute((const void*)dma_private_kernel<AUX, GRP, DEPTH, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                             WAVES * DEPTH * GRP * 1024));                                                                            \
      hipLaunchKernelGGL((dma_private_kernel<AUX, GRP, DEPTH, WAVES>), dim3(cus * BLK), dim3(WAVES * 64), WAVES * DEPTH * GRP * 1024, \
                         nullptr, bufs[i], nbytes, out);                                                                              \
      CK(hipGetLastError());                                                                                                          \
    } while (0)
    RUN("dma nt  4KBx4 4w 2blk (128K)", DMA(2, 4, 4, 4, 2));
    RUN("dma def 4KBx4 4w 2blk (128K)", DMA(0, 4, 4, 4, 2));
    RUN("dma nt  8KBx4 4w 1blk (128K)", DMA(2, 8, 4, 4, 1));
    RUN("dma nt  4KBx5 8w 1blk (160K)", DMA(2, 4, 5, 8, 1));
    RUN("dma nt  4KBx3 4w 3blk (144K)", DMA(2, 4, 3, 4, 3));
    RUN("dma nt  8KBx3 4w 1blk ( 96K)", DMA(2, 8, 3, 4, 1));
    RUN("dma nt 16KBx2 4w 1blk (128K)", DMA(2, 16, 2, 4, 1));
    RUN("dma nt  4KBx2 4w 4blk (128K)", DMA(2, 4, 2, 4, 4));
#define DMAS(AUX, GRP, DEPTH, WAVES, BLK, SH)                                                                                         \
    do {                                                                                                                              \
      CK(hipFuncSetAttribute((const void*)dma_private_kernel<AUX, GRP, DEPTH, WAVES, SH>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                             WAVES * DEPTH * GRP * 1024));                                                                            \
      hipLaunchKernelGGL((dma_private_kernel<AUX, GRP, DEPTH, WAVES, SH>), dim3(cus * BLK), dim3(WAVES * 64),                         \
                         WAVES * DEPTH * GRP * 1024, nullptr, bufs[i], nbytes, out);                                                  \
      CK(hipGetLastError());                                                                                                          \
    } while (0)
#define DMAM(AUX, GRP, DEPTH, WAVES, BLK, GB)                                                                                         \
    do {                                                                                                                              \
      CK(hipFuncSetAttribute((const void*)dma_masked_kernel<AUX, GRP, DEPTH, WAVES, GB>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                             WAVES * DEPTH * GRP * 1024));                                                                            \
      hipLaunchKernelGGL((dma_masked_kernel<AUX, GRP, DEPTH, WAVES, GB>), dim3(cus * BLK), dim3(WAVES * 64),                          \
                         WAVES * DEPTH * GRP * 1024, nullptr, bufs[i], nbytes, out);                                                  \
      CK(hipGetLastError());                                                                                                          \
    } while (0)
    RUN("dma nt 3136Bx3 4w 3blk masked", DMAM(2, 4, 3, 4, 3, 3136));
    RUN("dma nt 3920Bx3 4w 3blk masked", DMAM(2, 4, 3, 4, 3, 3920));
    RUN("dma nt 3136Bx2 4w 4blk masked", DMAM(2, 4, 2, 4, 4, 3136));
    RUN("dma nt 6272Bx2 4w 3blk masked", DMAM(2, 7, 2, 4, 3, 6272));
    RUN("dma nt 4096Bx3 4w 3blk masked", DMAM(2, 4, 3, 4, 3, 4096));
    RUN("dma nt  4KBx3 3blk +64B shift", DMAS(2, 4, 3, 4, 3, 64));
    RUN("dma nt  4KBx3 3blk +16B shift", DMAS(2, 4, 3, 4, 3, 16));
    RUN("dma def 4KBx3 3blk +64B shift", DMAS(0, 4, 3, 4, 3, 64));
    RUN("dma nt  4KBx3 3blk +128B shift", DMAS(2, 4, 3, 4, 3, 128));
  }
  for (auto& r : res) printf("%-32s %8.1f GB/s  (%.3f of 8 TB/s)\n", r.name, r.g, r.g / 8000.0);
  return 0;
}
