// LAB ONLY — not part of libsemanticlens_hip.so.  The round-2 ping-pong kernel ("kernel 3" of gemm_bf16x3.hpp), superseded by the
// 8-phase kernel (gemm_8phase.hpp) in round 2 and retired from the product library in round 6; tools/native/gemm3_lab.hip includes it
// after gemm_bf16x3.hpp to keep its A/B history reproducible.
#pragma once
#include "gemm_bf16x3.hpp"

namespace sl {
namespace gemm3 {

// ---- kernel 3: 256 x 256 block tile, two wave groups in ping-pong, four-stage LDS-DMA ring --------------------------
// A tile's LDS-DMA round trip is ~3000-5000 cycles under load while the L2 -> LDS path sustains > 20 TB/s
// (tools/native/feed_probe.hip): by Little's law kernel 2 would need ~124 KB of tile data in flight per CU to keep the
// matrix pipe busy, and its two 48 KB single buffers, in flight half of the time, allow about half of that.  This kernel
// lowers the demand (a 256 x 256 tile needs a third less data per flop) and keeps three of four 32 KB stages in flight:
//   * one 512-thread workgroup per CU = two groups of four waves (one wave of each group per SIMD); group g owns the
//     256 x 128 half of the output at columns 128 g; both groups share the A operand in LDS;
//   * a stage is ONE k-step (16 wide) of all 512 rows: per row 64 bytes [hi16 | lo16] (the two 32-byte halves of the
//     k-step inside the row's 128-byte line), staged by 1-KiB LDS-DMA loads of 16 rows; 16-byte slots XOR-swizzled by
//     (row >> 2) & 3 for conflict-free ds_read_b128;
//   * barrier-separated phases with opposite roles: in phase 2 s group 0 loads the 12 fragments of k-step s into
//     registers while group 1 runs the 24 MFMAs of k-step s - 1; in phase 2 s + 1 group 0 runs its 24 MFMAs while group 1
//     loads — every phase one wave per SIMD feeds the matrix pipe and the other uses the LDS;
//   * k-step s + 3 is issued at the start of phase 2 s + 1 (its stage was last read in phase 2 s - 1) and awaited with a
//     counted s_waitcnt vmcnt(8) at the end of phase 2 s + 3; raw s_barrier (__syncthreads would add vmcnt(0) and drain
//     the ring every phase).
// Same accumulation order per output element as kernels 1 and 2: bit-identical results.
// Measured (10000 x 9216 x 1152, tools/native/clock_probe.hip): 166 K cycles per 256 x 256 workgroup = 67 % matrix-pipe
// duty against 60 % for kernel 2, 457 vs 426 TFLOP/s on all-zero operands — but 355 vs 349 TFLOP/s on random operands,
// because the shader clock drops from 2.0 to 1.85 GHz: at ~1.05 PFLOP/s of issued bf16 MFMA work the part is at its power
// limit on real data.  Selected with SL_G3_TILE=512 only; kernel 2 stays the default for large grids.
constexpr int BM4 = 256, BN4 = 256;
constexpr int IMG4_BYTES = 256 * 64;          // 256 rows x [hi16 | lo16]
constexpr int STAGE4_BYTES = 2 * IMG4_BYTES;  // A rows, then B rows: 32 KB
constexpr int NSTAGE4 = 4;

template <class Epi>
__global__ __launch_bounds__(512, 2) void gemm3_nt_pingpong_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                                                   int64_t M, int64_t N, int64_t Kp, int tiles_n, Epi epi) {
  __shared__ __align__(1024) unsigned char smem[NSTAGE4 * STAGE4_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7
  const int grp = w >> 2, w4 = w & 3;
  const int wm = w4 >> 1, wn = w4 & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int tile = blockIdx.x;
  const int64_t m0 = (int64_t)(tile / tiles_n) * BM4;
  const int64_t n0 = (int64_t)(tile % tiles_n) * BN4;
#ifdef SL_GEMM_CLOCKPROBE  // tools/native/clock_probe.hip
  const unsigned long long probe_c0 = __builtin_amdgcn_s_memtime(), probe_r0 = __builtin_amdgcn_s_memrealtime();
#endif

  floatx16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // DMA: one 1-KiB load covers 16 rows x 64 bytes; wave w stages rows [32 w, 32 w + 32) of A and of B (two loads each).
  // Lane L lands in row L >> 2 of the load, slot L & 3; slot s of a row holds chunk s ^ ((row >> 2) & 3), where chunks
  // 0, 1 are the two 16-byte halves of hi16 and 2, 3 those of lo16.  In the global line ([hi32 | lo32], 128 bytes) the
  // k-step's hi16 starts at byte 32 (s & 1) and its lo16 at 64 + 32 (s & 1).
  int64_t a_src[2], b_src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = w * 32 + i * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    const int eoff = (chunk >> 1) * 32 + (chunk & 1) * 8;  // element offset inside the line for an even k-step
    a_src[i] = (m0 + row < M ? m0 + row : M - 1) * 2 * Kp + eoff;  // rows past the edge are clamped (never stored)
    b_src[i] = (n0 + row < N ? n0 + row : N - 1) * 2 * Kp + eoff;
  }
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  auto dma_step = [&](int s) __attribute__((always_inline)) {  // k-step s -> stage s & 3
    const int64_t koff = (int64_t)(s >> 1) * 64 + (s & 1) * 16;
    unsigned char* l = smem + (s & (NSTAGE4 - 1)) * STAGE4_BYTES + w * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((glb_void*)(A + a_src[i] + koff), (lds_void*)(l + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void*)(B + b_src[i] + koff), (lds_void*)(l + IMG4_BYTES + i * 1024), 16, 0, 0);
    }
  };
  int a_off[4], b_off[2];  // byte offset of this lane's hi fragment inside a stage; the lo fragment is 2 slots further (XOR 2)
  int a_lo[4], b_lo[2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ar = wm * 128 + t * 32 + li, sw = (ar >> 2) & 3;
    a_off[t] = ar * 64 + ((lh ^ sw) << 4);
    a_lo[t] = ar * 64 + (((2 + lh) ^ sw) << 4);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int br = grp * 128 + wn * 64 + t * 32 + li, sw = (br >> 2) & 3;
    b_off[t] = IMG4_BYTES + br * 64 + ((lh ^ sw) << 4);
    b_lo[t] = IMG4_BYTES + br * 64 + (((2 + lh) ^ sw) << 4);
  }
  bf16x8 fah[4], fal[4], fbh[2], fbl[2];
  auto load_phase = [&](int s) __attribute__((always_inline)) {
    const unsigned char* base = smem + (s & (NSTAGE4 - 1)) * STAGE4_BYTES;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      fbh[t] = *reinterpret_cast<const bf16x8*>(base + b_off[t]);
      fbl[t] = *reinterpret_cast<const bf16x8*>(base + b_lo[t]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      fah[t] = *reinterpret_cast<const bf16x8*>(base + a_off[t]);
      fal[t] = *reinterpret_cast<const bf16x8*>(base + a_lo[t]);
    }
  };
  auto mfma_phase = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[i], fbh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbh[j], acc[i][j], 0, 0, 0);
      }
  };

  // Phases p = 0 .. 2 ns (one barrier each).  Group 0: p = 2 s loads k-step s, p = 2 s + 1 computes it, the last phase is
  // empty.  Group 1: p = 0 is empty, p = 2 s + 1 loads k-step s, p = 2 s + 2 computes it.  DMA (all eight waves, four
  // loads each): k-steps 0..2 up front, k-step s + 3 at the start of phase 2 s + 1.  The end of phase 2 s + 1 waits until
  // k-step s + 1 has landed: the loads of k-steps s + 2 and s + 3 (8 per wave) may still be in flight.
  const int ns = (int)(Kp / 16);
  auto phase_barrier = [&]() __attribute__((always_inline)) {
    // pinned for the scheduler: hipcc otherwise lets the register-only MFMAs of a phase trail past the barrier into the
    // group's next (load) phase, where they collide with the other group's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto odd_phase_end = [&](int s) __attribute__((always_inline)) {  // end of phase 2 s + 1: k-step s + 1 must be complete
    __builtin_amdgcn_sched_barrier(0);
    if (s + 3 < ns) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if (s + 2 < ns) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    phase_barrier();
  };
  auto even_phase_end = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    phase_barrier();
  };
  if (ns > 0) {
    dma_step(0);
    if (ns > 1) dma_step(1);
    if (ns > 2) dma_step(2);
    if (ns > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ns > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    phase_barrier();
    if (grp == 0) {
      for (int s = 0; s < ns; ++s) {
        load_phase(s);
        even_phase_end();
        if (s + 3 < ns) dma_step(s + 3);
        mfma_phase();
        odd_phase_end(s);
      }
      even_phase_end();
    } else {
      even_phase_end();
      for (int s = 0; s < ns; ++s) {
        if (s + 3 < ns) dma_step(s + 3);
        load_phase(s);
        odd_phase_end(s);
        mfma_phase();
        even_phase_end();
      }
    }
  }

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      store_mfma_tile<true>(epi, m0 + wm * 128 + i * 32 + 4 * lh, n0 + grp * 128 + wn * 64 + j * 32 + li, acc[i][j], M, N);
#ifdef SL_GEMM_CLOCKPROBE
  if (tid == 0) {
    epi.probe[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - probe_c0;
    epi.probe[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - probe_r0;
  }
#endif
}

}  // namespace gemm3
}  // namespace sl
