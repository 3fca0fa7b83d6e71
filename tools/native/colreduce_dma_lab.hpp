// K2 through wave-private LDS-DMA rings — the variant VERDICT r03 #5 asked for, built and measured in round 4
// (profiles/r04_k2_lab.txt): 5.6-6.1 TB/s cold on (256, 197, 768) fp32 against 6.2-6.3 for colreduce2's register ping-pong,
// and 2.0-4.5 TB/s on half-precision inputs (its rings cap a CU at 16 waves).  Kept out of the shipped library; included by
// semanticlens_amd/csrc/reduce.hip only under -DSL_K2_DMA_LAB (SL_COLREDUCE_IMPL=dma then selects it).  Known defect: the
// (256, 257, 1024) bf16 mean differed from torch at depth 2 / 8 waves in the lab run — one more reason it is not shipped.
#pragma once

// ---- colreduce_dma: the same reduction fed through wave-private LDS-DMA rings (round 4) ----------------------------------
// colreduce_kernel issues eight VGPR loads per lane, reduces them, and only then issues the next eight: the bytes a wave
// has in flight swing between 8 KB and nothing once per round, and a (256, 197, 768) launch is only six such rounds long.
// Here the global side is K1's (rowreduce_dma_kernel): a wave owns a 64-piece column chunk (a piece = 16 bytes = 4 fp32 or 8
// half-precision components) and walks its rows in *batches* of kColRows rows; a batch is kColRows 1-KiB LDS-DMA
// instructions (`global_load_lds_dwordx4`: lane l fetches its own 16 bytes of one row) into a slot of a wave-private
// ring of DEPTH slots, DEPTH - 1 batches in flight while one is reduced, counted `s_waitcnt vmcnt`, no barrier until the
// waves of the workgroup (which split the reduced axis) combine.  The per-row work needs no cross-lane step: a lane reads
// its own piece of each row back with one ds_read_b128 (conflict-free: lane * 16 within a 1-KiB row).
// Always kColRows instructions per batch so that the waits are compile-time constants; a row past the end keeps lane 0
// alive on the tensor's first piece and lands in the workgroup's spare KiB (K1's rule).
// Needs: x 16-byte aligned, row and sample strides multiples of 16 bytes, F a multiple of the piece.
constexpr int kColRows = 4;
constexpr int kColDepthDefault = 2;  // ring slots per wave unless SL_COLREDUCE_DEPTH says otherwise (tools/k2_lab.py)

template <typename E, int OP, int NW, int DEPTH>
__global__ __launch_bounds__(64 * NW) void colreduce_dma_kernel(const E* __restrict__ x, int64_t B, int T, int64_t F, int64_t sb,
                                                                 int64_t st, int t_begin, int t_end, float denom,
                                                                 int64_t tail_from, uint16_t* __restrict__ cand,
                                                                 float* __restrict__ outf) {
  constexpr int EPP = 16 / (int)sizeof(E);  // components per piece
  constexpr int CW = 64 * EPP;              // components per chunk
  constexpr bool SUM = (OP == OP_SUM || OP == OP_ABSSUM);
  constexpr int kSlot = kColRows * 1024;
  extern __shared__ __align__(1024) unsigned char smem[];  // NW rings of DEPTH slots + 1 spare KiB
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  unsigned char* ring = smem + w * (DEPTH * kSlot);
  unsigned char* spare = smem + NW * DEPTH * kSlot;
  const int64_t nchunk = (F + CW - 1) / CW;
  const int64_t ntask = B * nchunk;
  const int64_t rot = (tail_from > 0 && tail_from < ntask) ? tail_from : 0;
  const int rows = t_end - t_begin;
  const int nb_total = (rows + kColRows - 1) / kColRows;
  const int nbw = w < nb_total ? (nb_total - w + NW - 1) / NW : 0;  // batches of this wave: w, w + NW, ...
  const unsigned char* x0 = reinterpret_cast<const unsigned char*>(x);
  auto wait_batches = [&](int younger) __attribute__((always_inline)) {  // at most `younger` batches still in flight
    switch (younger) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  static_assert(kColRows == 4 && DEPTH >= 2 && DEPTH <= 4, "the vmcnt table above");
  for (int64_t ti = blockIdx.x; ti < ntask; ti += gridDim.x) {
    int64_t task = ti + rot;
    if (task >= ntask) task -= ntask;
    const int64_t b = task / nchunk;
    const int64_t f0 = (task % nchunk) * CW + (int64_t)lane * EPP;
    const bool in = f0 < F;  // F % EPP == 0 on this path; lane 0 of a chunk is always in
    const unsigned char* base = x0 + (b * sb + (in ? f0 : 0)) * (int64_t)sizeof(E);
    const int64_t row_bytes = st * (int64_t)sizeof(E);
    const bool stream = task < tail_from;  // wave-uniform cache policy (top of this file)
    auto issue = [&](int jb) __attribute__((always_inline)) {
      const int t0 = t_begin + kColRows * (w + NW * jb);
      unsigned char* d = ring + (jb % DEPTH) * kSlot;
#pragma unroll
      for (int r = 0; r < kColRows; ++r) {
        const bool ok = t0 + r < t_end;
        unsigned char* dst = ok ? d + r * 1024 : spare;
        const unsigned char* src = ok ? base + (int64_t)(t0 + r) * row_bytes : x0;
        if ((ok && in) || lane == 0) {
          if (stream) __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)dst, 16, 0, 2 /* nt */);
          else __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)dst, 16, 0, 0);
        }
      }
    };
    Acc<OP> acc[EPP];
#pragma unroll
    for (int e = 0; e < EPP; ++e) acc[e].init();
    for (int jb = 0; jb < DEPTH - 1 && jb < nbw; ++jb) issue(jb);
#pragma unroll 1
    for (int jb = 0; jb < nbw; ++jb) {
      if (jb + DEPTH - 1 < nbw) issue(jb + DEPTH - 1);  // into the slot that was reduced one iteration ago
      const int left = nbw - 1 - jb;
      wait_batches(left < DEPTH - 1 ? left : DEPTH - 1);
      const int t0 = t_begin + kColRows * (w + NW * jb);
      const unsigned char* sl = ring + (jb % DEPTH) * kSlot + lane * 16;
      u32x4 v[kColRows];
#pragma unroll
      for (int r = 0; r < kColRows; ++r) v[r] = *reinterpret_cast<const u32x4*>(sl + r * 1024);
#pragma unroll
      for (int r = 0; r < kColRows; ++r) {
        const bool ok = t0 + r < t_end;
        if constexpr (EPP == 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e].add(bits_f32(v[r][e]), ok);
        } else {
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            float lo, hi;
            unpack2<E>(v[r][d], lo, hi);
            acc[2 * d].add(lo, ok);
            acc[2 * d + 1].add(hi, ok);
          }
        }
      }
    }
    // combine the waves: each leaves its partials in its own ring (all of its DMAs and reads are done)
    float* part = reinterpret_cast<float*>(ring);
#pragma unroll
    for (int e = 0; e < EPP; ++e) part[lane * EPP + e] = acc[e].lane_value();
    __syncthreads();
    for (int f = threadIdx.x; f < CW; f += 64 * NW) {
      const int64_t fg = (task % nchunk) * CW + f;
      if (fg < F) {
        float r = reinterpret_cast<const float*>(smem)[f];
#pragma unroll
        for (int i = 1; i < NW; ++i) r = combine<SUM>(r, reinterpret_cast<const float*>(smem + i * (DEPTH * kSlot))[f]);
        r = round_to_dtype<E>(finish<OP>(r, denom));  // the reference aggregates in the activation's dtype
        store_outputs(r, b * F + fg, cand, outf);
      }
    }
    __syncthreads();
  }
}


// colreduce through the LDS-DMA rings when the layout allows and the input is large enough to be bandwidth-bound.
// SL_COLREDUCE_IMPL = vgpr | dma (default: dma where legal), SL_COLREDUCE_DEPTH = 2 | 3 | 4, SL_COLREDUCE_NW = 4 | 8.
template <typename T, int OP, int NW, int DEPTH>
void launch_colreduce_dma_as(ProfScope& prof, const T* x, int64_t B, int T_, int64_t F, int64_t sb, int64_t st_, int t0, int t1,
                             float denom, int64_t tail_from, uint16_t* cand, float* outf, hipStream_t st) {
  constexpr int CW = 64 * (16 / (int)sizeof(T));
  const int lds = NW * DEPTH * kColRows * 1024 + 1024;
  auto kernel = colreduce_dma_kernel<T, OP, NW, DEPTH>;
  static bool once = [&] {
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    return true;
  }();
  (void)once;
  int64_t blocks = B * ((F + CW - 1) / CW);
  const int64_t cap = (int64_t)num_cus() * 8;
  if (blocks > cap) blocks = cap;
  SL_LAUNCH(prof, kernel, dim3((unsigned)blocks), dim3(64 * NW), lds, st, x, B, T_, F, sb, st_, t0, t1, denom, tail_from, cand, outf);
}

template <typename T, int OP>
bool launch_colreduce_dma(ProfScope& prof, const T* x, int64_t B, int T_, int64_t F, int64_t sb, int64_t st_, int t0, int t1,
                          float denom, uint16_t* cand, float* outf, hipStream_t st) {
  static const int impl = [] {  // only on request: the lab variant (tools/k2_lab.py)
    const char* e = getenv("SL_COLREDUCE_IMPL");
    return e && strcmp(e, "dma") == 0 ? 1 : 0;
  }();
  static const int forced_depth = [] {
    const char* e = getenv("SL_COLREDUCE_DEPTH");
    return e ? atoi(e) : 0;
  }();
  static const int forced_nw = [] {
    const char* e = getenv("SL_COLREDUCE_NW");
    return e ? atoi(e) : 0;
  }();
  constexpr int EPP = 16 / (int)sizeof(T);
  constexpr int CW = 64 * EPP;
  const int64_t rows = t1 - t0;
  const int64_t bytes = B * rows * F * (int64_t)sizeof(T);
  if (!impl || ((uintptr_t)x & 15) != 0 || (F % EPP) != 0 || ((st_ * (int64_t)sizeof(T)) & 15) != 0 ||
      ((sb * (int64_t)sizeof(T)) & 15) != 0 || rows < 16 || bytes < (8ll << 20))
    return false;
  const int64_t nchunk = (F + CW - 1) / CW, tasks = B * nchunk, cus = num_cus();
  if (tasks * 2 < cus) return false;  // few long tasks: the 16-wave VGPR kernel
  // cache policy (top of this file), in tasks = (b, chunk) pairs, b-major like the bytes
  const int64_t nt_min_bytes = nt_min_bytes_(), tail_bytes = tail_bytes_();
  const int64_t per_b = (int64_t)T_ * F * (int64_t)sizeof(T), all = B * per_b;
  int64_t tail_from = 0;
  if (all >= nt_min_bytes) tail_from = tail_bytes > 0 ? (all > tail_bytes ? (all - tail_bytes) / per_b * nchunk : 0) : INT64_MAX;
  int nw = (tasks < 2 * cus && rows >= 64) ? 8 : 4;
  if (forced_nw == 4 || forced_nw == 8) nw = forced_nw;
  int depth = kColDepthDefault;
  if (forced_depth >= 2 && forced_depth <= 4) depth = forced_depth;
#define SL_COLDMA(NW_, D_) launch_colreduce_dma_as<T, OP, NW_, D_>(prof, x, B, T_, F, sb, st_, t0, t1, denom, tail_from, cand, outf, st)
  if (nw == 8) {
    if (depth == 2) SL_COLDMA(8, 2);
    else if (depth == 3) SL_COLDMA(8, 3);
    else SL_COLDMA(8, 4);
  } else {
    if (depth == 2) SL_COLDMA(4, 2);
    else if (depth == 3) SL_COLDMA(4, 3);
    else SL_COLDMA(4, 4);
  }
#undef SL_COLDMA
  return true;
}

