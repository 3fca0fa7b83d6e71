// K4 with its exchange step: RCCL behind the C ABI (SURVEY.md §2.2 K4, §8b "the .so links librccl", §8e).
//
// The reference is single-process (no torch.distributed / NCCL call anywhere: SURVEY.md §2.1); the sharded build is the
// extension its path admits.  Rank r collects the contiguous sample range shard_range(N, r, R) into a full (C, k) state per
// layer; ONE ncclAllGather of all layers' packed states (sum_l C_l * k * 10 bytes per rank: 717 KB for ResNet-50 layer2-4 at
// k = 20 — latency-bound, far below xGMI's 153 GB/s per link) followed by the K4 merge kernel reading the other ranks'
// blocks in place gives every rank the global top-k.  The communicator is the one object of this library that outlives
// a call (sl_comm_init_from_unique_id / sl_comm_destroy); everything is enqueued on the caller's stream.
#include <rccl/rccl.h>

#include <cstring>
#include <vector>

#include "common.hpp"

namespace sl {
int merge_states_strided(const char* fn, uint16_t* d_vals, int64_t* d_ids, int64_t C, int64_t k, const uint16_t* d_other_vals,
                         const int64_t* d_other_ids, int64_t R, int64_t stride_v, int64_t stride_i, int64_t skip, hipStream_t st);

namespace {

struct Comm {
  ncclComm_t nccl;
  int world, rank, device;
};

int nccl_fail(ncclResult_t r, const char* what) {
  set_error("RCCL error %d (%s) in %s", (int)r, ncclGetErrorString(r), what);
  return SL_E_HIP;
}

#define SL_CHECK_NCCL(expr)                                  \
  do {                                                       \
    ncclResult_t _r = (expr);                                \
    if (_r != ncclSuccess) return ::sl::nccl_fail(_r, #expr); \
  } while (0)

inline int64_t round16(int64_t n) { return (n + 15) & ~(int64_t)15; }

// one rank's packed block: ids of every layer (8-byte aligned), then values of every layer, padded to 16 bytes
int64_t packed_bytes(int n_layers, const int64_t* h_C, int64_t k) {
  int64_t n = 0;
  for (int l = 0; l < n_layers; ++l) n += h_C[l] * k * 10;
  return round16(n);
}

}  // namespace
}  // namespace sl

using namespace sl;

static_assert(sizeof(ncclUniqueId) == SL_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");

SL_API int sl_comm_unique_id(uint8_t* h_id) {
  SL_REQUIRE(h_id, "sl_comm_unique_id: null output");
  ncclUniqueId id;
  SL_CHECK_NCCL(ncclGetUniqueId(&id));
  memcpy(h_id, &id, sizeof(id));
  return 0;
}

SL_API int sl_comm_init_from_unique_id(const uint8_t* h_id, int world, int rank, void** comm) {
  SL_REQUIRE(h_id && comm, "sl_comm_init_from_unique_id: null argument");
  SL_REQUIRE(world >= 1 && rank >= 0 && rank < world, "sl_comm_init_from_unique_id: rank %d of %d", rank, world);
  ncclUniqueId id;
  memcpy(&id, h_id, sizeof(id));
  Comm* c = new Comm{nullptr, world, rank, 0};
  hipError_t e = hipGetDevice(&c->device);
  if (e != hipSuccess) {
    delete c;
    return hip_fail(e, "hipGetDevice");
  }
  ncclResult_t r = ncclCommInitRank(&c->nccl, world, id, rank);  // one communicator per process, on the current device
  if (r != ncclSuccess) {
    delete c;
    return nccl_fail(r, "ncclCommInitRank");
  }
  *comm = c;
  return 0;
}

SL_API int sl_comm_destroy(void* comm) {
  if (!comm) return 0;
  Comm* c = (Comm*)comm;
  ncclResult_t r = ncclCommDestroy(c->nccl);
  delete c;
  if (r != ncclSuccess) return nccl_fail(r, "ncclCommDestroy");
  return 0;
}

SL_API int sl_comm_info(void* comm, int* world, int* rank, int* device) {
  SL_REQUIRE(comm, "sl_comm_info: null communicator");
  Comm* c = (Comm*)comm;
  if (world) *world = c->world;
  if (rank) *rank = c->rank;
  if (device) *device = c->device;
  return 0;
}

SL_API int sl_comm_allgather(void* comm, const void* d_send, void* d_recv, int64_t nbytes, void* stream) {
  SL_REQUIRE(comm, "sl_comm_allgather: null communicator");
  SL_REQUIRE(nbytes >= 0, "sl_comm_allgather: negative size");
  if (nbytes == 0) return 0;
  SL_REQUIRE(d_send && d_recv, "sl_comm_allgather: null buffer");
  SL_CHECK_NCCL(ncclAllGather(d_send, d_recv, (size_t)nbytes, ncclUint8, ((Comm*)comm)->nccl, (hipStream_t)stream));
  return 0;
}

SL_API int sl_comm_allreduce(void* comm, void* d_buf, int64_t n, int dtype, int op, void* stream) {
  SL_REQUIRE(comm, "sl_comm_allreduce: null communicator");
  SL_REQUIRE(n >= 0, "sl_comm_allreduce: negative count");
  if (n == 0) return 0;
  SL_REQUIRE(d_buf, "sl_comm_allreduce: null buffer");
  ncclDataType_t dt;
  switch (dtype) {
    case SL_COMM_F32: dt = ncclFloat32; break;
    case SL_COMM_F64: dt = ncclFloat64; break;
    case SL_COMM_I64: dt = ncclInt64; break;
    default: SL_REQUIRE(false, "sl_comm_allreduce: dtype %d", dtype);
  }
  ncclRedOp_t ro;
  switch (op) {
    case SL_COMM_SUM: ro = ncclSum; break;
    case SL_COMM_MAX: ro = ncclMax; break;
    case SL_COMM_MIN: ro = ncclMin; break;
    default: SL_REQUIRE(false, "sl_comm_allreduce: op %d", op);
  }
  SL_CHECK_NCCL(ncclAllReduce(d_buf, d_buf, (size_t)n, dt, ro, ((Comm*)comm)->nccl, (hipStream_t)stream));
  return 0;
}

namespace sl {
namespace {

struct PackLayout {
  std::vector<int64_t> off_i, off_v;
  int64_t used = 0, P = 0;
};

// [ids of layer 0 .. L-1 | values of layer 0 .. L-1 | pad to 16] — the layout of distributed.pack_states
int pack_layout(const char* fn, int n_layers, uint16_t* const* h_vals, int64_t* const* h_ids, const int64_t* h_C, int64_t k,
                PackLayout* lay) {
  SL_REQUIRE(n_layers >= 0 && k >= 0, "%s: negative shape", fn);
  SL_REQUIRE(n_layers == 0 || (h_vals && h_ids && h_C), "%s: null layer tables", fn);
  lay->off_i.resize(n_layers);
  lay->off_v.resize(n_layers);
  int64_t off = 0;
  for (int l = 0; l < n_layers; ++l) {
    SL_REQUIRE(h_C[l] >= 0 && (h_C[l] * k == 0 || (h_vals[l] && h_ids[l])), "%s: layer %d has a null state", fn, l);
    lay->off_i[l] = off;
    off += h_C[l] * k * 8;
  }
  for (int l = 0; l < n_layers; ++l) {
    lay->off_v[l] = off;
    off += h_C[l] * k * 2;
  }
  lay->used = off;
  lay->P = round16(off);
  return 0;
}

int pack(const PackLayout& lay, int n_layers, uint16_t* const* h_vals, int64_t* const* h_ids, const int64_t* h_C, int64_t k,
         unsigned char* send, hipStream_t st) {
  for (int l = 0; l < n_layers; ++l) {
    if (h_C[l] * k == 0) continue;
    SL_CHECK_HIP(hipMemcpyAsync(send + lay.off_i[l], h_ids[l], (size_t)(h_C[l] * k * 8), hipMemcpyDeviceToDevice, st));
    SL_CHECK_HIP(hipMemcpyAsync(send + lay.off_v[l], h_vals[l], (size_t)(h_C[l] * k * 2), hipMemcpyDeviceToDevice, st));
  }
  if (lay.used < lay.P) SL_CHECK_HIP(hipMemsetAsync(send + lay.used, 0, (size_t)(lay.P - lay.used), st));
  return 0;
}

// K4 of every layer against R packed blocks read in place (block r at recv + r * P), block `skip` left out
int merge_packed(const char* fn, const PackLayout& lay, int n_layers, uint16_t* const* h_vals, int64_t* const* h_ids,
                 const int64_t* h_C, int64_t k, const unsigned char* recv, int64_t R, int64_t skip, hipStream_t st) {
  for (int l = 0; l < n_layers; ++l) {
    if (h_C[l] * k == 0) continue;
    if (int rc = merge_states_strided(fn, h_vals[l], h_ids[l], h_C[l], k, (const uint16_t*)(recv + lay.off_v[l]),
                                      (const int64_t*)(recv + lay.off_i[l]), R, lay.P / 2, lay.P / 8, skip, st))
      return rc;
  }
  return 0;
}

}  // namespace
}  // namespace sl

SL_API size_t sl_actmax_packed_bytes(int n_layers, const int64_t* h_C, int64_t k) {
  if (n_layers <= 0 || !h_C || k <= 0) return 0;
  return (size_t)packed_bytes(n_layers, h_C, k);
}

SL_API int sl_actmax_pack(int n_layers, uint16_t* const* h_vals, int64_t* const* h_ids, const int64_t* h_C, int64_t k,
                          void* d_out, void* stream) {
  PackLayout lay;
  if (int rc = pack_layout("sl_actmax_pack", n_layers, h_vals, h_ids, h_C, k, &lay)) return rc;
  if (lay.P == 0) return 0;
  SL_REQUIRE(d_out, "sl_actmax_pack: null output");
  return pack(lay, n_layers, h_vals, h_ids, h_C, k, (unsigned char*)d_out, (hipStream_t)stream);
}

SL_API int sl_actmax_merge_packed(int n_layers, uint16_t* const* h_vals, int64_t* const* h_ids, const int64_t* h_C, int64_t k,
                                  const void* d_gathered, int64_t R, int64_t skip_rank, void* stream) {
  PackLayout lay;
  if (int rc = pack_layout("sl_actmax_merge_packed", n_layers, h_vals, h_ids, h_C, k, &lay)) return rc;
  SL_REQUIRE(R >= 0, "sl_actmax_merge_packed: negative R");
  if (lay.P == 0 || R == 0) return 0;
  SL_REQUIRE(d_gathered && ((uintptr_t)d_gathered & 7) == 0, "sl_actmax_merge_packed: gathered buffer null or not 8-byte aligned");
  return merge_packed("sl_actmax_merge_packed", lay, n_layers, h_vals, h_ids, h_C, k, (const unsigned char*)d_gathered, R,
                      skip_rank, (hipStream_t)stream);
}

SL_API size_t sl_actmax_allgather_merge_ws_bytes(int n_layers, const int64_t* h_C, int64_t k, int world) {
  if (n_layers <= 0 || !h_C || k <= 0 || world <= 0) return 0;
  return (size_t)(packed_bytes(n_layers, h_C, k) * (int64_t)(world + 1));
}

SL_API int sl_actmax_allgather_merge(void* comm, int n_layers, uint16_t* const* h_vals, int64_t* const* h_ids,
                                     const int64_t* h_C, int64_t k, void* d_ws, size_t ws_bytes, void* stream) {
  SL_REQUIRE(comm, "sl_actmax_allgather_merge: null communicator");
  PackLayout lay;
  if (int rc = pack_layout("sl_actmax_allgather_merge", n_layers, h_vals, h_ids, h_C, k, &lay)) return rc;
  if (lay.P == 0) return 0;
  Comm* c = (Comm*)comm;
  hipStream_t st = (hipStream_t)stream;
  const int64_t P = lay.P;
  SL_REQUIRE(d_ws && ws_bytes >= (size_t)(P * (c->world + 1)), "sl_actmax_allgather_merge: workspace of %zu bytes, need %lld",
             ws_bytes, (long long)(P * (c->world + 1)));
  SL_REQUIRE(((uintptr_t)d_ws & 15) == 0, "sl_actmax_allgather_merge: workspace must be 16-byte aligned");
  unsigned char* send = (unsigned char*)d_ws;
  unsigned char* recv = send + P;
  if (int rc = pack(lay, n_layers, h_vals, h_ids, h_C, k, send, st)) return rc;
  SL_CHECK_NCCL(ncclAllGather(send, recv, (size_t)P, ncclUint8, c->nccl, st));
  return merge_packed("sl_actmax_allgather_merge", lay, n_layers, h_vals, h_ids, h_C, k, recv, c->world, c->rank, st);
}
