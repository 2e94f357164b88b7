// Multi-GPU entry points of the C-ABI (include/segvlad.h, "row-sharded index"): one process per GPU, every rank keeps a
// contiguous block of the reference-segment rows in its context; a query batch is searched against every shard and the
// per-shard top-k lists travel in ONE all-gather of packed 12-byte records {fp32 distance bits, int64 global id} over
// RCCL (xGMI inside a node), after which every rank merges to the global top-k -- by distance, ties by lower global id:
// exactly what a single index over all rows returns.
//
// RCCL is bound at run time (dlopen + dlsym of the five entry points used), not at link time: the single-GPU library has no
// dependency on it, and a process that already carries an RCCL (PyTorch-ROCm ships one) shares that copy instead of
// loading a second.  Every failure is an error code (SEGVLAD_ERR_COMM) with the RCCL message in segvlad_last_error.
#include <dlfcn.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// No RCCL headers on the build machine: the handful of declarations the run-time binding needs (RCCL keeps NCCL's ABI:
// a 128-byte unique id, an opaque communicator, the result / data-type enumerations below), so that the single-GPU build
// really has no dependency on RCCL -- neither at link time nor at compile time.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5,
               ncclFloat16 = 6, ncclHalf = 6, ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId* uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclCommAbort(ncclComm_t comm);
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream);
const char* ncclGetErrorString(ncclResult_t result);
}
#endif

#include <mutex>

#include "ctx.h"

namespace {

struct RcclApi {
  void* handle = nullptr;
  char origin[256] = {0};
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

RcclApi g_rccl;
std::mutex g_rccl_mu;
}  // namespace

// path of the RCCL to bind, from the environment variable SEGVLAD_RCCL_LIB as read by segvlad_create (the one place of the
// library that reads the environment); empty = the search order of rccl_load
char sv_rccl_lib_override[256] = {0};

namespace {

// nullptr on success, else a static description of what went wrong
const char* rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.handle) return nullptr;
  static char why[512];
  const char* env = sv_rccl_lib_override;
  // a copy that is already in the process first (RTLD_NOLOAD), then the system's
  struct Try { const char* name; int flags; };
  const Try tries[] = {{env, RTLD_NOW | RTLD_LOCAL},
                       {"librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD},
                       {"librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD},
                       {"librccl.so.1", RTLD_NOW | RTLD_LOCAL},
                       {"librccl.so", RTLD_NOW | RTLD_LOCAL},
                       {"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL}};
  void* h = nullptr;
  for (const Try& t : tries) {
    if (!t.name || !*t.name) continue;
    h = dlopen(t.name, t.flags);
    if (h) {
      snprintf(g_rccl.origin, sizeof(g_rccl.origin), "%s%s", t.name, (t.flags & RTLD_NOLOAD) ? " (already in the process)" : "");
      break;
    }
  }
  if (!h) {
    snprintf(why, sizeof(why), "RCCL not found (librccl.so / librccl.so.1; set SEGVLAD_RCCL_LIB): %s", dlerror());
    return why;
  }
#define SV_SYM(field, sym)                                                       \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, sym));        \
  if (!g_rccl.field) {                                                           \
    snprintf(why, sizeof(why), "%s has no symbol %s", g_rccl.origin, sym);       \
    dlclose(h);                                                                  \
    return why;                                                                  \
  }
  SV_SYM(GetUniqueId, "ncclGetUniqueId")
  SV_SYM(CommInitRank, "ncclCommInitRank")
  SV_SYM(CommDestroy, "ncclCommDestroy")
  SV_SYM(CommAbort, "ncclCommAbort")
  SV_SYM(AllGather, "ncclAllGather")
  SV_SYM(GetErrorString, "ncclGetErrorString")
#undef SV_SYM
  g_rccl.handle = h;
  return nullptr;
}

#define SV_RCCL(expr)                                                                                         \
  do {                                                                                                        \
    ncclResult_t _r = (expr);                                                                                 \
    if (_r != ncclSuccess)                                                                                    \
      return ctx->fail(SEGVLAD_ERR_COMM, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
  } while (0)

// per-shard list -> packed records with GLOBAL ids (a missing entry keeps id -1)
__global__ __launch_bounds__(256) void pack_topk_kernel(const float* __restrict__ d2, const int64_t* __restrict__ idx, int64_t total,
                                                        int64_t id_base, uint32_t* __restrict__ rec) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= total) return;
  const int64_t id = idx[j];
  const int64_t g = id >= 0 ? id + id_base : id;
  rec[3 * j + 0] = __float_as_uint(d2[j]);
  rec[3 * j + 1] = (uint32_t)((uint64_t)g & 0xffffffffull);
  rec[3 * j + 2] = (uint32_t)((uint64_t)g >> 32);
}

// gathered records [world][nq][k] (rank-major) -> the merge's layout [nq][world * k] (shard-major within a row)
// A rank's block is its nq * k records followed by ONE trailer record {local status, 0, 0} (see segvlad_search_sharded); the
// trailers' status words are collected into flags[world].
__global__ __launch_bounds__(256) void unpack_topk_kernel(const uint32_t* __restrict__ rec, int world, int nq, int k,
                                                          float* __restrict__ d2c, int64_t* __restrict__ idc,
                                                          uint32_t* __restrict__ flags) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per = (int64_t)nq * k, total = (int64_t)world * per;
  if (j < world) flags[j] = rec[3 * ((j + 1) * (per + 1) - 1)];
  if (j >= total) return;
  const int r = (int)(j / per);
  const int64_t rem = j - (int64_t)r * per;
  const int64_t q = rem / k;
  const int c = (int)(rem - q * k);
  const int64_t o = q * ((int64_t)world * k) + (int64_t)r * k + c;
  const int64_t src = 3 * (j + r);   // r trailers lie in front of rank r's records
  d2c[o] = __uint_as_float(rec[src]);
  idc[o] = (int64_t)((uint64_t)rec[src + 1] | ((uint64_t)rec[src + 2] << 32));
}

}  // namespace

void sv_comm_release(segvlad_ctx* ctx) {
  if (ctx->comm && g_rccl.handle) (void)g_rccl.CommDestroy(reinterpret_cast<ncclComm_t>(ctx->comm));
  ctx->comm = nullptr;
  ctx->comm_rank = 0;
  ctx->comm_world = 1;
}

extern "C" {

int segvlad_comm_unique_id(void* id_out) {
  if (!id_out) return SEGVLAD_ERR_ARG;
  if (rccl_load()) return SEGVLAD_ERR_COMM;
  static_assert(sizeof(ncclUniqueId) == SEGVLAD_COMM_ID_BYTES, "segvlad.h promises 128 bytes");
  ncclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != ncclSuccess) return SEGVLAD_ERR_COMM;
  memcpy(id_out, &id, sizeof(id));
  return SEGVLAD_OK;
}

int segvlad_comm_init(segvlad_ctx* ctx, const void* id, int rank, int world) {
  if (!ctx) return SEGVLAD_ERR_ARG;
  sv_begin(ctx);
  if (!id || world < 1 || rank < 0 || rank >= world) return ctx->fail(SEGVLAD_ERR_ARG, "comm_init: need id, 0 <= rank < world");
  if (const char* why = rccl_load()) return ctx->fail(SEGVLAD_ERR_COMM, "comm_init: %s", why);
  sv_comm_release(ctx);
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t comm = nullptr;
  SV_RCCL(g_rccl.CommInitRank(&comm, world, uid, rank));
  ctx->comm = comm;
  ctx->comm_rank = rank;
  ctx->comm_world = world;
  return SEGVLAD_OK;
}

int segvlad_comm_destroy(segvlad_ctx* ctx) {
  if (!ctx) return SEGVLAD_ERR_ARG;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  sv_comm_release(ctx);
  return SEGVLAD_OK;
}

int segvlad_comm_info(segvlad_ctx* ctx, int* rank_out, int* world_out, char* origin_out, int origin_len) {
  if (!ctx) return SEGVLAD_ERR_ARG;
  if (rank_out) *rank_out = ctx->comm ? ctx->comm_rank : 0;
  if (world_out) *world_out = ctx->comm ? ctx->comm_world : 0;   // 0 = no communicator
  if (origin_out && origin_len > 0) snprintf(origin_out, (size_t)origin_len, "%s", g_rccl.origin);
  return SEGVLAD_OK;
}

// A collective must be ENTERED by every rank or by none.  Argument / state errors (the same on every rank of a correct
// program) return before it; a failure only THIS rank can have -- out of memory for the exchange buffers -- cannot join the
// collective any more and aborts the communicator instead (ncclCommAbort: the peers' pending collective fails instead of
// waiting for ever; the communicator is gone on this rank, segvlad_comm_init makes a new one).
static int comm_abort_local(segvlad_ctx* ctx, const char* what, int rc) {
  char why[400];
  snprintf(why, sizeof(why), "%s", ctx->err);
  if (ctx->comm && g_rccl.CommAbort) (void)g_rccl.CommAbort(reinterpret_cast<ncclComm_t>(ctx->comm));
  ctx->comm = nullptr;
  ctx->comm_rank = 0;
  ctx->comm_world = 1;
  return ctx->fail(rc, "%s: local failure before the collective (%s); the communicator was aborted", what, why);
}

int segvlad_allgather_rows(segvlad_ctx* ctx, const float* local_rows, int n_local, int d, float* all_rows) {
  if (!ctx) return SEGVLAD_ERR_ARG;
  sv_begin(ctx);
  if (!ctx->comm) return ctx->fail(SEGVLAD_ERR_STATE, "allgather_rows: call segvlad_comm_init first");
  if (n_local < 0 || d <= 0) return ctx->fail(SEGVLAD_ERR_ARG, "allgather_rows: bad shape");
  if (n_local == 0) return SEGVLAD_OK;   // (n_local is the same on every rank: segvlad.h)
  if (!local_rows || !all_rows) return ctx->fail(SEGVLAD_ERR_ARG, "allgather_rows: null pointer");
  const void* din;
  void* dout;
  int rc = sv_in(ctx, local_rows, (size_t)n_local * d * 4, &din);
  if (rc == SEGVLAD_OK) rc = sv_out(ctx, all_rows, (size_t)ctx->comm_world * n_local * d * 4, &dout);
  if (rc != SEGVLAD_OK) return comm_abort_local(ctx, "allgather_rows", rc);
  {
    const ncclResult_t r = g_rccl.AllGather(din, dout, (size_t)n_local * d, ncclFloat, reinterpret_cast<ncclComm_t>(ctx->comm), ctx->stream);
    if (r != ncclSuccess) {
      (void)ctx->fail(SEGVLAD_ERR_COMM, "ncclAllGather failed: %s", g_rccl.GetErrorString(r));
      return comm_abort_local(ctx, "allgather_rows", SEGVLAD_ERR_COMM);
    }
  }
  return sv_finish(ctx);
}

int segvlad_search_sharded(segvlad_ctx* ctx, const float* Q, int nq, int k, int64_t id_base, float* d2_out, int64_t* idx_out) {
  if (!ctx) return SEGVLAD_ERR_ARG;
  sv_begin(ctx);
  if (!ctx->comm) return ctx->fail(SEGVLAD_ERR_STATE, "search_sharded: call segvlad_comm_init first");
  if (nq < 0 || k < 1 || k > 1024) return ctx->fail(SEGVLAD_ERR_ARG, "search_sharded: need nq >= 0 and 1 <= k <= 1024");
  if (nq == 0) return SEGVLAD_OK;
  if (!Q || !d2_out || !idx_out) return ctx->fail(SEGVLAD_ERR_ARG, "search_sharded: null pointer");
  const int world = ctx->comm_world;
  const int64_t total = (int64_t)nq * k;
  // local lists, the packed records of this rank (+ its trailer record) and of all ranks, the merge's operands, the flags
  hipError_t he = ctx->s_sh_d2.reserve((size_t)total * 4);
  if (he == hipSuccess) he = ctx->s_sh_idx.reserve((size_t)total * 8);
  if (he == hipSuccess) he = ctx->s_sh_rec.reserve((size_t)(total + 1) * 12);
  if (he == hipSuccess) he = ctx->s_sh_all.reserve((size_t)world * (total + 1) * 12);
  if (he == hipSuccess) he = ctx->s_sh_d2c.reserve((size_t)world * total * 4);
  if (he == hipSuccess) he = ctx->s_sh_idc.reserve((size_t)world * total * 8 + (size_t)world * 4);
  if (he != hipSuccess) {
    (void)ctx->fail(SEGVLAD_ERR_NOMEM, "exchange buffers: %s", hipGetErrorString(he));
    return comm_abort_local(ctx, "search_sharded", SEGVLAD_ERR_NOMEM);
  }
  float* ld2 = ctx->s_sh_d2.as<float>();
  int64_t* lidx = ctx->s_sh_idx.as<int64_t>();
  uint32_t* flags = reinterpret_cast<uint32_t*>(ctx->s_sh_idc.as<int64_t>() + (size_t)world * total);
  // The local search may fail on THIS rank only (its shard, its memory).  The rank then still enters the all-gather -- with
  // (inf, -1) records, which never win a merge, and its status in the trailer record -- so that every rank leaves the
  // collective and every rank returns an error (the failing rank its own, the others SEGVLAD_ERR_COMM naming it).
  int local_rc = SEGVLAD_OK;
  char local_err[sizeof(ctx->err)] = {0};
  if (ctx->db_n > 0) {
    // (segvlad_search resets the staging state: Q is handed over as it came -- host or device)
    local_rc = segvlad_search(ctx, Q, nq, k, ld2, lidx);
    if (local_rc != SEGVLAD_OK) snprintf(local_err, sizeof(local_err), "%s", ctx->err);
    sv_begin(ctx);
  }
  if (ctx->opt.debug_fail_search == 2) {   // tests: a failure that can no longer join the collective
    (void)ctx->fail(SEGVLAD_ERR_STATE, "search_sharded: failing in front of the collective on request (option debug_fail_search = 2)");
    return comm_abort_local(ctx, "search_sharded", SEGVLAD_ERR_STATE);
  }
  // From here to the all-gather the peers may already be INSIDE the collective: a failure on this rank must not simply return
  // (a failed local search may have left a sticky HIP error, and then the very next memset fails as well): it aborts the
  // communicator, so that the peers' collective fails instead of waiting for ever.
#define SV_PRE(expr)                                                                                        \
  do {                                                                                                      \
    hipError_t _e = (expr);                                                                                 \
    if (_e != hipSuccess) {                                                                                 \
      (void)ctx->fail(SEGVLAD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return comm_abort_local(ctx, "search_sharded", SEGVLAD_ERR_HIP);                                      \
    }                                                                                                       \
  } while (0)
  if (ctx->db_n <= 0 || local_rc != SEGVLAD_OK) {   // an empty (or failed) shard contributes (inf, -1)
    SV_PRE(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ld2), 0x7f800000, (size_t)total, ctx->stream));
    SV_PRE(hipMemsetAsync(lidx, 0xff, (size_t)total * 8, ctx->stream));
  }
  void *od, *oi;
  int rc_out = sv_out(ctx, d2_out, (size_t)total * 4, &od);
  if (rc_out == SEGVLAD_OK) rc_out = sv_out(ctx, idx_out, (size_t)total * 8, &oi);
  if (rc_out != SEGVLAD_OK) return comm_abort_local(ctx, "search_sharded", rc_out);
  // the status words of all ranks land in a buffer the CONTEXT owns: the copy is asynchronous, and an error path that left
  // this frame before the stream was synchronised would have had the copy write into a dead stack vector
  ctx->sh_flags_host.assign((size_t)world, 0u);
  int post_rc = SEGVLAD_OK;   // failures behind the collective: remembered, the stream is synchronised either way
  {
    StageScope sc(ctx, "shard_exchange");
    hipLaunchKernelGGL(pack_topk_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, ld2, lidx, total, id_base,
                       ctx->s_sh_rec.as<uint32_t>());
    SV_PRE(hipGetLastError());
    SV_PRE(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->s_sh_rec.as<uint32_t>() + 3 * total), local_rc != SEGVLAD_OK ? 1 : 0, 3,
                             ctx->stream));   // the trailer record: {status, status, status}
    {
      const ncclResult_t r = g_rccl.AllGather(ctx->s_sh_rec.p, ctx->s_sh_all.p, (size_t)(total + 1) * 12, ncclUint8,
                                              reinterpret_cast<ncclComm_t>(ctx->comm), ctx->stream);
      if (r != ncclSuccess) {   // not enqueued on this rank (or the peers are gone): nothing to wait for, nobody to leave waiting
        (void)ctx->fail(SEGVLAD_ERR_COMM, "ncclAllGather failed: %s", g_rccl.GetErrorString(r));
        return comm_abort_local(ctx, "search_sharded", SEGVLAD_ERR_COMM);
      }
    }
#undef SV_PRE
    const int64_t all = (int64_t)world * total;
    hipLaunchKernelGGL(unpack_topk_kernel, dim3((unsigned)((all + 255) / 256)), dim3(256), 0, ctx->stream, ctx->s_sh_all.as<uint32_t>(),
                       world, nq, k, ctx->s_sh_d2c.as<float>(), ctx->s_sh_idc.as<int64_t>(), flags);
    hipError_t he2 = hipGetLastError();
    if (he2 == hipSuccess)
      he2 = hipMemcpyAsync(ctx->sh_flags_host.data(), flags, (size_t)world * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (he2 != hipSuccess) post_rc = ctx->fail(SEGVLAD_ERR_HIP, "search_sharded: behind the collective: %s", hipGetErrorString(he2));
    sc.count(3);
  }
  if (post_rc == SEGVLAD_OK) {
    StageScope sc(ctx, "shard_merge");
    post_rc = sv_launch_merge_topk(ctx, ctx->s_sh_d2c.as<float>(), ctx->s_sh_idc.as<int64_t>(), nq, world * k, k, (float*)od, (int64_t*)oi);
    sc.count();
  }
  {
    const hipError_t hs = hipStreamSynchronize(ctx->stream);   // the status words of all ranks (4 bytes each)
    if (hs != hipSuccess && post_rc == SEGVLAD_OK)
      post_rc = ctx->fail(SEGVLAD_ERR_HIP, "search_sharded: hipStreamSynchronize: %s", hipGetErrorString(hs));
  }
  if (post_rc != SEGVLAD_OK) return post_rc;
  const int rc_fin = sv_finish(ctx);
  if (local_rc != SEGVLAD_OK) return ctx->fail(local_rc, "search_sharded: this rank's local search failed: %s", local_err);
  for (int r = 0; r < world; ++r)
    if (ctx->sh_flags_host[(size_t)r])
      return ctx->fail(SEGVLAD_ERR_COMM, "search_sharded: rank %d failed its local search (every rank returns an error)", r);
  return rc_fin;
}

}  // extern "C"
