// C-ABI of libsegvlad_hip.so (see include/segvlad.h).  Host-side orchestration only: argument
// checks, host/device pointer staging, scratch sizing and kernel sequencing on the context stream.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>
#include <mutex>
#include <utility>

#include <initializer_list>

#include "ctx.h"
#include "small_pass_dev.h"

int segvlad_ctx::fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err, sizeof(err), fmt, ap);
  va_end(ap);
  return code;
}

bool sv_is_device_ptr(const void* p) {
  hipPointerAttribute_t a;
  memset(&a, 0, sizeof(a));
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // unregistered host memory reports an error on some runtimes
    return false;
  }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged || a.type == hipMemoryTypeUnified;
}

void sv_begin(segvlad_ctx* ctx) {
  ctx->stage_used = 0;
  ctx->pending_out.clear();
  ctx->err[0] = 0;
  (void)hipSetDevice(ctx->device);
}

static DevBuf* next_stage(segvlad_ctx* ctx) {
  if (ctx->stage_used >= (int)ctx->stage.size()) {
    ctx->stage.resize(ctx->stage_used + 1);
    ctx->stage.back().tag = "stage";
    ctx->stage.back().guard = ctx->guard;
  }
  return &ctx->stage[ctx->stage_used++];
}

int sv_in(segvlad_ctx* ctx, const void* p, size_t bytes, const void** dev) {
  if (bytes == 0) {
    *dev = p;
    return SEGVLAD_OK;
  }
  if (sv_is_device_ptr(p)) {
    *dev = p;
    return SEGVLAD_OK;
  }
  DevBuf* b = next_stage(ctx);
  SV_HIP(b->reserve(bytes));
  SV_HIP(hipMemcpyAsync(b->p, p, bytes, hipMemcpyHostToDevice, ctx->stream));
  // pageable host memory: hipMemcpyAsync returns after the staging copy, so the caller may reuse p
  *dev = b->p;
  return SEGVLAD_OK;
}

int sv_out(segvlad_ctx* ctx, void* p, size_t bytes, void** dev) {
  if (bytes == 0 || sv_is_device_ptr(p)) {
    *dev = p;
    return SEGVLAD_OK;
  }
  DevBuf* b = next_stage(ctx);
  SV_HIP(b->reserve(bytes));
  ctx->pending_out.push_back({p, b->p, bytes});
  *dev = b->p;
  return SEGVLAD_OK;
}

int sv_finish(segvlad_ctx* ctx) {
  if (!ctx->pending_out.empty()) {
    for (auto& po : ctx->pending_out) SV_HIP(hipMemcpyAsync(po.host, po.dev, po.bytes, hipMemcpyDeviceToHost, ctx->stream));
    SV_HIP(hipStreamSynchronize(ctx->stream));
    ctx->pending_out.clear();
  }
  if (ctx->guard) return sv_guard_check(ctx);
  return SEGVLAD_OK;
}

// ---- guard mode (ctx.h: DevBuf) -----------------------------------------------------------------------------------
static hipError_t guard_fill(void* at) { return hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(at), (int)SV_GUARD_WORD, SV_GUARD_BYTES / 4); }

// number of fence words that no longer hold the poison (0 = intact); < 0: a HIP error
static int guard_words_hit(const void* at) {
  uint32_t h[SV_GUARD_BYTES / 4];
  if (hipMemcpy(h, at, SV_GUARD_BYTES, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  int bad = 0;
  for (uint32_t w : h) bad += (w != SV_GUARD_WORD);
  return bad;
}

hipError_t DevBuf::reserve_guarded(size_t bytes) {
  // every earlier user of this buffer has finished (the fence may move INTO bytes an earlier, larger request used)
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return e;
  if (!raw || bytes > cap) {
    // (a trampled fence of the old allocation is caught by the check at the end of the call that trampled it)
    if (raw) {
      e = hipFree(raw);
      if (e != hipSuccess) return e;
      raw = nullptr;
      p = nullptr;
      cap = 0;
    }
    const size_t want = (bytes + 15) & ~(size_t)15;   // exact: no slack to absorb an overrun
    e = hipMalloc(&raw, want + 2 * SV_GUARD_BYTES);
    if (e != hipSuccess) return e;
    p = static_cast<unsigned char*>(raw) + SV_GUARD_BYTES;
    cap = want;
    req = cap;   // (where the back fence of a `fixed` buffer goes, and stays)
    e = guard_fill(raw);
    if (e == hipSuccess) e = guard_fill(static_cast<unsigned char*>(p) + cap);
    if (e != hipSuccess) return e;
  }
  if (!fixed) {
    req = bytes;
    e = guard_fill(static_cast<unsigned char*>(p) + fence_off());
  }
  return e;
}

int sv_fork_side(segvlad_ctx* ctx) {
  if (!ctx->side) {
    SV_HIP(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
    SV_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    SV_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  }
  SV_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
  SV_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
  return SEGVLAD_OK;
}

int sv_join_side(segvlad_ctx* ctx) {
  SV_HIP(hipEventRecord(ctx->ev_join, ctx->side));
  SV_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  return SEGVLAD_OK;
}

int sv_guard_check(segvlad_ctx* ctx) {
  if (!ctx->guard) return SEGVLAD_OK;
  SV_HIP(hipDeviceSynchronize());
  if (!ctx->guard_hit[0]) {
    ctx->for_each_buf([&](DevBuf& b) {
      if (!b.raw || ctx->guard_hit[0]) return;
      const int front = guard_words_hit(b.raw), back = guard_words_hit(static_cast<unsigned char*>(b.p) + b.fence_off());
      if (front)
        snprintf(ctx->guard_hit, sizeof(ctx->guard_hit), "%s: %d words written BELOW the buffer", b.tag, front);
      else if (back)
        snprintf(ctx->guard_hit, sizeof(ctx->guard_hit), "%s: %d words written beyond the %zu bytes requested", b.tag, back, b.fixed ? b.cap : b.req);
    });
  }
  if (ctx->guard_hit[0]) return ctx->fail(SEGVLAD_ERR_STATE, "guard: out-of-bounds write, buffer %s", ctx->guard_hit);
  return SEGVLAD_OK;
}

StageScope::StageScope(segvlad_ctx* c, const char* name) : ctx(c) {
  if (!c->profiling || c->scope_mute) return;
  t = &c->timers[name];
  if (t->used * 2 >= (int)t->ev.size()) {
    hipEvent_t a = nullptr, b = nullptr;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    t->ev.push_back(a);
    t->ev.push_back(b);
  }
  slot = t->used++;
  (void)hipEventRecord(t->ev[2 * slot], c->stream);
}
StageScope::~StageScope() {
  if (t) (void)hipEventRecord(t->ev[2 * slot + 1], ctx->stream);
}

#define CHECK_CTX()                 \
  if (!ctx) return SEGVLAD_ERR_ARG; \
  sv_begin(ctx)

// Between segvlad_describe_begin and segvlad_describe_end the context's per-batch scratch (segment / adjacency offsets, the
// token-major copy, the labels) belongs to THAT batch, and the mask branch is in flight on the side stream: every other entry
// point that would write them refuses instead of describing a different batch with the open one's offsets.
#define CHECK_NO_OPEN_DESCRIBE(what)                                                                                        \
  if (ctx->mask_branch_on_side)                                                                                             \
  return ctx->fail(SEGVLAD_ERR_STATE, what ": a segvlad_describe_begin is open on this context (call segvlad_describe_end first)")

hipError_t sv_max_dyn_lds(const void* fn, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> done;   // largest size granted so far
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lk(mu);
  auto it = done.find({dev, fn});
  if (it != done.end() && it->second >= bytes) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) done[{dev, fn}] = bytes;
  return e;
}

extern "C" {

int segvlad_version(void) { return 300; }

int segvlad_create(segvlad_ctx** out, int device_id) {
  if (!out) return SEGVLAD_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return SEGVLAD_ERR_HIP;
  if (hipSetDevice(device_id) != hipSuccess) return SEGVLAD_ERR_HIP;
  segvlad_ctx* c = new (std::nothrow) segvlad_ctx();
  if (!c) return SEGVLAD_ERR_NOMEM;
  c->device = device_id;
  // environment defaults of the tuning switches, read once here (segvlad_set_option overrides them later)
  static const char* const env_keys[][2] = {{"SEGVLAD_KNN_FILTER", "knn_filter"},   {"SEGVLAD_F16_CFG", "f16_cfg"},
                                            {"SEGVLAD_F16_GM", "f16_gm"},           {"SEGVLAD_X3_TILE", "x3_tile"},
                                            {"SEGVLAD_X3_GM", "x3_gm"},             {"SEGVLAD_SEARCH_STATS", "search_stats"},
                                            {"SEGVLAD_ASSIGN_NARROW", "assign_narrow"}, {"SEGVLAD_DEBUG_SEARCH", "debug_search"},
                                            {"SEGVLAD_AGG_KPB", "agg_kpb"},
                                            {"SEGVLAD_KNN_HEURISTIC", "knn_heuristic"}, {"SEGVLAD_PCA_PATH", "pca_path"}};
  for (auto& kv : env_keys)
    if (const char* v = getenv(kv[0])) (void)segvlad_set_option(c, kv[1], v);
#define SV_TAG_P(n) c->n.tag = #n; c->n.fixed = true;
#define SV_TAG_S(n) c->n.tag = #n;
  SV_PERSISTENT_BUFS(SV_TAG_P)
  SV_SCRATCH_BUFS(SV_TAG_S)
#undef SV_TAG_P
#undef SV_TAG_S
  if (const char* v = getenv("SEGVLAD_GUARD")) {
    if (v[0] && v[0] != '0') {
      c->guard = true;
      c->for_each_buf([](DevBuf& b) { b.guard = true; });
    }
  }
  if (const char* v = getenv("SEGVLAD_RCCL_LIB")) snprintf(sv_rccl_lib_override, sizeof(sv_rccl_lib_override), "%s", v);
  if (getenv("SEGVLAD_KNN_FP32")) (void)segvlad_set_option(c, "knn_filter", "fp32");
  if (getenv("SEGVLAD_PCA_FP32")) (void)segvlad_set_option(c, "pca_arith", "fp32");
  c->err[0] = 0;
  *out = c;
  return SEGVLAD_OK;
}

int segvlad_set_option(segvlad_ctx* ctx, const char* key, const char* value) {
  if (!ctx) return SEGVLAD_ERR_ARG;
  if (!key || !value) return ctx->fail(SEGVLAD_ERR_ARG, "set_option: null key/value");
  SvOptions& o = ctx->opt;
  auto as_int = [&](int* dst) -> int {
    char* end = nullptr;
    const long v = strtol(value, &end, 10);
    if (end == value || *end) return ctx->fail(SEGVLAD_ERR_ARG, "set_option(%s): '%s' is not an integer", key, value);
    *dst = (int)v;
    return SEGVLAD_OK;
  };
  if (!strcmp(key, "knn_filter")) {
    if (!strcmp(value, "auto")) o.knn_filter = 0;
    else if (!strcmp(value, "f16")) o.knn_filter = 1;
    else if (!strcmp(value, "bf16x3")) o.knn_filter = 2;
    else if (!strcmp(value, "fp32")) o.knn_filter = 3;
    else return ctx->fail(SEGVLAD_ERR_ARG, "set_option(knn_filter): want auto|f16|bf16x3|fp32, got '%s'", value);
    return SEGVLAD_OK;
  }
  if (!strcmp(key, "pca_arith")) {
    if (!strcmp(value, "auto") || !strcmp(value, "f16x3")) o.pca_fp32 = 0;
    else if (!strcmp(value, "fp32")) o.pca_fp32 = 1;
    else return ctx->fail(SEGVLAD_ERR_ARG, "set_option(pca_arith): want auto|f16x3|fp32, got '%s'", value);
    return SEGVLAD_OK;
  }
  if (!strcmp(key, "pca_path")) {
    if (!strcmp(value, "auto")) o.pca_path = 0;
    else if (!strcmp(value, "planes")) o.pca_path = 1;
    else if (!strcmp(value, "project")) o.pca_path = 2;
    else return ctx->fail(SEGVLAD_ERR_ARG, "set_option(pca_path): want auto|planes|project, got '%s'", value);
    return SEGVLAD_OK;
  }
  // Switches that select one of the fp16 filter's measured-and-not-kept kernel variants (csrc/segvlad_dev.h): the shipped
  // library holds the default kernels only and accepts just the values that mean "the default"; development builds
  // (-DSEGVLAD_ABLATIONS: lib/libsegvlad_hip_abl.so) hold every variant.
  auto as_dev = [&](int* dst, std::initializer_list<int> product_values) -> int {
    int v = *dst;
    SV_TRY(as_int(&v));
#ifndef SEGVLAD_ABLATIONS
    bool ok = false;
    for (int a : product_values) ok = ok || a == v;
    if (!ok)
      return ctx->fail(SEGVLAD_ERR_ARG, "set_option(%s=%d): a development switch -- this library holds the default kernel only "
                       "(SEGVLAD_BUILD_ABLATIONS=1 builds lib/libsegvlad_hip_abl.so)", key, v);
#else
    (void)product_values;
#endif
    *dst = v;
    return SEGVLAD_OK;
  };
  if (!strcmp(key, "f16_cfg")) return as_dev(&o.f16_cfg, {-1, 250, 300, 62, 63});
  if (!strcmp(key, "f16_gm")) return as_int(&o.f16_gm);
  if (!strcmp(key, "f16_walk")) return as_int(&o.f16_walk);
  if (!strcmp(key, "f16_epi")) return as_dev(&o.f16_epi, {-1, 1});
  if (!strcmp(key, "f16_mf")) return as_dev(&o.f16_mf, {-1, 1});
  if (!strcmp(key, "f16_deep_cfg")) return as_dev(&o.f16_deep_cfg, {-1, 4, 5});
  if (!strcmp(key, "f16_pp")) return as_dev(&o.f16_pp, {-1, 2});
  if (!strcmp(key, "f16_small_mf")) return as_dev(&o.f16_small_mf, {0});
  if (!strcmp(key, "f16_buf")) return as_dev(&o.f16_buf, {-1, 0});
  if (!strcmp(key, "f16_dsplit")) return as_dev(&o.f16_dsplit, {0});
  if (!strcmp(key, "tnk_gram")) return as_int(&o.tnk_gram);
  if (!strcmp(key, "tnk_fork")) return as_int(&o.tnk_fork);
  if (!strcmp(key, "x3_tile")) return as_int(&o.x3_tile);
  if (!strcmp(key, "x3_gm")) return as_int(&o.x3_gm);
  if (!strcmp(key, "search_stats")) return as_int(&o.search_stats);
  if (!strcmp(key, "knn_heuristic")) return as_int(&o.knn_heuristic);
  if (!strcmp(key, "assign_narrow")) return as_int(&o.assign_narrow);
  if (!strcmp(key, "agg_kpb")) return as_int(&o.agg_kpb);
  if (!strcmp(key, "pj_nw")) return as_int(&o.pj_nw);
  if (!strcmp(key, "pj_f16")) return as_int(&o.pj_f16);
  if (!strcmp(key, "small_plan")) return as_int(&o.small_plan);
  if (!strcmp(key, "small_tail")) return as_int(&o.small_tail);
  if (!strcmp(key, "small_head")) return as_int(&o.small_head);
  if (!strcmp(key, "batch_l0_f16")) return as_int(&o.batch_l0_f16);
  if (!strcmp(key, "level_carry")) return as_int(&o.level_carry);
  if (!strcmp(key, "f16_persist_wgs")) return as_int(&o.f16_persist_wgs);
  if (!strcmp(key, "debug_small_tail")) return as_int(&o.debug_small_tail);
  if (!strcmp(key, "refine_group")) return as_int(&o.refine_group);
  if (!strcmp(key, "query_group")) return as_int(&o.query_group);
  if (!strcmp(key, "debug_search")) return as_int(&o.debug_search);
  if (!strcmp(key, "debug_fail_search")) return as_int(&o.debug_fail_search);
  if (!strcmp(key, "guard_undersize")) {
    // tests of the guard itself: "<buffer tag>:<bytes>" puts that scratch buffer's back fence <bytes> EARLY from its next
    // reserve() on, so that a correct kernel writes into the fence (guard mode only; "" / "none" clears every shrink)
    if (!ctx->guard) return ctx->fail(SEGVLAD_ERR_STATE, "set_option(guard_undersize): the context was not created under SEGVLAD_GUARD=1");
    const char* colon = strchr(value, ':');
    bool found = false;
    ctx->for_each_buf([&](DevBuf& b) {
      if (!colon) b.shrink = 0;
      else if (strlen(b.tag) == (size_t)(colon - value) && !strncmp(b.tag, value, (size_t)(colon - value))) {
        b.shrink = atoi(colon + 1);
        found = true;
      }
    });
    if (colon && !found) return ctx->fail(SEGVLAD_ERR_ARG, "set_option(guard_undersize): no buffer named like '%s'", value);
    return SEGVLAD_OK;
  }
  return ctx->fail(SEGVLAD_ERR_ARG, "set_option: unknown key '%s'", key);
}

int segvlad_search_stats(segvlad_ctx* ctx, int64_t* stats_out, int n) {
  if (!ctx) return SEGVLAD_ERR_ARG;
  if (!stats_out || n < 0) return ctx->fail(SEGVLAD_ERR_ARG, "search_stats: bad arguments");
  if (ctx->tail_stats_dev) {   // the last search was a device-driven pass: its tail's counters are still on the device
    uint32_t h[4] = {0, 0, 0, 0};
    (void)hipSetDevice(ctx->device);
    SV_HIP(hipStreamSynchronize(ctx->stream));
    SV_HIP(hipMemcpy(h, ctx->tail_stats_dev, sizeof(h), hipMemcpyDeviceToHost));
    ctx->sstats.n_redo = h[0];      // rows redone exactly (brute force on the device)
    ctx->sstats.n_refine2 = h[1];   // rows refined from their candidate lists (second tier)
    ctx->tail_stats_dev = nullptr;
  }
  const SvSearchStats& t = ctx->sstats;
  const int64_t v[13] = {t.levels, t.filter, t.n_fallback, t.cand_max, t.cand_sum, t.refine_max, t.refine_sum, t.n_queries, t.n_redo, t.n_refine2,
                         t.grp_groups, t.grp_union_sum, t.carry_rows};
  for (int j = 0; j < n && j < 13; ++j) stats_out[j] = v[j];
  return SEGVLAD_OK;
}

int segvlad_destroy(segvlad_ctx* ctx) {
  if (!ctx) return SEGVLAD_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  sv_comm_release(ctx);
  if (ctx->side) {
    (void)hipStreamSynchronize(ctx->side);
    (void)hipStreamDestroy(ctx->side);
    (void)hipEventDestroy(ctx->ev_fork);
    (void)hipEventDestroy(ctx->ev_join);
  }
  if (ctx->h_pin) {
    (void)hipHostFree(ctx->h_pin);
    (void)hipEventDestroy(ctx->ev_scalars);
  }
  if (ctx->h_desc) (void)hipHostFree(ctx->h_desc);
  if (ctx->ev_desc) (void)hipEventDestroy(ctx->ev_desc);
  ctx->for_each_buf([](DevBuf& b) { b.release(); });
  for (auto& kv : ctx->timers)
    for (hipEvent_t e : kv.second.ev)
      if (e) (void)hipEventDestroy(e);
  delete ctx;
  return SEGVLAD_OK;
}

const char* segvlad_last_error(const segvlad_ctx* ctx) { return ctx ? ctx->err : "null context"; }

int segvlad_set_stream(segvlad_ctx* ctx, void* hip_stream) {
  if (!ctx) return SEGVLAD_ERR_ARG;
  ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
  return SEGVLAD_OK;
}

int segvlad_synchronize(segvlad_ctx* ctx) {
  CHECK_CTX();
  SV_HIP(hipStreamSynchronize(ctx->stream));
  return sv_guard_check(ctx);   // (guard mode only: SEGVLAD_ERR_STATE if a buffer's fence has been written)
}

int segvlad_set_profiling(segvlad_ctx* ctx, int on) {
  if (!ctx) return SEGVLAD_ERR_ARG;
  ctx->profiling = on != 0;
  return SEGVLAD_OK;
}

int segvlad_profile_reset(segvlad_ctx* ctx) {
  if (!ctx) return SEGVLAD_ERR_ARG;
  for (auto& kv : ctx->timers) {
    kv.second.used = 0;
    kv.second.launches = 0;
  }
  return SEGVLAD_OK;
}

int segvlad_stage_ms(segvlad_ctx* ctx, const char* stage, float* ms_out, int* launches_out) {
  CHECK_CTX();
  if (!stage || !ms_out) return ctx->fail(SEGVLAD_ERR_ARG, "stage_ms: null argument");
  auto it = ctx->timers.find(stage);
  if (it == ctx->timers.end() || it->second.used == 0)
    return ctx->fail(SEGVLAD_ERR_STATE, "stage '%s' has not run with profiling on", stage);
  StageTimer& t = it->second;
  double total = 0.0;
  for (int i = 0; i < t.used; ++i) {
    SV_HIP(hipEventSynchronize(t.ev[2 * i + 1]));
    float ms = 0.f;
    SV_HIP(hipEventElapsedTime(&ms, t.ev[2 * i], t.ev[2 * i + 1]));
    total += ms;
  }
  *ms_out = (float)total;
  if (launches_out) *launches_out = t.launches;
  return SEGVLAD_OK;
}

// ---- vocabulary ----------------------------------------------------------------------------------
int segvlad_set_vocab(segvlad_ctx* ctx, const float* C, int K, int D) {
  CHECK_CTX();
  if (!C || K <= 0 || D <= 0) return ctx->fail(SEGVLAD_ERR_ARG, "set_vocab: need C, K>0, D>0");
  if (D % 4) return ctx->fail(SEGVLAD_ERR_ARG, "set_vocab: D=%d must be a multiple of 4", D);
  if (K > 128) return ctx->fail(SEGVLAD_ERR_LIMIT, "set_vocab: K=%d > 128 clusters is not supported by this build", K);
  int Kpad = (K <= 32) ? 32 : (K <= 64) ? 64 : 128;
  const size_t bytes = (size_t)K * D * sizeof(float);
  SV_HIP(ctx->vocab.reserve(bytes));
  if (sv_is_device_ptr(C))
    SV_HIP(hipMemcpyAsync(ctx->vocab.p, C, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  else
    SV_HIP(hipMemcpyAsync(ctx->vocab.p, C, bytes, hipMemcpyHostToDevice, ctx->stream));
  ctx->K = K;
  ctx->D = D;
  ctx->Kpad = Kpad;
  SV_TRY(sv_launch_vocab_prepare(ctx));
  // largest centre component: bounds the token residuals x^ - C_k whose fp16 planes the "project" form builds (images_impl)
  SV_TRY(sv_maxabs(ctx, ctx->vocab.as<float>(), (int64_t)K * D, &ctx->vocab_maxabs));
  // largest centre norm: ||x^ - C_k|| <= 1 + max ||C_k||, the bound behind the fp16 split of the PROJECTED residuals (the P-space
  // sums of the "project" form on the 16-bit pipe)
  {
    float n2 = 0.f;
    SV_HIP(ctx->s_qnorm.reserve((size_t)K * sizeof(float)));
    SV_TRY(sv_launch_row_sumsq(ctx, ctx->vocab.as<float>(), K, D, ctx->s_qnorm.as<float>()));
    SV_TRY(sv_row_norm_max(ctx, ctx->s_qnorm.as<float>(), K, &n2));
    ctx->vocab_norm_max = std::sqrt(n2 > 0.f ? n2 : 0.f);
  }
  return sv_finish(ctx);
}

// ---- incidence / centroids -------------------------------------------------------------------------
static int incidence_impl(segvlad_ctx* ctx, const uint8_t* masks, int S, int Hm, int Wm, int H, int W, int patch,
                          uint64_t* inc_bits, double* centroids, bool want_centroids) {
  if (S < 0 || Hm <= 0 || Wm <= 0 || H <= 0 || W <= 0 || patch <= 0 || H / patch <= 0 || W / patch <= 0)
    return ctx->fail(SEGVLAD_ERR_ARG, "incidence: bad geometry S=%d masks %dx%d image %dx%d patch %d", S, Hm, Wm, H, W, patch);
  if (S == 0) return SEGVLAD_OK;
  if (!masks || !inc_bits || (want_centroids && !centroids)) return ctx->fail(SEGVLAD_ERR_ARG, "incidence: null pointer");
  const int N = (H / patch) * (W / patch), nw = (N + 63) / 64;
  const void* dm;
  void *dout, *dcent = nullptr;
  SV_TRY(sv_in(ctx, masks, (size_t)S * Hm * Wm, &dm));
  SV_TRY(sv_out(ctx, inc_bits, (size_t)S * nw * 8, &dout));
  if (want_centroids) SV_TRY(sv_out(ctx, centroids, (size_t)S * 2 * sizeof(double), &dcent));
  {
    StageScope sc(ctx, "incidence");
    SV_TRY(sv_launch_incidence(ctx, (const uint8_t*)dm, S, Hm, Wm, H, W, patch, (uint64_t*)dout, (double*)dcent));
    sc.count();
  }
  return sv_finish(ctx);
}

int segvlad_incidence(segvlad_ctx* ctx, const uint8_t* masks, int S, int Hm, int Wm, int H, int W, int patch,
                      uint64_t* inc_bits) {
  CHECK_CTX();
  CHECK_NO_OPEN_DESCRIBE("incidence");
  return incidence_impl(ctx, masks, S, Hm, Wm, H, W, patch, inc_bits, nullptr, false);
}

int segvlad_incidence_centroids(segvlad_ctx* ctx, const uint8_t* masks, int S, int Hm, int Wm, int H, int W, int patch,
                                uint64_t* inc_bits, double* centroids) {
  CHECK_CTX();
  CHECK_NO_OPEN_DESCRIBE("incidence_centroids");
  return incidence_impl(ctx, masks, S, Hm, Wm, H, W, patch, inc_bits, centroids, true);
}

int segvlad_mask_centroids(segvlad_ctx* ctx, const uint8_t* masks, int S, int Hm, int Wm, double* centroids) {
  CHECK_CTX();
  if (S < 0 || Hm <= 0 || Wm <= 0) return ctx->fail(SEGVLAD_ERR_ARG, "mask_centroids: bad geometry");
  if (S == 0) return SEGVLAD_OK;
  if (!masks || !centroids) return ctx->fail(SEGVLAD_ERR_ARG, "mask_centroids: null pointer");
  const void* dm;
  void* dout;
  SV_TRY(sv_in(ctx, masks, (size_t)S * Hm * Wm, &dm));
  SV_TRY(sv_out(ctx, centroids, (size_t)S * 2 * sizeof(double), &dout));
  SV_TRY(sv_launch_centroids(ctx, (const uint8_t*)dm, S, Hm, Wm, (double*)dout));
  return sv_finish(ctx);
}

static int adjacency_impl(segvlad_ctx* ctx, const double* centroids, const int32_t* seg_offsets, int B, int order,
                          uint8_t* adj_out, uint32_t* n_empty_out, uint8_t* img_flags_out) {
  if (B < 0 || order < 1) return ctx->fail(SEGVLAD_ERR_ARG, "adjacency: need B>=0 and order>=1 (order 0 = pass adj=NULL)");
  if (B == 0) return SEGVLAD_OK;
  if (!centroids || !seg_offsets || !adj_out) return ctx->fail(SEGVLAD_ERR_ARG, "adjacency: null pointer");
  if (sv_is_device_ptr(seg_offsets)) return ctx->fail(SEGVLAD_ERR_ARG, "adjacency: seg_offsets must be host memory");
  int S_max = 0;
  std::vector<int64_t> adj_off(B + 1, 0);
  for (int b = 0; b < B; ++b) {
    const int s = seg_offsets[b + 1] - seg_offsets[b];
    if (s < 0) return ctx->fail(SEGVLAD_ERR_ARG, "adjacency: seg_offsets must be non-decreasing");
    if (s > S_max) S_max = s;
    adj_off[b + 1] = adj_off[b] + (int64_t)s * s;
  }
  const int S_tot = seg_offsets[B];
  if (S_tot == 0) return SEGVLAD_OK;
  const void* dc;
  void* dout;
  SV_TRY(sv_in(ctx, centroids, (size_t)S_tot * 2 * sizeof(double), &dc));
  SV_TRY(sv_out(ctx, adj_out, (size_t)adj_off[B], &dout));
  SV_HIP(ctx->s_segoff.reserve((size_t)(B + 1) * sizeof(int32_t)));
  SV_HIP(ctx->s_adjoff.reserve((size_t)(B + 1) * sizeof(int64_t)));
  SV_HIP(ctx->s_flag.reserve(64));
  SV_HIP(hipMemcpyAsync(ctx->s_segoff.p, seg_offsets, (size_t)(B + 1) * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  SV_HIP(hipMemcpyAsync(ctx->s_adjoff.p, adj_off.data(), (size_t)(B + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  uint32_t* bad = ctx->s_flag.as<uint32_t>() + 4;
  SV_HIP(hipMemsetAsync(bad, 0, 4, ctx->stream));
  void* dflags = nullptr;
  if (img_flags_out) {
    SV_TRY(sv_out(ctx, img_flags_out, (size_t)B, &dflags));
    SV_HIP(hipMemsetAsync(dflags, 0, (size_t)B, ctx->stream));
  }
  {
    StageScope sc(ctx, "adjacency");
    SV_TRY(sv_launch_adjacency(ctx, (const double*)dc, ctx->s_segoff.as<int32_t>(), ctx->s_adjoff.as<int64_t>(), B, S_max, order,
                               (uint8_t*)dout, bad, (uint8_t*)dflags));
    sc.count();
  }
  if (n_empty_out) {
    SV_HIP(hipMemcpyAsync(n_empty_out, bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    SV_HIP(hipStreamSynchronize(ctx->stream));
  }
  return sv_finish(ctx);
}

int segvlad_adjacency(segvlad_ctx* ctx, const double* centroids, const int32_t* seg_offsets, int B, int order,
                      uint8_t* adj_out, uint32_t* n_empty_out) {
  CHECK_CTX();
  CHECK_NO_OPEN_DESCRIBE("adjacency");
  return adjacency_impl(ctx, centroids, seg_offsets, B, order, adj_out, n_empty_out, nullptr);
}

int segvlad_adjacency_flagged(segvlad_ctx* ctx, const double* centroids, const int32_t* seg_offsets, int B, int order,
                              uint8_t* adj_out, uint8_t* img_flags_out) {
  CHECK_CTX();
  CHECK_NO_OPEN_DESCRIBE("adjacency_flagged");
  if (B > 0 && !img_flags_out) return ctx->fail(SEGVLAD_ERR_ARG, "adjacency_flagged: null img_flags_out");
  return adjacency_impl(ctx, centroids, seg_offsets, B, order, adj_out, nullptr, img_flags_out);
}

// ---- segment VLAD -----------------------------------------------------------------------------------
// pca_y != NULL: fused projection (segvlad_images_pca); out may then be NULL
// The assignment pass on its own (segvlad_describe_begin): its outputs -- token-major copy, norms, labels -- stay in the
// context's scratch for images_impl(phase = 2).
static int assign_phase(segvlad_ctx* ctx, const float* tokens, int B, int N) {
  if (ctx->K == 0) return ctx->fail(SEGVLAD_ERR_STATE, "images: call segvlad_set_vocab first");
  if (B <= 0 || N <= 0 || !tokens) return ctx->fail(SEGVLAD_ERR_ARG, "images: B=%d N=%d", B, N);
  const int D = ctx->D;
  const void* d_tok;
  SV_TRY(sv_in(ctx, tokens, (size_t)B * D * N * sizeof(float), &d_tok));
  SV_HIP(ctx->s_xt.reserve((size_t)B * N * D * sizeof(float)));
  SV_HIP(ctx->s_rnorm.reserve((size_t)B * N * sizeof(float)));
  SV_HIP(ctx->s_labels.reserve((size_t)B * N));
  StageScope sc(ctx, "assign");
  SV_TRY(sv_launch_assign(ctx, (const float*)d_tok, B, N, ctx->s_xt.as<float>(), ctx->s_labels.as<uint8_t>(), ctx->s_rnorm.as<float>(), nullptr));
  sc.count();
  return SEGVLAD_OK;
}

// phase 0: everything.  2: from `prep` on, the assignment pass having run in an earlier call (assign_phase, same tokens) -- its
// outputs live in the context's scratch, every reservation below is idempotent.
static int images_impl(segvlad_ctx* ctx, const float* tokens, int B, int N, const uint64_t* inc_bits,
                       const int32_t* seg_offsets, const uint8_t* adj, float* out, uint8_t* labels_out, float* gap_out,
                       float* block_norms_out, float* pca_y, int l2norm, int phase = 0) {
  if (ctx->K == 0) return ctx->fail(SEGVLAD_ERR_STATE, "images: call segvlad_set_vocab first");
  if (B < 0 || N <= 0) return ctx->fail(SEGVLAD_ERR_ARG, "images: B=%d N=%d", B, N);
  if (B == 0) return SEGVLAD_OK;
  if (!tokens || !seg_offsets) return ctx->fail(SEGVLAD_ERR_ARG, "images: null pointer");
  if (sv_is_device_ptr(seg_offsets)) return ctx->fail(SEGVLAD_ERR_ARG, "images: seg_offsets must be host memory");
  const int K = ctx->K, D = ctx->D;
  const int nw = (N + 63) / 64;
  int S_max = 0;
  std::vector<int64_t> adj_off(B + 1, 0);
  if (seg_offsets[0] != 0) return ctx->fail(SEGVLAD_ERR_ARG, "images: seg_offsets[0] must be 0");
  for (int b = 0; b < B; ++b) {
    const int s = seg_offsets[b + 1] - seg_offsets[b];
    if (s < 0) return ctx->fail(SEGVLAD_ERR_ARG, "images: seg_offsets must be non-decreasing");
    if (s > S_max) S_max = s;
    adj_off[b + 1] = adj_off[b] + (int64_t)s * s;
  }
  const int S_tot = seg_offsets[B];
  if (S_tot > 0 && (!inc_bits || (!out && !pca_y))) return ctx->fail(SEGVLAD_ERR_ARG, "images: null inc_bits/out");
  const int SC = S_max > 0 ? (S_max + 63) / 64 : 1;
  // fused projection: the descriptor entries are <= 1 in magnitude by construction, so the fp16 split scale is known
  // before the data exists (pca_apply has to measure max|x| first)
  const bool fused = pca_y != nullptr && S_tot > 0;
  float xscale = 1.f;
  void* d_y = nullptr;
  // "project then aggregate" (project_kernels.hip): when the descriptor itself is not asked for, project every token with
  // its cluster's slice of the components and aggregate the segments in the P-dimensional space
  // (auto: when it is the smaller product -- N D P per image against S K D P: 2.1 x fewer flops at S = 50, K = 64, N = 1530
  //  -- and the batch holds >= 128 tokens per cluster on average: every cluster's rows are padded to whole 256-row GEMM
  //  tiles, and below that the split-K descriptor GEMM is quicker -- 1 image: 0.62 against 0.84 ms, 16 images: 1.45 against 1.02)
  const bool project = fused && out == nullptr && D % 32 == 0 && ctx->P % 4 == 0 && ctx->opt.pca_path != 1 &&
                       (ctx->opt.pca_path == 2 ||
                        ((double)S_tot * K >= 1.25 * (double)B * N && (double)B * N >= 128.0 * K));
  int64_t rows_pad = 0;   // grouped token rows: every cluster's rows padded to whole 256-row GEMM tiles
  if (fused && project) {
    // residuals of unit tokens against the centres: |r| <= 1 + max|C|.  Centres that are means of unit tokens (every
    // k-means vocabulary) give 2^13, like the planes form's data-derived scale; arbitrary centres (segvlad_set_vocab takes
    // any) get the scale their magnitude needs instead of overflowing fp16 silently
    {
      int e;
      const float bound = 1.f + ctx->vocab_maxabs;
      if (!std::isfinite(bound)) return ctx->fail(SEGVLAD_ERR_ARG, "images_pca: the vocabulary holds a non-finite centre");
      frexpf(bound, &e);
      xscale = ldexpf(1.f, 14 - e);
    }
    rows_pad = (((int64_t)B * N + 255) & ~255ll) + 256ll * K;
    SV_HIP(ctx->s_xh1.reserve((size_t)(rows_pad + 256) * D * 2));   // + one tile: the dummy row of sv_launch_token_norms
    SV_HIP(ctx->s_xh2.reserve((size_t)(rows_pad + 256) * D * 2));
    SV_HIP(ctx->s_pz.reserve((size_t)rows_pad * ctx->P * sizeof(float)));
    SV_HIP(ctx->s_rowbase.reserve((size_t)B * K * sizeof(int32_t)));
    SV_HIP(ctx->s_tilegrp.reserve((size_t)(rows_pad >> 8) * sizeof(int32_t)));
    SV_HIP(ctx->s_bn.reserve((size_t)S_tot * K * sizeof(float)));
    SV_TRY(sv_out(ctx, pca_y, (size_t)S_tot * ctx->P * sizeof(float), &d_y));
    if (!ctx->pca_cproj_valid) {
      SV_HIP(ctx->pca_cproj.reserve((size_t)ctx->P * sizeof(float)));
      SV_TRY(sv_launch_project_consts(ctx, ctx->pca_comps.as<float>(), ctx->pca_mean.as<float>(), ctx->P, ctx->KD,
                                      ctx->pca_cproj.as<float>()));
      ctx->pca_cproj_valid = true;
    }
  } else if (fused) {
    int e;
    frexpf(1.f + ctx->pca_mean_maxabs, &e);
    xscale = ldexpf(1.f, 14 - e);
    SV_HIP(ctx->s_xh1.reserve((size_t)sv_x3_rows(S_tot) * ctx->KD * 2));   // blocked planes, rows padded to whole tiles
    SV_HIP(ctx->s_xh2.reserve((size_t)sv_x3_rows(S_tot) * ctx->KD * 2));
    SV_TRY(sv_out(ctx, pca_y, (size_t)S_tot * ctx->P * sizeof(float), &d_y));
  }

  const void *d_tok, *d_inc = nullptr, *d_adj = nullptr;
  void *d_out = nullptr, *d_lab = nullptr, *d_gap = nullptr, *d_bn = nullptr;
  SV_TRY(sv_in(ctx, tokens, (size_t)B * D * N * sizeof(float), &d_tok));
  if (S_tot > 0) SV_TRY(sv_in(ctx, inc_bits, (size_t)S_tot * nw * 8, &d_inc));
  if (adj && S_tot > 0) SV_TRY(sv_in(ctx, adj, (size_t)adj_off[B], &d_adj));
  if (S_tot > 0 && out) SV_TRY(sv_out(ctx, out, (size_t)S_tot * K * D * sizeof(float), &d_out));
  if (labels_out) SV_TRY(sv_out(ctx, labels_out, (size_t)B * N, &d_lab));
  if (gap_out) SV_TRY(sv_out(ctx, gap_out, (size_t)B * N * sizeof(float), &d_gap));
  if (block_norms_out && S_tot > 0) SV_TRY(sv_out(ctx, block_norms_out, (size_t)S_tot * K * sizeof(float), &d_bn));

  SV_HIP(ctx->s_xt.reserve((size_t)B * N * D * sizeof(float)));
  SV_HIP(ctx->s_rnorm.reserve((size_t)B * N * sizeof(float)));
  if (!d_lab) {
    SV_HIP(ctx->s_labels.reserve((size_t)B * N));
    d_lab = ctx->s_labels.p;
  }
  SV_HIP(ctx->s_colmask.reserve((size_t)B * N * SC * 8));
  SV_HIP(ctx->s_gscale.reserve((size_t)(S_tot + 1) * sizeof(float)));
  SV_HIP(ctx->s_segoff.reserve((size_t)(B + 1) * sizeof(int32_t)));
  SV_HIP(ctx->s_adjoff.reserve((size_t)(B + 1) * sizeof(int64_t)));
  if (!ctx->mask_branch_on_side) {   // (segvlad_describe: the adjacency call on the side stream has put both in place)
    SV_HIP(hipMemcpyAsync(ctx->s_segoff.p, seg_offsets, (size_t)(B + 1) * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    SV_HIP(hipMemcpyAsync(ctx->s_adjoff.p, adj_off.data(), (size_t)(B + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  }
  // (pageable sources: the runtime stages them before hipMemcpyAsync returns, so the local vector
  //  and the caller's array may be reused as soon as this call returns)

  if (phase != 2) {
    StageScope sc(ctx, "assign");
    SV_TRY(sv_launch_assign(ctx, (const float*)d_tok, B, N, ctx->s_xt.as<float>(), (uint8_t*)d_lab, ctx->s_rnorm.as<float>(),
                            (float*)d_gap));
    sc.count();
  }
  if (ctx->mask_branch_on_side) {   // segvlad_describe: incidence + centroids + adjacency ran beside the assignment pass
    ctx->mask_branch_on_side = false;
    SV_TRY(sv_join_side(ctx));
  }
  if (S_tot > 0) {
    {
      StageScope sc(ctx, "prep");
      SV_TRY(sv_launch_prep(ctx, (const uint8_t*)d_lab, (const uint64_t*)d_inc, ctx->s_segoff.as<int32_t>(),
                            ctx->s_adjoff.as<int64_t>(), (const uint8_t*)d_adj, B, N, K, S_max, SC,
                            ctx->s_colmask.as<uint64_t>(), ctx->s_gscale.as<float>()));
      sc.count();
    }
    if (project) {
      float* bn = d_bn ? (float*)d_bn : ctx->s_bn.as<float>();
      {
        StageScope sc(ctx, "aggregate");   // block norms + the normalised tokens' fp16 planes, grouped by cluster
        SV_TRY(sv_launch_group_plan(ctx, ctx->s_laboff.as<int32_t>(), B, K, ctx->s_rowbase.as<int32_t>(),
                                    ctx->s_tilegrp.as<int32_t>(), (int)(rows_pad >> 8)));
        SV_TRY(sv_launch_token_norms(ctx, ctx->s_xt.as<float>(), ctx->s_colmask.as<uint64_t>(), ctx->vocab.as<float>(), K, D,
                                     ctx->s_segoff.as<int32_t>(), B, N, SC, bn, xscale, ctx->s_xh1.as<uint16_t>(),
                                     ctx->s_xh2.as<uint16_t>(), ctx->s_rowbase.as<int32_t>(), ctx->opt.debug_search == 7 ? -rows_pad : rows_pad));
        sc.count(2);
      }
      StageScope sc(ctx, "pca");
      SV_TRY(sv_launch_gemm_f16x3_grouped(ctx, ctx->s_xh1.as<uint16_t>(), ctx->s_xh2.as<uint16_t>(), ctx->pca_w1.as<uint16_t>(),
                                          ctx->pca_w2.as<uint16_t>(), (int)rows_pad, ctx->P, D, K, ctx->s_tilegrp.as<int32_t>(),
                                          1.f / (xscale * ctx->pca_w_scale), ctx->s_pz.as<float>()));
      // |z_tp| = |W_k[p, :] . r_t| <= sqrt(D) max|W| (1 + max ||C_k||): a power-of-two scale that keeps z * zscale inside fp16.
      // Both premises hold BY CONSTRUCTION, not by the caller's grace: r_t = x^_t - C_k with x^_t normalised by the kernels
      // themselves (||x^_t|| = 1 whatever the caller's tokens are), and vocab_norm_max is set by the one function that can change the
      // centres (segvlad_set_vocab); a bound that is not finite and positive switches the 16-bit form off (zscale = 0: fp32 MFMA)
      // (the elementwise bound of W is the one pca_set left in pca_w_scale; typical |z| sits ~2^-7 below the bound, still
      // 2^10 above the point where the low half of the split would go subnormal)
      float zscale = 0.f;
      if (ctx->pca_w_scale > 0.f) {
        const float zb = std::sqrt((float)D) * (16384.f / ctx->pca_w_scale) * (1.f + ctx->vocab_norm_max);
        if (std::isfinite(zb) && zb > 0.f) {
          int e;
          frexpf(zb, &e);
          zscale = ldexpf(1.f, 14 - e);
        }
      }
      SV_TRY(sv_launch_project_aggregate(ctx, ctx->s_pz.as<float>(), ctx->pca_cproj.as<float>(), bn, ctx->s_gscale.as<float>(),
                                         ctx->s_colmask.as<uint64_t>(), ctx->s_laboff.as<int32_t>(), ctx->s_rowbase.as<int32_t>(),
                                         ctx->s_segoff.as<int32_t>(), B, N, K, ctx->P, SC, S_max, ctx->pca_scale.as<float>(),
                                         (float*)d_y, zscale));
      sc.count(2);
      if (l2norm) {
        SV_TRY(sv_launch_normalize_rows(ctx, (const float*)d_y, S_tot, ctx->P, (float*)d_y));
        sc.count();
      }
    } else {
    {
      StageScope sc(ctx, "aggregate");
      SV_TRY(sv_launch_aggregate(ctx, ctx->s_xt.as<float>(), ctx->s_rnorm.as<float>(), (const uint8_t*)d_lab,
                                 ctx->s_colmask.as<uint64_t>(), ctx->vocab.as<float>(), K, D, ctx->s_segoff.as<int32_t>(),
                                 ctx->s_gscale.as<float>(), B, N, SC, (float*)d_out, (float*)d_bn,
                                 fused ? ctx->pca_mean.as<float>() : nullptr, xscale,
                                 fused ? ctx->s_xh1.as<uint16_t>() : nullptr, fused ? ctx->s_xh2.as<uint16_t>() : nullptr));
      sc.count();
    }
    if (fused) {
      StageScope sc(ctx, "pca");
      SV_TRY(sv_launch_gemm_f16x3(ctx, ctx->s_xh1.as<uint16_t>(), ctx->s_xh2.as<uint16_t>(), ctx->pca_w1.as<uint16_t>(),
                                  ctx->pca_w2.as<uint16_t>(), S_tot, ctx->P, ctx->KD, 1.f / (xscale * ctx->pca_w_scale),
                                  ctx->pca_scale.as<float>(), (float*)d_y));
      sc.count(2);
      if (l2norm) {
        SV_TRY(sv_launch_normalize_rows(ctx, (const float*)d_y, S_tot, ctx->P, (float*)d_y));
        sc.count();
      }
    }
    }
  }
  return sv_finish(ctx);
}

int segvlad_images(segvlad_ctx* ctx, const float* tokens, int B, int N, const uint64_t* inc_bits,
                   const int32_t* seg_offsets, const uint8_t* adj, float* out, uint8_t* labels_out, float* gap_out,
                   float* block_norms_out) {
  CHECK_CTX();
  CHECK_NO_OPEN_DESCRIBE("images");
  return images_impl(ctx, tokens, B, N, inc_bits, seg_offsets, adj, out, labels_out, gap_out, block_norms_out, nullptr, 0);
}

static int images_pca_impl(segvlad_ctx* ctx, const float* tokens, int B, int N, const uint64_t* inc_bits,
                           const int32_t* seg_offsets, const uint8_t* adj, float* y, int l2norm, float* desc_out,
                           uint8_t* labels_out, float* gap_out, int phase = 0) {
  if (ctx->P == 0) return ctx->fail(SEGVLAD_ERR_STATE, "images_pca: call segvlad_pca_set first");
  if (ctx->K == 0) return ctx->fail(SEGVLAD_ERR_STATE, "images_pca: call segvlad_set_vocab first");
  if (ctx->KD != ctx->K * ctx->D)
    return ctx->fail(SEGVLAD_ERR_ARG, "images_pca: the PCA model expects %d-d rows, the vocabulary gives %d", ctx->KD,
                     ctx->K * ctx->D);
  if (B > 0 && seg_offsets && !sv_is_device_ptr(seg_offsets) && seg_offsets[B] > 0 && !y)
    return ctx->fail(SEGVLAD_ERR_ARG, "images_pca: null y");
  const bool x3 = ctx->pca_w_scale > 0.f && !ctx->opt.pca_fp32;
  if (x3) return images_impl(ctx, tokens, B, N, inc_bits, seg_offsets, adj, desc_out, labels_out, gap_out, nullptr, y, l2norm, phase);
  if (phase != 0) return ctx->fail(SEGVLAD_ERR_STATE, "describe: the split call needs the fp16x3 projection (pca_arith)");
  // shapes the split GEMM does not take (or the fp32 knob): descriptor to HBM, then the plain projection
  if (B <= 0 || !seg_offsets || sv_is_device_ptr(seg_offsets))
    return images_impl(ctx, tokens, B, N, inc_bits, seg_offsets, adj, desc_out, labels_out, gap_out, nullptr, nullptr, 0);
  const int S_tot = seg_offsets[B];
  float* desc = desc_out;
  if (!desc && S_tot > 0) {
    SV_HIP(ctx->s_desc.reserve((size_t)S_tot * ctx->KD * sizeof(float)));
    desc = ctx->s_desc.as<float>();
  }
  SV_TRY(images_impl(ctx, tokens, B, N, inc_bits, seg_offsets, adj, desc, labels_out, gap_out, nullptr, nullptr, 0));
  return S_tot > 0 ? segvlad_pca_apply(ctx, desc, S_tot, y, l2norm) : SEGVLAD_OK;
}

int segvlad_images_pca(segvlad_ctx* ctx, const float* tokens, int B, int N, const uint64_t* inc_bits,
                       const int32_t* seg_offsets, const uint8_t* adj, float* y, int l2norm, float* desc_out,
                       uint8_t* labels_out, float* gap_out) {
  CHECK_CTX();
  CHECK_NO_OPEN_DESCRIBE("images_pca");
  return images_pca_impl(ctx, tokens, B, N, inc_bits, seg_offsets, adj, y, l2norm, desc_out, labels_out, gap_out);
}

// ---- the whole describe stage of a batch in ONE call: masks + tokens -> (projected) segment descriptors ----------------------
// place_rec_main.py:244-270 per image (masks -> adjacency -> seg_vlad_gpu_single) and per batch (apply_pca_transform_from_pkl).
// The mask branch (incidence + centroids -> adjacency: two latency-bound launches, 0.34 ms per 200 images) does not depend on
// the tokens, the assignment pass (0.95 ms, HBM-bound) does not depend on the masks: the former runs on the context's side
// stream BESIDE the latter and is joined in front of `prep`, the first kernel that needs both.
static int describe_begin_impl(segvlad_ctx* ctx, const uint8_t* masks, int Hm, int Wm, int H, int W, int patch, const float* tokens, int B,
                               int N, const int32_t* seg_offsets, int order, uint64_t* inc_bits_out, double* centroids_out,
                               uint8_t* adj_out, uint8_t* img_flags_out, bool pca) {
  if (ctx->mask_branch_on_side) return ctx->fail(SEGVLAD_ERR_STATE, "describe_begin: the previous describe_begin has no describe_end yet");
  if (B < 0 || order < 1) return ctx->fail(SEGVLAD_ERR_ARG, "describe: need B >= 0 and order >= 1");
  if (!masks || !tokens || !seg_offsets || !inc_bits_out || !centroids_out || !adj_out || !img_flags_out)
    return ctx->fail(SEGVLAD_ERR_ARG, "describe: null pointer");
  if (sv_is_device_ptr(seg_offsets)) return ctx->fail(SEGVLAD_ERR_ARG, "describe: seg_offsets must be host memory");
  const void* bulk[] = {masks, tokens, inc_bits_out, centroids_out, adj_out, img_flags_out};
  for (const void* p : bulk)
    if (!sv_is_device_ptr(p))
      return ctx->fail(SEGVLAD_ERR_ARG, "describe: bulk pointers must be device memory (the separate entry points stage host data)");
  if (patch <= 0 || N != (H / patch) * (W / patch))
    return ctx->fail(SEGVLAD_ERR_ARG, "describe: N=%d tokens do not match the %dx%d image at patch %d", N, H, W, patch);
  const int S_tot = B > 0 ? seg_offsets[B] : 0;
  // Configurations the split call does not take are reported as SEGVLAD_ERR_LIMIT -- the code of "this entry point cannot, the
  // separate ones (incidence_centroids -> adjacency -> images[_pca]) can": an empty batch, and a PCA model without the fp16x3
  // form (option pca_arith=fp32, or K*D not a multiple of 32: images_pca then writes the descriptor and projects it with the
  // plain GEMM).  ADVICE r05: both used to come back as ERR_ARG / ERR_STATE, which a caller cannot tell from misuse.
  if (B == 0 || S_tot <= 0) return ctx->fail(SEGVLAD_ERR_LIMIT, "describe: no images / no segments (use the separate entry points)");
  if (pca && ctx->P == 0) return ctx->fail(SEGVLAD_ERR_STATE, "describe: call segvlad_pca_set first");
  if (pca && (ctx->pca_w_scale <= 0.f || ctx->opt.pca_fp32))
    return ctx->fail(SEGVLAD_ERR_LIMIT, "describe: the split call needs the fp16x3 projection (pca_arith; K*D %% 32 == 0): use the separate entry points");
  // pinned landing zone of the flags and the centroids (read by segvlad_describe_flags while the assignment pass runs)
  const size_t need = (size_t)B + (size_t)S_tot * 16 + 64;
  if (need > ctx->h_desc_cap) {
    if (ctx->h_desc) SV_HIP(hipHostFree(ctx->h_desc));
    ctx->h_desc = nullptr;
    ctx->h_desc_cap = 0;
    SV_HIP(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_desc), need + need / 2, hipHostMallocDefault));
    ctx->h_desc_cap = need + need / 2;
  }
  if (!ctx->ev_desc) SV_HIP(hipEventCreateWithFlags(&ctx->ev_desc, hipEventDisableTiming));
  // stage "describe": from here to the end of segvlad_describe_end, as the context's stream sees it (the parts overlap)
  ctx->desc_timer = nullptr;
  if (ctx->profiling && !ctx->scope_mute) {
    StageTimer* t = &ctx->timers["describe"];
    if (t->used * 2 >= (int)t->ev.size()) {
      hipEvent_t a = nullptr, b = nullptr;
      (void)hipEventCreate(&a);
      (void)hipEventCreate(&b);
      t->ev.push_back(a);
      t->ev.push_back(b);
    }
    ctx->desc_slot = t->used++;
    ctx->desc_timer = t;
    (void)hipEventRecord(t->ev[2 * ctx->desc_slot], ctx->stream);
  }
  // ---- mask branch on the side stream: the same two calls as the separate entry points, issued there -----------------------
  SV_TRY(sv_fork_side(ctx));
  hipStream_t main_stream = ctx->stream;
  ctx->stream = ctx->side;
  int rc = incidence_impl(ctx, masks, S_tot, Hm, Wm, H, W, patch, inc_bits_out, centroids_out, true);
  if (rc == SEGVLAD_OK) rc = adjacency_impl(ctx, centroids_out, seg_offsets, B, order, adj_out, nullptr, img_flags_out);
  if (rc == SEGVLAD_OK) {
    hipError_t e = hipMemcpyAsync(ctx->h_desc, img_flags_out, (size_t)B, hipMemcpyDeviceToHost, ctx->side);
    if (e == hipSuccess)
      e = hipMemcpyAsync(ctx->h_desc + (((size_t)B + 63) & ~(size_t)63), centroids_out, (size_t)S_tot * 16, hipMemcpyDeviceToHost, ctx->side);
    if (e == hipSuccess) e = hipEventRecord(ctx->ev_desc, ctx->side);
    if (e != hipSuccess) rc = ctx->fail(SEGVLAD_ERR_HIP, "describe: flags read-back: %s", hipGetErrorString(e));
  }
  ctx->stream = main_stream;
  auto close_timer = [&]() {
    if (ctx->desc_timer) (void)hipEventRecord(ctx->desc_timer->ev[2 * ctx->desc_slot + 1], ctx->stream);
    ctx->desc_timer = nullptr;
  };
  if (rc != SEGVLAD_OK) {
    (void)sv_join_side(ctx);   // (nothing of the branch stays behind on the side stream unobserved)
    close_timer();
    return rc;
  }
  ctx->mask_branch_on_side = true;   // images_impl (phase 2 / 0) joins in front of prep
  ctx->desc_B = B;
  ctx->desc_S = S_tot;
  // ---- main stream: the assignment pass -------------------------------------------------------------------------------------------
  rc = assign_phase(ctx, tokens, B, N);
  if (rc != SEGVLAD_OK) {
    ctx->mask_branch_on_side = false;
    (void)sv_join_side(ctx);
    close_timer();
  }
  return rc;
}

int segvlad_describe_begin(segvlad_ctx* ctx, const uint8_t* masks, int Hm, int Wm, int H, int W, int patch, const float* tokens, int B,
                           int N, const int32_t* seg_offsets, int order, uint64_t* inc_bits_out, double* centroids_out, uint8_t* adj_out,
                           uint8_t* img_flags_out, int pca) {
  CHECK_CTX();
  return describe_begin_impl(ctx, masks, Hm, Wm, H, W, patch, tokens, B, N, seg_offsets, order, inc_bits_out, centroids_out, adj_out,
                             img_flags_out, pca != 0);
}

int segvlad_describe_flags(segvlad_ctx* ctx, uint8_t* flags_host, double* centroids_host) {
  CHECK_CTX();
  if (!ctx->mask_branch_on_side) return ctx->fail(SEGVLAD_ERR_STATE, "describe_flags: call segvlad_describe_begin first");
  if (!flags_host) return ctx->fail(SEGVLAD_ERR_ARG, "describe_flags: null pointer");
  SV_HIP(hipEventSynchronize(ctx->ev_desc));   // the MASK BRANCH only: the assignment pass on the main stream keeps running
  memcpy(flags_host, ctx->h_desc, (size_t)ctx->desc_B);
  if (centroids_host) memcpy(centroids_host, ctx->h_desc + (((size_t)ctx->desc_B + 63) & ~(size_t)63), (size_t)ctx->desc_S * 16);
  return SEGVLAD_OK;
}

int segvlad_describe_end(segvlad_ctx* ctx, const float* tokens, int B, int N, const uint64_t* inc_bits, const int32_t* seg_offsets,
                         uint8_t* adj, int n_patch, const int32_t* patch_images, const uint8_t* patch_blocks, float* desc_out, float* y,
                         int l2norm) {
  CHECK_CTX();
  if (!ctx->mask_branch_on_side) return ctx->fail(SEGVLAD_ERR_STATE, "describe_end: call segvlad_describe_begin first");
  auto bail = [&](int rc) {   // leaves the context usable: the mask branch is joined, the split call is over
    if (ctx->desc_timer) {
      (void)hipEventRecord(ctx->desc_timer->ev[2 * ctx->desc_slot + 1], ctx->stream);
      ctx->desc_timer = nullptr;
    }
    ctx->mask_branch_on_side = false;
    (void)sv_join_side(ctx);
    return rc;
  };
  if (!tokens || !inc_bits || !seg_offsets || !adj || (!desc_out && !y) || B != ctx->desc_B || seg_offsets[B] != ctx->desc_S ||
      n_patch < 0 || (n_patch > 0 && (!patch_images || !patch_blocks)))
    return bail(ctx->fail(SEGVLAD_ERR_ARG, "describe_end: arguments do not match segvlad_describe_begin's"));
  if (n_patch > 0) {
    // adjacency blocks the caller recomputed on the host (Qhull, for images whose centroids are in a non-generic configuration):
    // written over the device kernel's behind the mask branch, in front of prep
    if (sv_join_side(ctx) != SEGVLAD_OK) return bail(SEGVLAD_ERR_HIP);
    std::vector<int64_t> aoff((size_t)B + 1, 0);
    for (int b = 0; b < B; ++b) {
      const int64_t sb = seg_offsets[b + 1] - seg_offsets[b];
      aoff[(size_t)b + 1] = aoff[(size_t)b] + sb * sb;
    }
    size_t src = 0;
    for (int j = 0; j < n_patch; ++j) {
      const int b = patch_images[j];
      if (b < 0 || b >= B) return bail(ctx->fail(SEGVLAD_ERR_ARG, "describe_end: patch image %d out of range", b));
      const size_t bytes = (size_t)(aoff[(size_t)b + 1] - aoff[(size_t)b]);
      const hipError_t e = hipMemcpyAsync(adj + aoff[(size_t)b], patch_blocks + src, bytes, hipMemcpyHostToDevice, ctx->stream);
      if (e != hipSuccess) return bail(ctx->fail(SEGVLAD_ERR_HIP, "describe_end: adjacency patch: %s", hipGetErrorString(e)));
      src += bytes;
    }
  }
  const int rc = y ? images_pca_impl(ctx, tokens, B, N, inc_bits, seg_offsets, adj, y, l2norm, desc_out, nullptr, nullptr, 2)
                   : images_impl(ctx, tokens, B, N, inc_bits, seg_offsets, adj, desc_out, nullptr, nullptr, nullptr, nullptr, 0, 2);
  if (ctx->desc_timer) {
    (void)hipEventRecord(ctx->desc_timer->ev[2 * ctx->desc_slot + 1], ctx->stream);
    ctx->desc_timer->launches += 1;
    ctx->desc_timer = nullptr;
  }
  if (ctx->mask_branch_on_side) return bail(rc);   // an early return in front of the join
  return rc;
}

int segvlad_describe(segvlad_ctx* ctx, const uint8_t* masks, int Hm, int Wm, int H, int W, int patch, const float* tokens, int B, int N,
                     const int32_t* seg_offsets, int order, uint64_t* inc_bits_out, double* centroids_out, uint8_t* adj_out,
                     uint8_t* img_flags_out, float* desc_out, float* y, int l2norm) {
  CHECK_CTX();
  if (!desc_out && !y) return ctx->fail(SEGVLAD_ERR_ARG, "describe: null pointer");
  if ((desc_out && !sv_is_device_ptr(desc_out)) || (y && !sv_is_device_ptr(y)))
    return ctx->fail(SEGVLAD_ERR_ARG, "describe: bulk pointers must be device memory (the separate entry points stage host data)");
  SV_TRY(describe_begin_impl(ctx, masks, Hm, Wm, H, W, patch, tokens, B, N, seg_offsets, order, inc_bits_out, centroids_out, adj_out,
                             img_flags_out, y != nullptr));
  return segvlad_describe_end(ctx, tokens, B, N, inc_bits_out, seg_offsets, adj_out, 0, nullptr, nullptr, desc_out, y, l2norm);
}

// ---- vocabulary k-means: one Lloyd half-step over a batch of images (utilities.py:749-791, vlad_c_centers_pt_gen.py:86-158) -------
int segvlad_kmeans_step(segvlad_ctx* ctx, const float* tokens, int B, int N, double* sums, int64_t* counts, uint8_t* labels_out) {
  CHECK_CTX();
  CHECK_NO_OPEN_DESCRIBE("kmeans_step");
  if (ctx->K == 0) return ctx->fail(SEGVLAD_ERR_STATE, "kmeans_step: call segvlad_set_vocab with the current centres first");
  if (B < 0 || N <= 0) return ctx->fail(SEGVLAD_ERR_ARG, "kmeans_step: B=%d N=%d", B, N);
  if (B == 0) return SEGVLAD_OK;
  if (!tokens || !sums || !counts) return ctx->fail(SEGVLAD_ERR_ARG, "kmeans_step: null pointer");
  const int K = ctx->K, D = ctx->D;
  void *d_sums, *d_cnt, *d_lab = nullptr;
  if (!sv_is_device_ptr(sums) || !sv_is_device_ptr(counts))
    return ctx->fail(SEGVLAD_ERR_ARG, "kmeans_step: sums / counts are accumulated into and must be device memory");
  d_sums = sums;
  d_cnt = counts;
  SV_TRY(assign_phase(ctx, tokens, B, N));   // labels (cosine arg-max against the normalised centres, first maximum), Xt, 1 / ||x||
  {
    StageScope sc(ctx, "kmeans");
    SV_TRY(sv_launch_centroid_sums(ctx, ctx->s_xt.as<float>(), ctx->s_rnorm.as<float>(), ctx->s_labels.as<uint8_t>(), B, N, D, K,
                                   (double*)d_sums, (int64_t*)d_cnt));
    sc.count(2);
  }
  if (labels_out) {
    SV_TRY(sv_out(ctx, labels_out, (size_t)B * N, &d_lab));
    SV_HIP(hipMemcpyAsync(d_lab, ctx->s_labels.p, (size_t)B * N, hipMemcpyDeviceToDevice, ctx->stream));
  }
  return sv_finish(ctx);
}

// ---- K-parametric aggregation of given residuals + labels (vlad_matmuls_per_cluster) --------------------
__global__ void fill_ones_kernel(float* p, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) p[j] = 1.f;
}

int segvlad_cluster_aggregate(segvlad_ctx* ctx, int num_c, const float* res, const uint8_t* labels, int N, int D,
                              const uint64_t* inc_bits, int S, const uint8_t* adj, float* out) {
  CHECK_CTX();
  CHECK_NO_OPEN_DESCRIBE("cluster_aggregate");
  if (num_c <= 0 || num_c > 256 || N <= 0 || D <= 0 || (D % 4) || S < 0)
    return ctx->fail(SEGVLAD_ERR_ARG, "cluster_aggregate: bad shape num_c=%d N=%d D=%d S=%d", num_c, N, D, S);
  if (S == 0) return SEGVLAD_OK;
  if (!res || !labels || !inc_bits || !out) return ctx->fail(SEGVLAD_ERR_ARG, "cluster_aggregate: null pointer");
  const int nw = (N + 63) / 64, SC = (S + 63) / 64;
  const void *d_res, *d_lab, *d_inc, *d_adj = nullptr;
  void* d_out;
  SV_TRY(sv_in(ctx, res, (size_t)N * D * 4, &d_res));
  SV_TRY(sv_in(ctx, labels, (size_t)N, &d_lab));
  SV_TRY(sv_in(ctx, inc_bits, (size_t)S * nw * 8, &d_inc));
  if (adj) SV_TRY(sv_in(ctx, adj, (size_t)S * S, &d_adj));
  SV_TRY(sv_out(ctx, out, (size_t)S * num_c * D * 4, &d_out));
  SV_HIP(ctx->s_rnorm.reserve((size_t)N * 4));
  SV_HIP(ctx->s_colmask.reserve((size_t)N * SC * 8));
  SV_HIP(ctx->s_gscale.reserve((size_t)(S + 1) * 4));
  SV_HIP(ctx->s_segoff.reserve(2 * sizeof(int32_t)));
  SV_HIP(ctx->s_adjoff.reserve(2 * sizeof(int64_t)));
  const int32_t so[2] = {0, S};
  const int64_t ao[2] = {0, (int64_t)S * S};
  SV_HIP(hipMemcpyAsync(ctx->s_segoff.p, so, sizeof(so), hipMemcpyHostToDevice, ctx->stream));
  SV_HIP(hipMemcpyAsync(ctx->s_adjoff.p, ao, sizeof(ao), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(fill_ones_kernel, dim3((N + 255) / 256), dim3(256), 0, ctx->stream, ctx->s_rnorm.as<float>(), N);
  SV_TRY(sv_launch_prep(ctx, (const uint8_t*)d_lab, (const uint64_t*)d_inc, ctx->s_segoff.as<int32_t>(),
                        ctx->s_adjoff.as<int64_t>(), (const uint8_t*)d_adj, 1, N, num_c, S, SC, ctx->s_colmask.as<uint64_t>(),
                        ctx->s_gscale.as<float>()));
  SV_TRY(sv_launch_aggregate(ctx, (const float*)d_res, ctx->s_rnorm.as<float>(), (const uint8_t*)d_lab,
                             ctx->s_colmask.as<uint64_t>(), nullptr, num_c, D, ctx->s_segoff.as<int32_t>(),
                             ctx->s_gscale.as<float>(), 1, N, SC, (float*)d_out, nullptr));
  SV_HIP(hipStreamSynchronize(ctx->stream));  // so[] / ao[] live on this frame
  return sv_finish(ctx);
}

// ---- PCA ------------------------------------------------------------------------------------------------
__global__ void pca_scale_kernel(const float* var, int P, int whiten, float* scale) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < P) scale[j] = whiten ? (float)(1.0 / sqrt((double)var[j])) : 1.f;
}

int segvlad_pca_set(segvlad_ctx* ctx, const float* mean, const float* comps, const float* expl_var, int P, int KD,
                    int whiten) {
  CHECK_CTX();
  if (!comps || P <= 0 || KD <= 0) return ctx->fail(SEGVLAD_ERR_ARG, "pca_set: need comps, P>0, KD>0");
  if (whiten && !expl_var) return ctx->fail(SEGVLAD_ERR_ARG, "pca_set: whiten needs explained_variance");
  SV_HIP(ctx->pca_comps.reserve((size_t)P * KD * sizeof(float)));
  SV_HIP(ctx->pca_mean.reserve((size_t)KD * sizeof(float)));
  SV_HIP(ctx->pca_scale.reserve((size_t)P * sizeof(float)));
  SV_HIP(hipMemcpyAsync(ctx->pca_comps.p, comps, (size_t)P * KD * sizeof(float), hipMemcpyDefault, ctx->stream));
  if (mean)
    SV_HIP(hipMemcpyAsync(ctx->pca_mean.p, mean, (size_t)KD * sizeof(float), hipMemcpyDefault, ctx->stream));
  else
    SV_HIP(hipMemsetAsync(ctx->pca_mean.p, 0, (size_t)KD * sizeof(float), ctx->stream));
  const void* dvar = nullptr;
  if (whiten) SV_TRY(sv_in(ctx, expl_var, (size_t)P * sizeof(float), &dvar));
  hipLaunchKernelGGL(pca_scale_kernel, dim3((P + 255) / 256), dim3(256), 0, ctx->stream, (const float*)dvar, P, whiten,
                     ctx->pca_scale.as<float>());
  SV_HIP(hipGetLastError());
  ctx->P = P;
  ctx->KD = KD;
  ctx->whiten = whiten;
  ctx->pca_cproj_valid = false;
  ctx->pca_w_scale = 0.f;
  if (KD % 32 == 0) {  // fp16 two-term split of the components for the 16-bit MFMA projection
    float wmax = 0.f, mmax = 0.f;
    SV_TRY(sv_maxabs(ctx, ctx->pca_comps.as<float>(), (int64_t)P * KD, &wmax));
    SV_TRY(sv_maxabs(ctx, ctx->pca_mean.as<float>(), KD, &mmax));
    if (wmax > 0.f && std::isfinite(wmax)) {
      int e;
      frexpf(wmax, &e);
      ctx->pca_w_scale = ldexpf(1.f, 14 - e);
      ctx->pca_mean_maxabs = mmax;
      SV_HIP(ctx->pca_w1.reserve((size_t)sv_x3_rows(P) * KD * 2));   // blocked planes, rows padded to whole tiles
      SV_HIP(ctx->pca_w2.reserve((size_t)sv_x3_rows(P) * KD * 2));
      SV_TRY(sv_launch_split_f16x2(ctx, ctx->pca_comps.as<float>(), P, KD, nullptr, ctx->pca_w_scale, ctx->pca_w1.as<uint16_t>(),
                                   ctx->pca_w2.as<uint16_t>()));
    }
  }
  SV_HIP(hipStreamSynchronize(ctx->stream));
  return sv_finish(ctx);
}

int segvlad_pca_apply(segvlad_ctx* ctx, const float* X, int n, float* Y, int l2norm) {
  CHECK_CTX();
  if (ctx->P == 0) return ctx->fail(SEGVLAD_ERR_STATE, "pca_apply: call segvlad_pca_set first");
  if (n < 0) return ctx->fail(SEGVLAD_ERR_ARG, "pca_apply: n<0");
  if (n == 0) return SEGVLAD_OK;
  if (!X || !Y) return ctx->fail(SEGVLAD_ERR_ARG, "pca_apply: null pointer");
  const void* dx;
  void* dy;
  SV_TRY(sv_in(ctx, X, (size_t)n * ctx->KD * sizeof(float), &dx));
  SV_TRY(sv_out(ctx, Y, (size_t)n * ctx->P * sizeof(float), &dy));
  const bool x3 = ctx->pca_w_scale > 0.f && !ctx->opt.pca_fp32;
  float xscale = 1.f;
  if (x3) {  // scale so that |x - mean| * s < 2^15: no fp16 overflow, sub-normal losses far below fp32 epsilon
    float xmax = 0.f;
    SV_TRY(sv_maxabs(ctx, (const float*)dx, (int64_t)n * ctx->KD, &xmax));
    const float bound = xmax + ctx->pca_mean_maxabs;
    if (bound > 0.f && std::isfinite(bound)) {
      int e;
      frexpf(bound, &e);
      xscale = ldexpf(1.f, 14 - e);
    }
    SV_HIP(ctx->s_xh1.reserve((size_t)sv_x3_rows(n) * ctx->KD * 2));
    SV_HIP(ctx->s_xh2.reserve((size_t)sv_x3_rows(n) * ctx->KD * 2));
  }
  {
    StageScope sc(ctx, "pca");
    if (x3) {
      SV_TRY(sv_launch_split_f16x2(ctx, (const float*)dx, n, ctx->KD, ctx->pca_mean.as<float>(), xscale, ctx->s_xh1.as<uint16_t>(),
                                   ctx->s_xh2.as<uint16_t>()));
      SV_TRY(sv_launch_gemm_f16x3(ctx, ctx->s_xh1.as<uint16_t>(), ctx->s_xh2.as<uint16_t>(), ctx->pca_w1.as<uint16_t>(),
                                  ctx->pca_w2.as<uint16_t>(), n, ctx->P, ctx->KD, 1.f / (xscale * ctx->pca_w_scale),
                                  ctx->pca_scale.as<float>(), (float*)dy));
      sc.count(3);
    } else {
      SV_TRY(sv_launch_gemm_nt(ctx, 0, (const float*)dx, ctx->pca_comps.as<float>(), (float*)dy, n, ctx->P, ctx->KD, ctx->P,
                               ctx->pca_mean.as<float>(), ctx->pca_scale.as<float>(), nullptr, nullptr));
      sc.count();
    }
    if (l2norm) {
      SV_TRY(sv_launch_normalize_rows(ctx, (const float*)dy, n, ctx->P, (float*)dy));
      sc.count();
    }
  }
  return sv_finish(ctx);
}

int segvlad_normalize_rows(segvlad_ctx* ctx, const float* X, int n, int d, float* Y) {
  CHECK_CTX();
  if (n < 0 || d <= 0) return ctx->fail(SEGVLAD_ERR_ARG, "normalize_rows: bad shape");
  if (n == 0) return SEGVLAD_OK;
  if (!X || !Y) return ctx->fail(SEGVLAD_ERR_ARG, "normalize_rows: null pointer");
  const void* dx;
  void* dy;
  SV_TRY(sv_in(ctx, X, (size_t)n * d * sizeof(float), &dx));
  SV_TRY(sv_out(ctx, Y, (size_t)n * d * sizeof(float), &dy));
  SV_TRY(sv_launch_normalize_rows(ctx, (const float*)dx, n, d, (float*)dy));
  return sv_finish(ctx);
}

// ---- database / search ---------------------------------------------------------------------------------------
int segvlad_db_reset(segvlad_ctx* ctx) {
  CHECK_CTX();
  ctx->db_n = 0;
  ctx->db_d = 0;
  ctx->db_has_img = false;
  ctx->db_split_rows = 0;
  ctx->db_f16_rows = 0;
  ctx->db_f16_scale = 0.f;
  ctx->db_maxabs = 0.f;
  ctx->db_rn_max = 0.f;
  ctx->db_rn_max_rows = 0;
  ctx->db_heur_off = false;
  if (ctx->tail_rows_since > 0) (void)hipStreamSynchronize(ctx->stream);   // (the tail's pinned totals have landed)
  ctx->tail_fail_base = ctx->h_pin ? reinterpret_cast<volatile uint32_t*>(ctx->h_pin)[8] : 0u;
  ctx->tail_rows_since = 0;
  return SEGVLAD_OK;
}

int segvlad_db_add(segvlad_ctx* ctx, const float* R, int n, int d, const int32_t* img_of_seg) {
  CHECK_CTX();
  if (n < 0 || d <= 0) return ctx->fail(SEGVLAD_ERR_ARG, "db_add: bad shape");
  if (ctx->db_n > 0 && d != ctx->db_d) return ctx->fail(SEGVLAD_ERR_ARG, "db_add: d=%d but the index holds d=%d", d, ctx->db_d);
  if (ctx->db_n > 0 && ctx->db_has_img != (img_of_seg != nullptr))
    return ctx->fail(SEGVLAD_ERR_ARG, "db_add: img_of_seg must be given for all rows or none");
  if (n == 0) return SEGVLAD_OK;
  if (!R) return ctx->fail(SEGVLAD_ERR_ARG, "db_add: null rows");
  const int64_t n_new = ctx->db_n + n;
  // grow (keeping old contents)
  auto grow = [&](DevBuf& b, size_t old_bytes, size_t new_bytes) -> hipError_t {
    if (new_bytes <= b.cap) return hipSuccess;
    DevBuf nb;
    nb.tag = b.tag;
    nb.guard = b.guard;
    nb.fixed = b.fixed;
    hipError_t e = nb.reserve(new_bytes + new_bytes / 2);
    if (e != hipSuccess) return e;
    if (old_bytes) {
      e = hipMemcpyAsync(nb.p, b.p, old_bytes, hipMemcpyDeviceToDevice, ctx->stream);
      if (e != hipSuccess) return e;
      e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess) return e;
    }
    b.release();
    b = nb;
    return hipSuccess;
  };
  SV_HIP(grow(ctx->db_rows, (size_t)ctx->db_n * d * 4, (size_t)n_new * d * 4));
  SV_HIP(grow(ctx->db_norms, (size_t)ctx->db_n * 4, (size_t)n_new * 4));
  float* dst = ctx->db_rows.as<float>() + (size_t)ctx->db_n * d;
  SV_HIP(hipMemcpyAsync(dst, R, (size_t)n * d * 4, hipMemcpyDefault, ctx->stream));
  if (img_of_seg) {
    SV_HIP(grow(ctx->db_img, (size_t)ctx->db_n * 4, (size_t)n_new * 4));
    SV_HIP(hipMemcpyAsync(ctx->db_img.as<int32_t>() + ctx->db_n, img_of_seg, (size_t)n * 4, hipMemcpyDefault, ctx->stream));
    ctx->db_has_img = true;
  }
  SV_TRY(sv_launch_row_sumsq(ctx, dst, n, d, ctx->db_norms.as<float>() + ctx->db_n));
  ctx->db_n = n_new;
  ctx->db_d = d;
  ctx->db_heur_off = false;
  if (ctx->tail_rows_since > 0) (void)hipStreamSynchronize(ctx->stream);   // (the tail's pinned totals have landed)
  ctx->tail_fail_base = ctx->h_pin ? reinterpret_cast<volatile uint32_t*>(ctx->h_pin)[8] : 0u;
  ctx->tail_rows_since = 0;
  return sv_finish(ctx);
}

int segvlad_db_size(segvlad_ctx* ctx, int64_t* n_rows, int* d) {
  if (!ctx) return SEGVLAD_ERR_ARG;
  if (n_rows) *n_rows = ctx->db_n;
  if (d) *d = ctx->db_d;
  return SEGVLAD_OK;
}

// Exact search.  Small databases: distance matrix + radix select.  Large ones: a strided 1/16^L sample gives an
// exact UPPER bound T0[q] of the k-th smallest distance; each finer level re-runs the distance GEMM with the
// epilogue keeping only entries <= T[q] (about 16 k per query), whose exact top-k tightens T for the next level;
// the last level covers every row, so the final top-k is exact (ties included: all entries <= T are candidates
// and the final order is (distance, id)).  A query whose candidate or refine list overflows (adversarial data) is
// redone -- alone -- on the matrix path; the other queries keep their filtered result.
static int search_matrix(segvlad_ctx* ctx, const float* dq, int m, int64_t n, int d, int k, const float* qn, float* dd2,
                         int64_t* didx) {
  const int64_t ld = (n + 3) & ~3ll;
  int64_t rows = ld > 0 ? (int64_t)(2ll << 30) / (ld * 4) : m;
  if (rows < 128) rows = 128;
  if (rows > m) rows = m;
  SV_HIP(ctx->s_dist.reserve((size_t)rows * (ld > 0 ? ld : 1) * 4));
  for (int64_t q0 = 0; q0 < m; q0 += rows) {
    const int mm = (int)((m - q0 < rows) ? (m - q0) : rows);
    {
      StageScope sc(ctx, "knn_gemm");
      SV_TRY(sv_launch_gemm_nt(ctx, 1, dq + (size_t)q0 * d, ctx->db_rows.as<float>(), ctx->s_dist.as<float>(), mm, (int)n, d,
                               ld, nullptr, nullptr, qn + q0, ctx->db_norms.as<float>()));
      sc.count();
    }
    {
      StageScope sc(ctx, "knn_select");
      SV_TRY(sv_launch_select_topk(ctx, ctx->s_dist.as<float>(), ld, mm, n, k, dd2 + (size_t)q0 * k, didx + (size_t)q0 * k, k, 0));
      sc.count();
    }
  }
  return SEGVLAD_OK;
}

// rows of the redo / overflow passes: gather flagged query rows into a dense block, scatter their results back
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ X, const float* __restrict__ xn,
                                                          const int32_t* __restrict__ rows, int d, float* __restrict__ Y,
                                                          float* __restrict__ yn) {
  const int r = blockIdx.x;
  const int64_t src = rows[r];
  for (int j = threadIdx.x; j < d; j += 256) Y[(int64_t)r * d + j] = X[src * d + j];
  if (threadIdx.x == 0) yn[r] = xn[src];
}
__global__ __launch_bounds__(256) void scatter_topk_kernel(const float* __restrict__ d2, const int64_t* __restrict__ idx,
                                                           const int32_t* __restrict__ rows, int k, float* __restrict__ d2_out,
                                                           int64_t* __restrict__ idx_out) {
  const int r = blockIdx.x;
  const int64_t dst = rows[r];
  for (int j = threadIdx.x; j < k; j += 256) {
    d2_out[dst * k + j] = d2[(int64_t)r * k + j];
    idx_out[dst * k + j] = idx[(int64_t)r * k + j];
  }
}

// ---- the level scheme ---------------------------------------------------------------------------------------------
struct SearchPlan {
  int levels = 0;          // filter levels after the sampled exact level
  int64_t stride0 = 1;     // stride of the sampled exact level (16^levels)
  int kind = 3;            // filter arithmetic: 1 f16, 2 bf16x3, 3 fp32
  int d = 0, k = 0;
  int64_t n = 0;
  float c_eps = 0.f, inv_scale = 1.f, rn_max = 0.f;
  int ratio_last = 16;     // sample growth into the LAST (full) level; the levels before it grow by SV_RATIO = 16
  double pfail = 2e-5;     // heur_rank_small's tolerance when ratio_last != 16 (one redone query in ~1000 passes of 50)
};
constexpr int SV_RATIO = 16, SV_CAP = 8192, SV_RCAP = 512, SV_CHUNK = 16384;

// Smallest rank r such that a threshold at the r-th smallest value of a 1/16 sample admits, 4 sigma below its
// expectation 16 r, still `target` values of the full set (relative spread of the r-th order statistic ~ 1/sqrt(r)).
static int heur_rank(int target) {
  int r = 16;
  while (16.0 * r - 64.0 * std::sqrt((double)r) < (double)target) ++r;
  return r;
}

// The same question for a sample `ratio` times smaller and SMALL ranks, where the normal approximation is off: the
// number of full-set values below the r-th smallest sample value is ~ ratio * Gamma(r, 1), so take the smallest r with
// P[Gamma(r, 1) < target / ratio] < pfail (2e-5: one redo in ~1000 passes of 50 queries).
static int heur_rank_small(int target, int ratio, double pfail) {
  const double x = (double)target / ratio, ex = std::exp(-x);
  double term = 1.0, sum = 0.0;   // sum_{i < r} x^i / i!
  for (int r = 1; r < target; ++r) {
    sum += term;
    term *= x / r;
    if (r >= 2 && 1.0 - ex * sum < pfail) return r;
  }
  return target;
}

// One pass of the level scheme over m <= SV_CHUNK query rows: exact distances to the coarsest sample, then `levels`
// filter GEMMs over samples 16x larger each, the last one covering every row, then the exact refinement.
//   rigorous : every threshold is the k-th smallest distance of the previous (coarser) sample -- an UPPER bound of
//              the k-th smallest of the finer one, so no true neighbour is ever dropped; ~16 k candidates per level.
//   heuristic: thresholds at much lower ranks r_j (heur_rank) that admit ~16 r_j candidates -- 5-10x fewer -- and are
//              verified afterwards: a level whose list holds fewer than r_{j+1} entries, or a final list whose k-th
//              smallest approximate distance A_k exceeds the threshold T it was collected under (then {d2~ <= A_k + 2 eps}
//              might not be contained in the collected {d2~ <= T + 2 eps}), flags the query; flagged queries are redone
//              with the rigorous thresholds.  Exactness never depends on the ranks; they only decide how often the redo runs.
// q16a / q16b: this chunk's 16-bit query planes (f16: plane, unused; bf16x3: hi, lo).  fail_rows [m]: flags;
// fail_count[0] / fail_count[1]: running counts of the flagged rows and of the second-tier rows (adjacent: ONE read-back per
// chunk); rovf_rows_in [m]: second-tier flags, zero on entry; n_rovf_seen: the caller's copy of fail_count[1] so far.
static int levels_chunk(segvlad_ctx* ctx, const SearchPlan& pl, bool heuristic, const float* qp, const uint16_t* q16a,
                        const uint16_t* q16b, const float* qn, int m, float* out_d2, int64_t* out_idx, uint32_t* fail_rows,
                        uint32_t* fail_count, uint32_t* rovf_rows_in, uint32_t* n_fail_host, uint32_t* n_rovf_seen, int phase = 0,
                        uint32_t* tail_stats = nullptr) {
  // tail_stats != null (one query image per pass, round 6): NO read-back -- small_tail_kernel is launched behind the refinement,
  // reads the two counters on the device and finishes flagged rows there (small_pass_kernels.hip); the host learns nothing
  // about them in this call (*n_fail_host = 0), segvlad_search_stats fetches the kernel's counters on demand
  // phase 0: the whole pass.  1: the exact sample level only; 2: everything behind it -- a batch search enqueues phase 1 (which
  // needs neither the query plane nor the filter's margin) BEFORE it waits for the two scalars that fix them, so that the device
  // has that level to run while the host waits (m > 128 only: the single-image plan's fused level 0 is one chain with its pass)
  const int d = pl.d, k = pl.k, levels = pl.levels;
  const int64_t n = pl.n;
  const float* R = ctx->db_rows.as<float>();
  const float* rn = ctx->db_norms.as<float>();
  // rank of the threshold handed to level j+1 (rank[levels] = k: the final top-k)
  int rank[8];
  rank[levels] = k;
  for (int j = levels - 1; j >= 0; --j)
  {
    const int ratio_j = (j + 1 == levels) ? pl.ratio_last : SV_RATIO;   // growth from level j's sample to level j + 1's
    rank[j] = !heuristic ? k
                         : std::min(rank[j + 1], ratio_j == SV_RATIO ? heur_rank(rank[j + 1]) : heur_rank_small(rank[j + 1], ratio_j, pl.pfail));
  }
  const int64_t n0 = (n + pl.stride0 - 1) / pl.stride0;
  const int64_t ld0 = (n0 + 3) & ~3ll;
  float* thr = ctx->s_thr_d2.as<float>();
  // second refinement tier (filters with an approximate domain only): [m] row flags + 1 count, [m] band limits
  uint32_t* rovf_rows = nullptr;
  uint32_t* rovf_count = fail_count + 1;
  float* ref_lim = nullptr;
  if (pl.kind != 3) {
    SV_HIP(ctx->s_ref_lim.reserve((size_t)m * 4));
    rovf_rows = rovf_rows_in;
    ref_lim = ctx->s_ref_lim.as<float>();
  }
  const uint32_t* poison_dev = nullptr;   // see sv_launch_refine_exact
  bool tail_fused = false;                // the refinement kernel finished the flagged rows itself (no small_tail_kernel)
  const int r0 = rank[0];
  bool l0_fused = false, l0_lists = false;
  if (phase != 2) {  // level 0: exact (fp32) top-r0 of the coarsest sample -> thr[q][r0-1]
    {
      StageScope sc(ctx, "knn_level0");   // its own stage: "knn_gemm" then times the filter kernel's launches only
      if (m <= 128 && n0 <= 4096 && pl.kind != 3) {
        // one query image per pass: the K split's reduction and the rank select in ONE launch (l0_reduce_rank_kernel)
        const float* parts = nullptr;
        int splits = 1;
        SV_TRY(sv_launch_l2_strided_parts(ctx, qp, R, ctx->s_dist.as<float>(), m, (int)n0, d, ld0, qn, rn, (int)pl.stride0, &parts, &splits));
        sc.count();
        if (splits > 1) {
          SV_TRY(sv_launch_l0_reduce_rank(ctx, parts, splits, m, (int)n0, ld0, qn, rn, (int)pl.stride0, r0, thr,
                                          ctx->s_cand_cnt.as<uint32_t>(), fail_rows));
          sc.count();
          l0_fused = true;
        }
      } else if (heuristic && pl.kind == 1 && m > 128 && n0 <= 4096 && q16a && ctx->opt.batch_l0_f16 && !ctx->f16_scale_dev &&
                 sv_f16_kblock(ctx->opt, d) && ctx->opt.batch_l0_f16 != 2) {
        // deep rows (raw K*D descriptors): the sample through the FILTER KERNEL under thresholds of +inf -- every sampled row lands in
        // the candidate lists (n0 <= 4096 entries per query), the mode-0 select below ranks them like any level's.  The no-LDS sample
        // kernel re-reads the query plane (2 GB at 10 000 x 98 304: not L2 resident) once per 32 sample rows: 8.5 ms for 196 rows,
        // against ~0.6 ms at the filter's rate.
        SV_HIP(hipMemsetD32Async((hipDeviceptr_t)thr, 0x7f800000, (size_t)m, ctx->stream));
        SV_HIP(hipMemsetAsync(ctx->s_cand_cnt.p, 0, (size_t)m * 4, ctx->stream));
        SV_TRY(sv_launch_f16_filter(ctx, q16a, ctx->db_f16.as<uint16_t>(), m, (int)n0, d, (int)pl.stride0, pl.inv_scale, qn, rn, thr, 1, 2.f,
                                    pl.c_eps, pl.rn_max, ctx->s_cand_cnt.as<uint32_t>(), ctx->s_cand_d2.as<float>(),
                                    ctx->s_cand_id.as<uint32_t>(), SV_CAP));
        sc.count();
        l0_lists = true;
      } else if (heuristic && pl.kind == 1 && m > 128 && n0 <= 4096 && q16a && ctx->opt.batch_l0_f16 && !ctx->f16_scale_dev) {
        // round 6: a guessed threshold needs no exact distances -- the sample through the filter's own fp16 product (~15 us instead
        // of 115-160 us of poorly filled fp32 MFMA tiles); the thresholds are then approximate-domain values like every later level's
        SV_TRY(sv_launch_sample_f16_batch(ctx, q16a, ctx->db_f16.as<uint16_t>(), m, (int)n0, d, pl.stride0, pl.inv_scale, qn, rn,
                                          ctx->s_dist.as<float>(), ld0));
        sc.count();
      } else {
        SV_TRY(sv_launch_l2_strided(ctx, qp, R, ctx->s_dist.as<float>(), m, (int)n0, d, ld0, qn, rn, (int)pl.stride0, true));
        sc.count();
      }
    }
    StageScope sc(ctx, "knn_select");
    if (l0_fused) {
      // (thr[q] is in place)
    } else if (l0_lists) {
      SV_TRY(sv_launch_select_approx(ctx, ctx->s_cand_cnt.as<uint32_t>(), ctx->s_cand_d2.as<float>(), ctx->s_cand_id.as<uint32_t>(), m,
                                     SV_CAP, r0, 0, 1, nullptr, 0, qn, pl.c_eps, pl.rn_max, thr, ctx->s_ref_cnt.as<uint32_t>(),
                                     ctx->s_ref_id.as<uint32_t>(), SV_RCAP, fail_rows, fail_count));
    } else if (n0 <= 4096 && pl.kind != 3) {
      // a short sample row: only its r0-th smallest distance is needed -- the wave-per-query register select of the
      // candidate lists (mode 0: thr[q] = rank-th smallest), the distance block standing in for a list of n0 entries
      SV_TRY(sv_launch_select_approx(ctx, ctx->s_cand_cnt.as<uint32_t>(), ctx->s_dist.as<float>(), ctx->s_cand_id.as<uint32_t>(), m,
                                     (int)ld0, r0, 0, 0, nullptr, 0, qn, pl.c_eps, pl.rn_max, thr, ctx->s_ref_cnt.as<uint32_t>(),
                                     ctx->s_ref_id.as<uint32_t>(), SV_RCAP, fail_rows, fail_count, nullptr, nullptr, nullptr, (int)n0));
    } else {
      SV_TRY(sv_launch_select_topk(ctx, ctx->s_dist.as<float>(), ld0, m, n0, r0, thr, ctx->s_thr_idx.as<int64_t>(), r0, 0));
    }
    if (!l0_fused) sc.count();
  }
  if (phase == 1) return SEGVLAD_OK;
  const bool l0_small = n0 <= 4096 && pl.kind != 3;
  const float* thr_ptr = l0_small ? thr : thr + (r0 - 1);
  int64_t thr_ld = l0_small ? 1 : r0;
  // rigorous: level-0 thresholds are exact distances, one margin covers the filter's error.  heuristic: the final
  // check needs the collected set to reach 2 eps beyond the threshold at every level.
  float eps_mult = heuristic ? 2.f : 1.f;
  int64_t stride = pl.stride0;
  std::vector<uint32_t> hcnt;
  // carry (round 6b): the stride-16 level's filter has already evaluated 1/16 of the rows with the arithmetic of the last level; its
  // survivors under the NEXT threshold stay in the lists (select mode 2) and the last level runs over the other 15/16 only.  Guessed
  // thresholds only (every level collects under thr + 2 eps there, so the kept set is exactly what the last level would append).
  const int64_t n_comp = n - (n + SV_RATIO - 1) / SV_RATIO;   // rows that are not multiples of 16
  const bool carry = heuristic && pl.kind == 1 && levels >= 2 && pl.ratio_last == SV_RATIO && m > 128 && n_comp < 0x7fffffffLL &&
                     sv_f16_filter_skip_ok(ctx, m, n_comp, d);
  if (carry) ctx->sstats.carry_rows = n - n_comp;
  for (int lv = 1; lv <= levels; ++lv) {
    stride /= (lv == levels) ? pl.ratio_last : SV_RATIO;
    const bool last = (lv == levels);
    const int64_t ns = (last && carry) ? n_comp : (n + stride - 1) / stride;
    // the candidate counters start every level at zero: the mode-0 selects of the approximate-domain filters leave them so;
    // only the first filter level behind a select_topk level 0, and the fp32 filter's select, need the memset
    if (pl.kind == 3 || (lv == 1 && !l0_small)) SV_HIP(hipMemsetAsync(ctx->s_cand_cnt.p, 0, (size_t)m * 4, ctx->stream));
    if (pl.kind != 3) {
      {
        StageScope sc(ctx, "knn_gemm");
        if (pl.kind == 1)
          SV_TRY(sv_launch_f16_filter(ctx, q16a, ctx->db_f16.as<uint16_t>(), m, (int)ns, d, (int)stride, pl.inv_scale, qn, rn, thr_ptr,
                                      thr_ld, eps_mult, pl.c_eps, pl.rn_max, ctx->s_cand_cnt.as<uint32_t>(), ctx->s_cand_d2.as<float>(),
                                      ctx->s_cand_id.as<uint32_t>(), SV_CAP, (last && carry) ? SV_RATIO : 0));
        else
          SV_TRY(sv_launch_bf16_filter(ctx, q16a, q16b, ctx->db_hi.as<uint16_t>(), ctx->db_lo.as<uint16_t>(), m, (int)ns, d, (int)stride,
                                       qn, rn, thr_ptr, thr_ld, eps_mult, pl.c_eps, pl.rn_max, ctx->s_cand_cnt.as<uint32_t>(),
                                       ctx->s_cand_d2.as<float>(), ctx->s_cand_id.as<uint32_t>(), SV_CAP));
        sc.count();
      }
      if (ctx->opt.debug_search || (ctx->opt.search_stats && last)) {  // candidate-list statistics (synchronises)
        hcnt.resize(m);
        SV_HIP(hipStreamSynchronize(ctx->stream));
        SV_HIP(hipMemcpy(hcnt.data(), ctx->s_cand_cnt.p, (size_t)m * 4, hipMemcpyDeviceToHost));
        uint64_t tot = 0;
        uint32_t mx = 0, over = 0;
        for (uint32_t c : hcnt) {
          tot += c;
          if (c > mx) mx = c;
          if (c > (uint32_t)SV_CAP) ++over;
        }
        if (last) {
          ctx->sstats.cand_sum += (int64_t)tot;
          if ((int64_t)mx > ctx->sstats.cand_max) ctx->sstats.cand_max = mx;
        }
        if (ctx->opt.debug_search)
          fprintf(stderr, "[search] %s m=%d level %d/%d ns=%lld rank %d: candidates mean %.1f max %u, %u lists over cap %d\n",
                  heuristic ? "heuristic" : "rigorous", m, lv, levels, (long long)ns, rank[lv], (double)tot / m, mx, over, SV_CAP);
      }
      {
        StageScope sc(ctx, "knn_select");
        SV_TRY(sv_launch_select_approx(ctx, ctx->s_cand_cnt.as<uint32_t>(), ctx->s_cand_d2.as<float>(), ctx->s_cand_id.as<uint32_t>(), m,
                                       SV_CAP, rank[lv], last ? 1 : ((carry && lv + 1 == levels) ? 2 : 0), heuristic ? 1 : 0, thr_ptr, thr_ld, qn,
                                       pl.c_eps, pl.rn_max, thr,
                                       ctx->s_ref_cnt.as<uint32_t>(), ctx->s_ref_id.as<uint32_t>(), SV_RCAP, fail_rows, fail_count,
                                       rovf_rows, rovf_count, ref_lim));
        sc.count();
        if (last && m > 128) {
          // batches: bands of 32 consecutive query rows (the segments of an image) over the union of their rows where they
          // overlap, the rest row by row (refine_group_kernels.hip); same bits either way
          int nl = 0;
          SV_TRY(sv_launch_refine_grouped(ctx, qp, R, m, d, qn, rn, ctx->s_ref_cnt.as<uint32_t>(), ctx->s_ref_id.as<uint32_t>(), SV_RCAP,
                                          k, out_d2, out_idx, &nl));
          sc.count(nl);
          if (ctx->opt.search_stats && nl > 1) {
            int64_t ng = 0, gg = 0, us = 0;
            SV_TRY(sv_refine_group_stats(ctx, m, &ng, &gg, &us));
            ctx->sstats.grp_groups += gg;
            ctx->sstats.grp_union_sum += us;
          }
        } else if (last) {
          // a device-driven pass whose head was small_head_kernel (it repairs a poisoned hand-over buffer): the refinement finishes the
          // flagged rows itself -- no small_tail_kernel behind it (a kernel boundary of 4-5 us, whatever the kernel does)
          SvSmallFinish fz;
          if (tail_stats && ctx->small_head_ran && ctx->opt.small_tail != 2) {
            fz.on = 1;
            fz.rovf_rows = rovf_rows;
            fz.ref_lim = ref_lim;
            fz.cand_cnt = ctx->s_cand_cnt.as<uint32_t>();
            fz.cand_d2 = ctx->s_cand_d2.as<float>();
            fz.cand_id = ctx->s_cand_id.as<uint32_t>();
            fz.cap = SV_CAP;
            fz.n_db = n;
            fz.stats = tail_stats;
            SV_TRY(sv_launch_small_tail_debug(ctx, m, fail_rows, fail_count, rovf_rows, ref_lim));
          }
          SV_TRY(sv_launch_refine_exact(ctx, qp, R, m, d, qn, rn, ctx->s_ref_cnt.as<uint32_t>(), ctx->s_ref_id.as<uint32_t>(), SV_RCAP,
                                        k, out_d2, out_idx, nullptr, fail_rows, fail_count, &poison_dev, &fz, &tail_fused));
          sc.count();
        }
      }   // (the stage's stop event is recorded before the host waits below)
      if (last && tail_stats && tail_fused) {
        if (n_fail_host) *n_fail_host = 0;   // (the refinement kernel has finished its flagged rows itself)
      } else if (last && tail_stats) {
        StageScope sc(ctx, "knn_select");
        SV_TRY(sv_launch_small_tail(ctx, qp, R, qn, rn, n, d, m, k, fail_rows, fail_count, rovf_rows, ref_lim, ctx->s_cand_cnt.as<uint32_t>(),
                                    ctx->s_cand_d2.as<float>(), ctx->s_cand_id.as<uint32_t>(), SV_CAP, out_d2, out_idx, tail_stats));
        sc.count();
        if (n_fail_host) *n_fail_host = 0;
      } else if (last) {
        // one read-back per chunk: rows flagged for the redo / matrix-path fallback (handled by the caller) and rows whose
        // refine band outgrew the first-tier list.  The latter are refined here, straight from their candidate lists,
        // which the next chunk would overwrite.
        uint32_t h_cnt[2] = {0, 0}, h_poison = 0;
        SV_HIP(hipMemcpyAsync(&h_cnt[0], fail_count, 8, hipMemcpyDeviceToHost, ctx->stream));
        if (poison_dev) SV_HIP(hipMemcpyAsync(&h_poison, poison_dev, 4, hipMemcpyDeviceToHost, ctx->stream));
        SV_HIP(hipStreamSynchronize(ctx->stream));
        if (h_poison) SV_TRY(sv_refine_small_repair(ctx));   // a checked hand-over failed (its rows are flagged): fresh buffers
        if (n_fail_host) *n_fail_host = h_cnt[0];
        const uint32_t seen = n_rovf_seen ? *n_rovf_seen : 0u;
        if (n_rovf_seen) *n_rovf_seen = h_cnt[1];
        h_cnt[1] -= seen;   // the second-tier rows of THIS chunk
        if (h_cnt[1]) {
          StageScope sc(ctx, "knn_select");
          SV_TRY(sv_launch_refine2_compact(ctx, rovf_rows, ref_lim, ctx->s_cand_cnt.as<uint32_t>(), ctx->s_cand_d2.as<float>(),
                                           ctx->s_cand_id.as<uint32_t>(), m, SV_CAP));
          if (m > 128 && d > 4096) {   // deep rows: a second-tier list as its own union GEMM (parallel over its 128-row tiles)
            // In blocks of <= 1024 + 128 query rows: the grouped path's scratch is per ROW of the block it is handed (positions
            // [rows][8192] u16, keys [rows][2048] u64, ids [rows][2048] u32: 40 KiB per row -- 670 MB for a whole 16 384-row chunk, kept
            // for the context's life, where a handful of rows are flagged).  Blocks without a flagged row cost four
            // early-exit launches in a path that is taken once in a blue moon.
            int nl_sum = 0;
            for (int r0 = 0; r0 < m;) {
              int mb = std::min(1024, m - r0);
              if (m - (r0 + mb) < 129) mb = m - r0;   // (the grouped path wants > 128 rows: the tail joins the last block)
              int nl = 0;
              SV_TRY(sv_launch_refine_grouped(ctx, qp + (size_t)r0 * d, R, mb, d, qn + r0, rn, ctx->s_cand_cnt.as<uint32_t>() + r0,
                                              ctx->s_cand_id.as<uint32_t>() + (size_t)r0 * SV_CAP, SV_CAP, k, out_d2 + (size_t)r0 * k,
                                              out_idx + (size_t)r0 * k, &nl, rovf_rows + r0, std::min<int>((int)h_cnt[1], mb)));
              nl_sum += nl;
              r0 += mb;
            }
            sc.count(1 + nl_sum);
          } else {
            SV_TRY(sv_launch_refine_exact(ctx, qp, R, m, d, qn, rn, ctx->s_cand_cnt.as<uint32_t>(), ctx->s_cand_id.as<uint32_t>(), SV_CAP,
                                          k, out_d2, out_idx, rovf_rows));
            sc.count(2);
          }
          ctx->sstats.n_refine2 += h_cnt[1];
        }
      }
      if (last && ctx->opt.search_stats) {
        hcnt.resize(m);
        SV_HIP(hipStreamSynchronize(ctx->stream));
        SV_HIP(hipMemcpy(hcnt.data(), ctx->s_ref_cnt.p, (size_t)m * 4, hipMemcpyDeviceToHost));
        for (uint32_t c : hcnt) {
          ctx->sstats.refine_sum += c;
          if ((int64_t)c > ctx->sstats.refine_max) ctx->sstats.refine_max = c;
        }
      }
      // thresholds now hold approximate rank-th distances A_r: the exact one is <= A_r + eps, and any row at least that
      // close has d2~ <= A_r + 2 eps
      thr_ptr = thr;
      thr_ld = 1;
      eps_mult = 2.f;
    } else {
      {
        StageScope sc(ctx, "knn_gemm");
        SV_TRY(sv_launch_l2_filter(ctx, qp, R, m, (int)ns, d, qn, rn, (int)stride, thr_ptr, thr_ld, ctx->s_cand_cnt.as<uint32_t>(),
                                   ctx->s_cand_d2.as<float>(), ctx->s_cand_id.as<uint32_t>(), SV_CAP));
        sc.count();
      }
      StageScope sc(ctx, "knn_select");
      // the filter pass has consumed thr; the select may overwrite it with the tighter thresholds
      SV_TRY(sv_launch_select_cand(ctx, ctx->s_cand_cnt.as<uint32_t>(), ctx->s_cand_d2.as<float>(), ctx->s_cand_id.as<uint32_t>(), m,
                                   SV_CAP, k, last ? out_d2 : thr, last ? out_idx : nullptr, fail_rows, fail_count));
      sc.count();
      thr_ptr = thr + (k - 1);
      thr_ld = k;
      if (last && n_fail_host) {
        SV_HIP(hipMemcpyAsync(n_fail_host, fail_count, 4, hipMemcpyDeviceToHost, ctx->stream));
        SV_HIP(hipStreamSynchronize(ctx->stream));
      }
    }
  }
  return SEGVLAD_OK;
}

// rows flagged in flags[0..nrows) are redone on the exact distance-matrix path; their results replace rows of (d2, idx)
static int fallback_rows(segvlad_ctx* ctx, const SearchPlan& pl, const float* q, const float* qn, const uint32_t* flags, int nrows,
                         float* d2, int64_t* idx, int* n_done) {
  std::vector<uint32_t> hf(nrows);
  SV_HIP(hipMemcpyAsync(hf.data(), flags, (size_t)nrows * 4, hipMemcpyDeviceToHost, ctx->stream));
  SV_HIP(hipStreamSynchronize(ctx->stream));
  std::vector<int32_t> rows;
  for (int r = 0; r < nrows; ++r)
    if (hf[r]) rows.push_back(r);
  const int nf = (int)rows.size();
  *n_done = nf;
  if (nf == 0) return SEGVLAD_OK;
  const int d = pl.d, k = pl.k;
  SV_HIP(ctx->s_fb_rows.reserve((size_t)nf * 4));
  SV_HIP(ctx->s_fb_q.reserve((size_t)nf * ((size_t)d + 1) * 4));
  SV_HIP(ctx->s_fb_d2.reserve((size_t)nf * k * 4));
  SV_HIP(ctx->s_fb_idx.reserve((size_t)nf * k * 8));
  SV_HIP(hipMemcpyAsync(ctx->s_fb_rows.p, rows.data(), (size_t)nf * 4, hipMemcpyHostToDevice, ctx->stream));
  float* fq = ctx->s_fb_q.as<float>();
  float* fqn = fq + (size_t)nf * d;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(nf), dim3(256), 0, ctx->stream, q, qn, ctx->s_fb_rows.as<int32_t>(), d, fq, fqn);
  SV_HIP(hipGetLastError());
  {
    StageScope sc(ctx, "knn_fallback");   // one stage (the matrix path's own scopes are muted: no double counting)
    const bool was_muted = ctx->scope_mute;
    ctx->scope_mute = true;
    const int rc = search_matrix(ctx, fq, nf, pl.n, d, k, fqn, ctx->s_fb_d2.as<float>(), ctx->s_fb_idx.as<int64_t>());
    ctx->scope_mute = was_muted;
    SV_TRY(rc);
    sc.count(nf);
  }
  hipLaunchKernelGGL(scatter_topk_kernel, dim3(nf), dim3(256), 0, ctx->stream, ctx->s_fb_d2.as<float>(), ctx->s_fb_idx.as<int64_t>(),
                     ctx->s_fb_rows.as<int32_t>(), k, d2, idx);
  SV_HIP(hipGetLastError());
  SV_HIP(hipStreamSynchronize(ctx->stream));  // rows[] lives on this frame
  return SEGVLAD_OK;
}

int segvlad_search(segvlad_ctx* ctx, const float* Q, int nq, int k, float* d2_out, int64_t* idx_out) {
  CHECK_CTX();
  if (nq < 0 || k < 1 || k > 1024) return ctx->fail(SEGVLAD_ERR_ARG, "search: need nq>=0 and 1<=k<=1024 (k=%d)", k);
  if (nq == 0) return SEGVLAD_OK;
  if (!Q || !d2_out || !idx_out) return ctx->fail(SEGVLAD_ERR_ARG, "search: null pointer");
  if (ctx->db_d == 0) return ctx->fail(SEGVLAD_ERR_STATE, "search: the index is empty and has no dimension yet");
  if (ctx->opt.debug_fail_search == 1) return ctx->fail(SEGVLAD_ERR_STATE, "search: failing on request (option debug_fail_search)");
  const int d = ctx->db_d;
  const int64_t n = ctx->db_n;
  ctx->f16_scale_dev = nullptr;
  const void* dq;
  void *dd2, *didx;
  SV_TRY(sv_in(ctx, Q, (size_t)nq * d * 4, &dq));
  SV_TRY(sv_out(ctx, d2_out, (size_t)nq * k * 4, &dd2));
  SV_TRY(sv_out(ctx, idx_out, (size_t)nq * k * 8, &didx));
  SV_HIP(ctx->s_qnorm.reserve((size_t)nq * 4));
  const float* qn = ctx->s_qnorm.as<float>();
  // the queries' squared norms: a launch of their own -- except on the single-image fp16 path, whose query-preparation kernel
  // computes them too (one dependent launch less in a pass that is a chain of them)
  bool qn_done = false;
  auto ensure_qn = [&]() -> int {
    if (!qn_done) SV_TRY(sv_launch_row_sumsq(ctx, (const float*)dq, nq, d, ctx->s_qnorm.as<float>()));
    qn_done = true;
    return SEGVLAD_OK;
  };
  ctx->sstats = SvSearchStats();
  ctx->sstats.n_queries = nq;
  ctx->tail_stats_dev = nullptr;
  ctx->small_head_ran = false;
  // device-driven single-image passes never tell the host how many rows they had to redo -- but their running total lands in a
  // pinned word (small_tail_kernel): looked at here WITHOUT synchronising (it may be a pass or two behind).  More than a quarter
  // of >= 64 rows redone since the index last changed: the sample misleads on this database, stop guessing (as the read-back path
  // decides below for its own redos).
  if (ctx->h_pin && ctx->tail_rows_since >= 64) {
    const uint32_t redone = reinterpret_cast<volatile uint32_t*>(ctx->h_pin)[8] - ctx->tail_fail_base;
    if ((int64_t)redone * 4 > ctx->tail_rows_since) ctx->db_heur_off = true;
  }

  // level plan: strides 16^L, ..., 16, 1.  The coarsest sample goes through the exact fp32 matrix path (an order of
  // magnitude dearer per row than the fp16 filter), so take as many levels as keep it selective: a sample of s rows
  // admits a fraction k/s of the next level, which must stay below the filter's per-wave list capacity (25 % of a
  // block) -> s >= 4.8 k.  Databases of <= 32768 rows keep the plain matrix path; the sample never exceeds 32768 rows.
  // (Row shards of 250 k - 500 k rows -- the 1 M-row database on 2 or 4 GPUs -- get two levels instead of a 15 k - 31 k
  // row exact level.)
  SearchPlan pl;
  pl.d = d;
  pl.k = k;
  pl.n = n;
  if (n > 32768) {
    const int64_t want = std::max<int64_t>((24 * (int64_t)k + 4) / 5, 512);
    while (n / (pl.stride0 * SV_RATIO) >= want) {
      pl.stride0 *= SV_RATIO;
      ++pl.levels;
    }
    while (n / pl.stride0 > 32768) {
      pl.stride0 *= SV_RATIO;
      ++pl.levels;
    }
  }
  if (pl.levels == 0 || n / pl.stride0 < 4 * (int64_t)k) {
    SV_TRY(ensure_qn());
    SV_TRY(search_matrix(ctx, (const float*)dq, nq, n, d, k, qn, (float*)dd2, (int64_t*)didx));
    return sv_finish(ctx);
  }
  const float* R = ctx->db_rows.as<float>();
  const float* rn = ctx->db_norms.as<float>();
  // filter arithmetic (ctx->opt.knn_filter, see SvOptions): "f16" = one fp16 product (d % 64 == 0), "bf16x3" = three
  // bf16 products (d % 32 == 0), else plain fp32
  const int want_f = ctx->opt.knn_filter;
  const bool f16_path = (want_f == 0 || want_f == 1) && (d % 64 == 0);
  const bool bf16_path = !f16_path && want_f != 3 && (d % 32 == 0);
  pl.kind = f16_path ? 1 : bf16_path ? 2 : 3;
  ctx->sstats.filter = pl.kind;
  // The low-rank ("heuristic") thresholds need only a few dozen sample rows per rank, so their plan goes deeper: the
  // exact fp32 level (an order of magnitude dearer per row than a filter level) shrinks to >= 192 rows -- 1 M rows:
  // strides 4096, 256, 16, 1 instead of 256, 16, 1 (0.9 ms of exact GEMM per 10 000 queries -> 0.06 ms); a 125 k-row
  // shard: 256, 16, 1 instead of 16, 1 (1.35 -> 0.1 ms).  The rigorous redo keeps the plan above.
  const bool heuristic = pl.kind != 3 && ctx->opt.knn_heuristic && !ctx->db_heur_off && heur_rank(k) < k;
  SearchPlan plh = pl;
  // One query image per pass (<= 128 rows) is bound by its chain of dependent launches, not by the exact level's flops
  // (option small_plan): ONE filter level behind an exact sample of 2048..4096 rows (stride = the power of two that gives
  // it: 256 for 1 M rows, 64 for a 250 k-row shard).  The threshold is a low rank of that sample (heur_rank_small: ~7 at
  // stride 256), the full level collects stride x rank candidates per query (~1800 + the margin's ~900 at 1 M rows: a
  // workgroup select's worth, SV_CAP bounds it -- hence stride <= 512), and the deeper plan's two sampled filter launches
  // with their selects (~90 us of a ~500 us pass) are gone.  (The same last step for batches -- strides 4096, 256, 1 instead
  // of 4096, 256, 16, 1 -- was measured and dropped: the stride-16 level's 1.3 ms come back as epilogue time of the full
  // level, which then collects 2900 candidates per query instead of 780, and as 0.2 ms of longer selects.)
  int small_stride = 16;
  while ((n + small_stride - 1) / small_stride > 4096 && small_stride <= 512) small_stride *= 2;
  bool small_taken = false;
  if (heuristic && nq <= 128 && ctx->opt.small_plan && small_stride <= 512) {
    small_taken = true;
    plh.levels = 1;
    plh.stride0 = small_stride;
    plh.ratio_last = small_stride;
  } else if (heuristic) {
    while (plh.levels < 6 && n / (plh.stride0 * SV_RATIO) >= 192) {
      plh.stride0 *= SV_RATIO;
      ++plh.levels;
    }
  }
  ctx->sstats.levels = plh.levels;
  auto grow = [&](DevBuf& b, size_t old_bytes, size_t new_bytes) -> hipError_t {
    if (new_bytes <= b.cap) return hipSuccess;
    DevBuf nb;
    nb.tag = b.tag;
    nb.guard = b.guard;
    nb.fixed = b.fixed;
    hipError_t e = nb.reserve(new_bytes + new_bytes / 4);
    if (e != hipSuccess) return e;
    if (old_bytes) {
      e = hipMemcpyAsync(nb.p, b.p, old_bytes, hipMemcpyDeviceToDevice, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess) return e;
    }
    b.release();
    b = nb;
    return hipSuccess;
  };
  // power-of-two scales that put the largest magnitude in [8192, 16384): no overflow, negligible underflow
  auto pow2_scale = [](float maxabs) -> float {
    if (!(maxabs > 0.f) || !std::isfinite(maxabs)) return 1.f;
    int e;
    frexpf(maxabs, &e);  // maxabs = m * 2^e, m in [0.5, 1)
    return ldexpf(1.f, 14 - e);
  };
  float qscale = 1.f;
  // per-query scratch is sized for the rows of one chunk -- min(nq, SV_CHUNK), not SV_CHUNK: a one-image search on a large
  // index used to allocate > 1 GiB of candidate lists (the buffers only ever grow, so a later large batch re-allocates once)
  const size_t mrows = (size_t)std::min(nq, SV_CHUNK);
  // flag block: [nq] row flags, 2 counts (flagged, second-tier), [mrows] second-tier flags of the current chunk -- one fill
  // (+ 4 words behind them: the device-driven tail's counters of this search)
  const size_t ovf_bytes = (((size_t)nq + 2 + mrows + 4) * 4 + 255) & ~(size_t)255;   // (a whole number of 256-byte blocks: one fill kernel)
  bool ovf_zeroed = false;
  // the per-chunk scratch and the flag block, reserved (and the flags zeroed) once -- by the early exact level of a batch
  // search (below) or in front of the chunk loop
  bool scratch_ready = false, level0_done = false;
  const bool heuristic_pre = heuristic;
  SearchPlan plh_pre = plh;
  auto reserve_scratch = [&]() -> int {
    if (scratch_ready) return SEGVLAD_OK;
    SV_HIP(ctx->s_cand_cnt.reserve(mrows * 4));
    SV_HIP(ctx->s_cand_d2.reserve(mrows * SV_CAP * 4));
    SV_HIP(ctx->s_cand_id.reserve(mrows * SV_CAP * 4));
    SV_HIP(ctx->s_thr_d2.reserve(mrows * k * 4));
    SV_HIP(ctx->s_thr_idx.reserve(mrows * k * 8));
    // the exact level's distance block: the larger of the two plans' samples (the single-image plan's stride can be SMALLER than
    // the rigorous plan's: 64 against 256 for a 250 k-row shard)
    const int64_t stride_min = heuristic ? std::min(pl.stride0, plh.stride0) : pl.stride0;
    const int64_t n0 = (n + stride_min - 1) / stride_min;
    SV_HIP(ctx->s_dist.reserve(mrows * ((n0 + 3) & ~3ll) * 4));
    // layout: [nq] row flags, 2 counts (flagged, second-tier), [mrows] second-tier flags of the current chunk -- one memset
    SV_HIP(ctx->s_ovf.reserve(ovf_bytes));
    if (!ovf_zeroed) SV_HIP(hipMemsetAsync(ctx->s_ovf.p, 0, ovf_bytes, ctx->stream));
    ovf_zeroed = true;
    scratch_ready = true;
    return SEGVLAD_OK;
  };
  if (f16_path || bf16_path) {
    if (ctx->db_rn_max_rows < n) {
      float m = 0.f;
      SV_TRY(sv_row_norm_max(ctx, rn + ctx->db_rn_max_rows, n - ctx->db_rn_max_rows, &m));
      if (m > ctx->db_rn_max) ctx->db_rn_max = m;
      ctx->db_rn_max_rows = n;
    }
    pl.rn_max = ctx->db_rn_max;
    SV_HIP(ctx->s_ref_cnt.reserve((size_t)mrows * 4));
    SV_HIP(ctx->s_ref_id.reserve((size_t)mrows * SV_RCAP * 4));
  }
  if (f16_path) {
    if (ctx->db_f16_rows < n) {
      float m_new = 0.f;
      SV_TRY(sv_maxabs(ctx, R + (size_t)ctx->db_f16_rows * d, (n - ctx->db_f16_rows) * d, &m_new));
      const bool rescale = ctx->db_f16_rows == 0 || m_new * ctx->db_f16_scale >= 32768.f;
      if (m_new > ctx->db_maxabs) ctx->db_maxabs = m_new;
      int64_t r0 = ctx->db_f16_rows;
      if (rescale) {
        ctx->db_f16_scale = pow2_scale(ctx->db_maxabs);
        r0 = 0;
      }
      SV_HIP(grow(ctx->db_f16, (size_t)r0 * d * 2, (size_t)n * d * 2));
      SV_TRY(sv_launch_to_f16(ctx, R + (size_t)r0 * d, (n - r0) * d, ctx->db_f16_scale, ctx->db_f16.as<uint16_t>() + (size_t)r0 * d));
      ctx->db_f16_rows = n;
    }
    SV_HIP(ctx->s_qf16.reserve((size_t)nq * d * 2));
    // batches on the default configuration take the biased-accumulator kernel when the norms allow it (below): min ||q||^2
    const bool want_q2min = nq > 128 && (ctx->opt.f16_cfg < 0 || ctx->opt.f16_cfg == 250 || ctx->opt.f16_cfg == 300);
    float q2min_pre = 0.f;
    bool have_q2min = false;
    if (nq <= 128 && (((int64_t)nq * d) & 3) == 0 && (int64_t)nq * d <= (1 << 18)) {
      // one query image per pass: the scale is computed AND consumed on the device (a host round trip in front of every pass
      // was ~45 us of a ~600 us call).  One workgroup reads the block twice: up to 1 MiB of queries (128 x 2048 floats);
      // deeper rows (raw K*D descriptors) keep the many-workgroup kernels and their read-back.
      SV_HIP(ctx->s_qscale.reserve(16));
      const bool fuse_qn = !qn_done && (reinterpret_cast<uintptr_t>(dq) & 15) == 0;   // (d % 64 == 0 on this path)
      SV_HIP(ctx->s_ovf.reserve(ovf_bytes));
      // round 6: the whole head of a single-image pass in ONE launch (small_pass_kernels.hip) -- plane, scale, norms, flags AND the
      // sample thresholds (from the filter's own fp16 product on the strided sample: the thresholds are guesses the pass verifies,
      // they need no exact distances) -- instead of query preparation -> exact sample GEMM -> reduce + rank (three dependent launches)
      const int64_t n0_small = (n + small_stride - 1) / small_stride;
      const int r0_small = std::min(k, small_stride == SV_RATIO ? heur_rank(k) : heur_rank_small(k, small_stride, plh.pfail));   // (levels_chunk's rank[0])
      if (small_taken && fuse_qn && ctx->opt.small_head && ctx->opt.debug_search == 0 && n0_small <= 4096 &&
          sv_small_head_ok(nq, d, (int)n0_small, r0_small)) {
        ovf_zeroed = true;
        SV_TRY(reserve_scratch());
        StageScope sc(ctx, "knn_level0");
        SV_TRY(sv_launch_small_head(ctx, (const float*)dq, nq, d, ctx->db_f16.as<uint16_t>(), rn, small_stride, (int)n0_small,
                                    ctx->db_f16_scale, r0_small, ctx->s_qf16.as<uint16_t>(), ctx->s_qscale.as<float>(),
                                    ctx->s_qnorm.as<float>(), ctx->s_ovf.as<uint32_t>(), (int)(ovf_bytes / 4), ctx->s_dist.as<float>(),
                                    ctx->s_thr_d2.as<float>(), ctx->s_cand_cnt.as<uint32_t>()));
        sc.count();
        level0_done = true;
      } else {
      SV_TRY(sv_launch_query_f16_small(ctx, (const float*)dq, (int64_t)nq * d, ctx->db_f16_scale, ctx->s_qf16.as<uint16_t>(),
                                       ctx->s_qscale.as<float>(), fuse_qn ? ctx->s_qnorm.as<float>() : nullptr, nq, d,
                                       ctx->s_ovf.as<uint32_t>(), (int)(ovf_bytes / 4)));
      }
      ovf_zeroed = true;   // (the flag block of this search: no fill launch of its own in front of the pass)
      if (fuse_qn) qn_done = true;
      ctx->f16_scale_dev = ctx->s_qscale.as<float>();
      pl.inv_scale = 0.f;   // (unused: the kernels read s_qscale[1])
    } else {
      SV_TRY(ensure_qn());
      float qmax = 0.f;
      if (want_q2min) {   // both scalars behind one read-back
        SV_TRY(sv_maxabs_and_norm_min_begin(ctx, (const float*)dq, (int64_t)nq * d, qn, nq));
        // the exact sample level of the first chunk goes out BEFORE the host waits for the two scalars: it needs neither the
        // query plane nor the margin they fix, and the device runs it while the host round trip is under way (round 5; the
        // wait used to leave the device idle in front of every batch search)
        // (not when the sampled level runs on the fp16 product -- batch_l0_f16: that needs the query plane the scalars fix; it is
        //  ten times cheaper than the exact GEMM it replaces, which is worth more than hiding that GEMM under the host's wait)
        const int64_t n0_pre = (n + plh_pre.stride0 - 1) / plh_pre.stride0;
        const bool l0_f16 = ctx->opt.batch_l0_f16 && n0_pre <= 4096;
        if (heuristic_pre && nq > 128 && ctx->opt.debug_search == 0 && !l0_f16) {
          SV_TRY(reserve_scratch());
          SV_TRY(levels_chunk(ctx, plh_pre, true, (const float*)dq, nullptr, nullptr, qn, std::min(nq, SV_CHUNK), nullptr, nullptr,
                              ctx->s_ovf.as<uint32_t>(), ctx->s_ovf.as<uint32_t>() + nq, nullptr, nullptr, nullptr, 1));
          level0_done = true;
        }
        SV_TRY(sv_maxabs_and_norm_min_end(ctx, (int64_t)nq * d, nq, &qmax, &q2min_pre));
        have_q2min = true;
      } else {
        SV_TRY(sv_maxabs(ctx, (const float*)dq, (int64_t)nq * d, &qmax));
      }
      qscale = pow2_scale(qmax);
      SV_TRY(sv_launch_to_f16(ctx, (const float*)dq, (int64_t)nq * d, qscale, ctx->s_qf16.as<uint16_t>()));
      pl.inv_scale = 1.f / (qscale * ctx->db_f16_scale);
    }
    // |d2~ - d2| <= c_eps ||q|| ||r||: ctx.h, sv_f16_c_eps (the constant of the kernel variant that will run).  Batches
    // (> 128 queries, default configuration) take the biased-accumulator kernel when the norms are balanced enough for
    // its margin: bias_mult = 1 + max||r|| / (2 min||q||), 1.5 for unit vectors
    float bias_mult = 1.f;
    ctx->f16_bias_ok = false;
    if (want_q2min) {   // (deep rows too: round 4)
      float q2min = q2min_pre;
      if (!have_q2min) SV_TRY(sv_row_norm_min(ctx, qn, nq, &q2min));
      if (q2min > 0.f && pl.rn_max > 0.f) {
        const float bm = 1.f + std::sqrt(pl.rn_max) / (2.f * std::sqrt(q2min));
        if (std::isfinite(bm) && bm <= 5.f) {
          bias_mult = bm;
          ctx->f16_bias_ok = true;
        }
      }
    }
    pl.c_eps = sv_f16_c_eps(d, sv_f16_eps_kblock(ctx->opt, d), bias_mult);
  } else if (bf16_path) {
    // lazily extend the bf16 planes to the rows added since the last search
    if (ctx->db_split_rows < n) {
      SV_HIP(grow(ctx->db_hi, (size_t)ctx->db_split_rows * d * 2, (size_t)n * d * 2));
      SV_HIP(grow(ctx->db_lo, (size_t)ctx->db_split_rows * d * 2, (size_t)n * d * 2));
      const int64_t r0 = ctx->db_split_rows;
      SV_TRY(sv_launch_split_bf16(ctx, R + (size_t)r0 * d, (n - r0) * d, ctx->db_hi.as<uint16_t>() + (size_t)r0 * d,
                                  ctx->db_lo.as<uint16_t>() + (size_t)r0 * d));
      ctx->db_split_rows = n;
    }
    // |d2~ - d2| <= 2 * (3*2^-16 + 4*d*2^-24) * ||q|| * ||r||   (+25 % slack)
    pl.c_eps = 2.5f * (3.f / 65536.f + 4.f * (float)d / 16777216.f);
    SV_HIP(ctx->s_qh.reserve((size_t)nq * d * 2));
    SV_HIP(ctx->s_ql.reserve((size_t)nq * d * 2));
    SV_TRY(sv_launch_split_bf16(ctx, (const float*)dq, (int64_t)nq * d, ctx->s_qh.as<uint16_t>(), ctx->s_ql.as<uint16_t>()));
  }
  SV_TRY(ensure_qn());   // (every path that has not produced the norms on its way)
  if (ovf_zeroed && !scratch_ready) {   // (the single-image query preparation zeroed the flag block in its own launch)
    SV_HIP(ctx->s_ovf.reserve(ovf_bytes));
  }
  SV_TRY(reserve_scratch());
  // Flags: [nq] rows + 1 count.  A heuristic pass flags the queries whose low-rank thresholds did not verify (-> redo
  // with the rigorous thresholds, below); a rigorous pass flags list overflows (-> exact distance-matrix path, alone).
  plh.c_eps = pl.c_eps;
  plh.inv_scale = pl.inv_scale;
  plh.rn_max = pl.rn_max;
  uint32_t* flag_rows = ctx->s_ovf.as<uint32_t>();
  uint32_t* flag_count = flag_rows + nq;
  uint32_t* rovf_flags = flag_count + 2;
  uint32_t n_rovf_seen = 0;
  auto plane_a = [&](const DevBuf& f16, const DevBuf& hi, int64_t q0) -> const uint16_t* {
    return (pl.kind == 1 ? reinterpret_cast<const uint16_t*>(f16.p) : reinterpret_cast<const uint16_t*>(hi.p)) + (size_t)q0 * d;
  };
  uint32_t n_flag = 0;   // running count of the flagged rows, read back by every chunk (levels_chunk synchronises once)
  // one query image per pass on the single-image plan with the scale on the device: the pass ends in small_tail_kernel instead of
  // a read-back (option small_tail = 0: the round-5 path, for A/B and for the tests that compare the two)
  uint32_t* tail_stats = nullptr;
  if (small_taken && pl.kind == 1 && ctx->f16_scale_dev && ctx->opt.small_tail &&
      ctx->opt.debug_search == 0 && n <= 0xffffffffLL)
    tail_stats = rovf_flags + mrows;
  for (int q0 = 0; q0 < nq; q0 += SV_CHUNK) {
    const int m = (nq - q0 < SV_CHUNK) ? (nq - q0) : SV_CHUNK;
    if (q0) SV_HIP(hipMemsetAsync(rovf_flags, 0, (size_t)m * 4, ctx->stream));
    SV_TRY(levels_chunk(ctx, heuristic ? plh : pl, heuristic, (const float*)dq + (size_t)q0 * d,
                        pl.kind == 3 ? nullptr : plane_a(ctx->s_qf16, ctx->s_qh, q0),
                        pl.kind == 2 ? ctx->s_ql.as<uint16_t>() + (size_t)q0 * d : nullptr, qn + q0, m, (float*)dd2 + (size_t)q0 * k,
                        (int64_t*)didx + (size_t)q0 * k, flag_rows + q0, flag_count, rovf_flags, &n_flag, &n_rovf_seen,
                        (q0 == 0 && level0_done) ? 2 : 0, tail_stats));
  }
  if (tail_stats) {
    ctx->tail_stats_dev = tail_stats;
    ctx->tail_rows_since += nq;
  }
  if (n_flag && !heuristic) {
    int nf = 0;
    SV_TRY(fallback_rows(ctx, pl, (const float*)dq, qn, flag_rows, nq, (float*)dd2, (int64_t*)didx, &nf));
    ctx->sstats.n_fallback = nf;
  } else if (n_flag) {
    // ---- redo of the flagged queries with the rigorous thresholds, as one dense batch ---------------------------------
    std::vector<uint32_t> hf(nq);
    SV_HIP(hipMemcpy(hf.data(), flag_rows, (size_t)nq * 4, hipMemcpyDeviceToHost));
    std::vector<int32_t> rows;
    for (int q = 0; q < nq; ++q)
      if (hf[q]) rows.push_back(q);
    const int nr = (int)rows.size();
    ctx->sstats.n_redo = nr;
    if ((int64_t)nr * 4 > nq && nq >= 64) ctx->db_heur_off = true;   // the sample misleads on this database: stop guessing
    SV_HIP(ctx->s_rd_rows.reserve((size_t)nr * 4));
    SV_HIP(ctx->s_rd_q.reserve((size_t)nr * ((size_t)d + 1) * 4));
    SV_HIP(ctx->s_rd_d2.reserve((size_t)nr * k * 4));
    SV_HIP(ctx->s_rd_idx.reserve((size_t)nr * k * 8));
    const size_t rd_m = (size_t)std::min(nr, SV_CHUNK);
    SV_HIP(ctx->s_rd_flags.reserve(((size_t)nr + 2 + rd_m) * 4));   // (same layout as s_ovf, without the tail's counters)
    SV_HIP(hipMemsetAsync(ctx->s_rd_flags.p, 0, ((size_t)nr + 2 + rd_m) * 4, ctx->stream));
    SV_HIP(hipMemcpyAsync(ctx->s_rd_rows.p, rows.data(), (size_t)nr * 4, hipMemcpyHostToDevice, ctx->stream));
    float* rq = ctx->s_rd_q.as<float>();
    float* rqn = rq + (size_t)nr * d;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(nr), dim3(256), 0, ctx->stream, (const float*)dq, qn, ctx->s_rd_rows.as<int32_t>(), d, rq,
                       rqn);
    SV_HIP(hipGetLastError());
    SV_HIP(ctx->s_rd_p1.reserve((size_t)nr * d * 2));
    if (pl.kind == 1) {
      if (ctx->f16_scale_dev)   // same scale as the main pass
        SV_TRY(sv_launch_to_f16_devscale(ctx, rq, (int64_t)nr * d, ctx->f16_scale_dev, ctx->s_rd_p1.as<uint16_t>()));
      else
        SV_TRY(sv_launch_to_f16(ctx, rq, (int64_t)nr * d, qscale, ctx->s_rd_p1.as<uint16_t>()));
    } else {
      SV_HIP(ctx->s_rd_p2.reserve((size_t)nr * d * 2));
      SV_TRY(sv_launch_split_bf16(ctx, rq, (int64_t)nr * d, ctx->s_rd_p1.as<uint16_t>(), ctx->s_rd_p2.as<uint16_t>()));
    }
    uint32_t* rflags = ctx->s_rd_flags.as<uint32_t>();
    uint32_t n_ovf = 0, rd_rovf_seen = 0;
    for (int q0 = 0; q0 < nr; q0 += SV_CHUNK) {
      const int m = (nr - q0 < SV_CHUNK) ? (nr - q0) : SV_CHUNK;
      if (q0) SV_HIP(hipMemsetAsync(rflags + nr + 2, 0, (size_t)m * 4, ctx->stream));
      StageScope sc(ctx, "knn_redo");   // the whole redo is ONE stage: its inner level / filter / select scopes are muted,
      ctx->scope_mute = true;           // so "knn_gemm" etc. keep describing the main pass only (no double counting)
      const int rc = levels_chunk(ctx, pl, false, rq + (size_t)q0 * d, ctx->s_rd_p1.as<uint16_t>() + (size_t)q0 * d,
                                  pl.kind == 2 ? ctx->s_rd_p2.as<uint16_t>() + (size_t)q0 * d : nullptr, rqn + q0, m,
                                  ctx->s_rd_d2.as<float>() + (size_t)q0 * k, ctx->s_rd_idx.as<int64_t>() + (size_t)q0 * k, rflags + q0,
                                  rflags + nr, rflags + nr + 2, &n_ovf, &rd_rovf_seen);
      ctx->scope_mute = false;
      SV_TRY(rc);
      sc.count(m);
    }
    SV_HIP(hipStreamSynchronize(ctx->stream));   // rows[] lives on this frame
    if (n_ovf) {
      int nf = 0;
      SV_TRY(fallback_rows(ctx, pl, rq, rqn, rflags, nr, ctx->s_rd_d2.as<float>(), ctx->s_rd_idx.as<int64_t>(), &nf));
      ctx->sstats.n_fallback = nf;
    }
    hipLaunchKernelGGL(scatter_topk_kernel, dim3(nr), dim3(256), 0, ctx->stream, ctx->s_rd_d2.as<float>(), ctx->s_rd_idx.as<int64_t>(),
                       ctx->s_rd_rows.as<int32_t>(), k, (float*)dd2, (int64_t*)didx);
    SV_HIP(hipGetLastError());
  }
  return sv_finish(ctx);
}

int segvlad_merge_topk(segvlad_ctx* ctx, const float* d2_parts, const int64_t* idx_parts, int nq, int parts, int k,
                       float* d2_out, int64_t* idx_out) {
  CHECK_CTX();
  if (nq < 0 || parts < 1 || k < 1) return ctx->fail(SEGVLAD_ERR_ARG, "merge_topk: bad shape");
  if (nq == 0) return SEGVLAD_OK;
  if (!d2_parts || !idx_parts || !d2_out || !idx_out) return ctx->fail(SEGVLAD_ERR_ARG, "merge_topk: null pointer");
  const int cand = parts * k;
  const void *dd, *di;
  void *od, *oi;
  SV_TRY(sv_in(ctx, d2_parts, (size_t)nq * cand * 4, &dd));
  SV_TRY(sv_in(ctx, idx_parts, (size_t)nq * cand * 8, &di));
  SV_TRY(sv_out(ctx, d2_out, (size_t)nq * k * 4, &od));
  SV_TRY(sv_out(ctx, idx_out, (size_t)nq * k * 8, &oi));
  SV_TRY(sv_launch_merge_topk(ctx, (const float*)dd, (const int64_t*)di, nq, cand, k, (float*)od, (int64_t*)oi));
  return sv_finish(ctx);
}

int segvlad_sims_from_d2(segvlad_ctx* ctx, const float* d2, const int64_t* idx, int nq, int k_in, int k_keep,
                         float* sims_out, int64_t* idx_out) {
  CHECK_CTX();
  if (nq < 0 || k_in < 1 || k_keep < 1 || k_keep > k_in) return ctx->fail(SEGVLAD_ERR_ARG, "sims_from_d2: bad shape");
  if (nq == 0) return SEGVLAD_OK;
  if (!d2 || !idx || !sims_out || !idx_out) return ctx->fail(SEGVLAD_ERR_ARG, "sims_from_d2: null pointer");
  const void *dd, *di;
  void *os, *oi;
  SV_TRY(sv_in(ctx, d2, (size_t)nq * k_in * 4, &dd));
  SV_TRY(sv_in(ctx, idx, (size_t)nq * k_in * 8, &di));
  SV_TRY(sv_out(ctx, sims_out, (size_t)nq * k_keep * 4, &os));
  SV_TRY(sv_out(ctx, idx_out, (size_t)nq * k_keep * 8, &oi));
  SV_TRY(sv_launch_sims(ctx, (const float*)dd, (const int64_t*)di, nq, k_in, k_keep, (float*)os, (int64_t*)oi));
  return sv_finish(ctx);
}

int segvlad_minmax(segvlad_ctx* ctx, const float* sims, int64_t count, float* minmax_out) {
  CHECK_CTX();
  if (count < 0 || !minmax_out) return ctx->fail(SEGVLAD_ERR_ARG, "minmax: bad arguments");
  const void* ds = sims;
  void* dout;
  if (count > 0) {
    if (!sims) return ctx->fail(SEGVLAD_ERR_ARG, "minmax: null sims");
    SV_TRY(sv_in(ctx, sims, (size_t)count * 4, &ds));
  }
  SV_TRY(sv_out(ctx, minmax_out, 2 * sizeof(float), &dout));
  SV_TRY(sv_launch_minmax(ctx, (const float*)ds, count, (float*)dout));
  return sv_finish(ctx);
}

int segvlad_vote(segvlad_ctx* ctx, const int64_t* idx, const float* sims, const int32_t* img_of_seg,
                 int64_t n_ref_seg, const int32_t* qseg_offsets, int n_img, int k, float smin, float smax, int n_top, int mode,
                 int32_t* pred_out, double* score_out) {
  CHECK_CTX();
  if (n_img < 0 || k < 1 || n_top < 1) return ctx->fail(SEGVLAD_ERR_ARG, "vote: bad shape");
  if (mode != SEGVLAD_VOTE_WT_BORDA_IM && mode != SEGVLAD_VOTE_COUNT) return ctx->fail(SEGVLAD_ERR_ARG, "vote: unknown mode %d", mode);
  if (n_img == 0) return SEGVLAD_OK;
  if (!idx || !qseg_offsets || !pred_out) return ctx->fail(SEGVLAD_ERR_ARG, "vote: null pointer");
  if (mode == SEGVLAD_VOTE_WT_BORDA_IM && !sims) return ctx->fail(SEGVLAD_ERR_ARG, "vote: weighted mode needs sims");
  if (sv_is_device_ptr(qseg_offsets)) return ctx->fail(SEGVLAD_ERR_ARG, "vote: qseg_offsets must be host memory");
  const int nq = qseg_offsets[n_img];
  int64_t n_ref = ctx->db_n;
  const void* dimg = nullptr;
  if (img_of_seg) {
    if (n_ref_seg <= 0) return ctx->fail(SEGVLAD_ERR_ARG, "vote: an explicit img_of_seg map needs n_ref_seg > 0");
    n_ref = n_ref_seg;
    SV_TRY(sv_in(ctx, img_of_seg, (size_t)n_ref * 4, &dimg));
  } else {
    if (!ctx->db_has_img) return ctx->fail(SEGVLAD_ERR_STATE, "vote: no img_of_seg map: give it to segvlad_db_add");
    dimg = ctx->db_img.p;
  }
  const void *di, *ds = nullptr;
  void *op, *os = nullptr;
  SV_TRY(sv_in(ctx, idx, (size_t)nq * k * 8, &di));
  if (sims) SV_TRY(sv_in(ctx, sims, (size_t)nq * k * 4, &ds));
  SV_TRY(sv_out(ctx, pred_out, (size_t)n_img * n_top * 4, &op));
  if (score_out) SV_TRY(sv_out(ctx, score_out, (size_t)n_img * n_top * 8, &os));
  SV_HIP(ctx->s_voteoff.reserve((size_t)(n_img + 1) * 4 + 32));
  SV_HIP(hipMemcpyAsync(ctx->s_voteoff.p, qseg_offsets, (size_t)(n_img + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  float* mm = reinterpret_cast<float*>(ctx->s_voteoff.as<char>() + (((size_t)(n_img + 1) * 4 + 7) & ~7ull));
  StageScope sc(ctx, "vote");
  if (mode == SEGVLAD_VOTE_WT_BORDA_IM) {
    if (std::isnan(smin) || std::isnan(smax)) {
      SV_TRY(sv_launch_minmax(ctx, (const float*)ds, (int64_t)nq * k, mm));
      sc.count(3);
    } else {
      const float h[2] = {smin, smax};
      SV_HIP(hipMemcpyAsync(mm, h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
      SV_HIP(hipStreamSynchronize(ctx->stream));
    }
  }
  SV_TRY(sv_launch_vote(ctx, (const int64_t*)di, (const float*)ds, (const int32_t*)dimg, n_ref, ctx->s_voteoff.as<int32_t>(),
                        qseg_offsets, n_img, k, mm, n_top, mode, (int32_t*)op, (double*)os));
  sc.count();
  return sv_finish(ctx);
}

}  // extern "C"
