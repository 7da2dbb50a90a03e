// demi_gpu.hip — C-ABI host side of libdemi_gpu.so (see include/demi_gpu.h).
// Owns the device copies of the transition table and the external-event trace, validates what
// crosses the boundary, and launches the gfx950 kernels.  No CPU execution path exists here: if
// HIP is unavailable every entry point fails with DEMI_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <memory>
#include <queue>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/demi_gpu.h"
#include "k1_random_explore.hpp"
#include "k2_replay.hpp"
#include "k3_dpor.hpp"
#include "k_collect.hpp"
#include "dpor_host.hpp"

using namespace demi;

struct demi_ctx {
  int device = 0;
  int num_cu = 0;
  std::string err;
  // model
  bool have_model = false;
  DevModel hmodel;
  DevModel* d_model = nullptr;
  // trace
  bool have_trace = false;
  std::vector<demi_ext_event> trace;
  uint64_t* d_trace = nullptr;
  uint32_t started_mask = 0;
  uint32_t n_batches = 1;           // injection batches: WaitQuiescence events + 1
  // scratch
  unsigned long long* d_counter = nullptr;
  demi_verdict* d_out = nullptr;
  size_t out_cap = 0;
  uint64_t* d_seeds = nullptr;
  size_t seeds_cap = 0;
  demi_rec_event* d_rec = nullptr;
  uint32_t* d_rec_count = nullptr;
  uint32_t* d_spill = nullptr;
  size_t spill_bytes = 0;
  // K2 (replay of an original execution)
  bool have_replay = false;
  uint64_t* d_rext = nullptr;       // original externals
  uint32_t n_rext = 0;
  uint64_t* d_expected = nullptr;   // lowered original trace
  uint32_t n_expected = 0;
  uint32_t replay_spawned = 0;      // actors with a SpawnEvent in the original trace
  uint64_t* d_masks = nullptr;
  size_t masks_cap = 0;
  std::vector<int32_t> exp_of_rec;  // recorded-event index -> index in d_expected (-1: not lowered, a nop in replay)
  std::vector<uint8_t> rec_is_delivery;  // recorded event is a MsgEvent / TimerDelivery
  uint32_t* d_skip = nullptr;       // removal candidates (demi_replay_removal_batch)
  size_t skip_cap = 0;
  uint8_t* d_kept = nullptr;        // executed-trace marks of one candidate (demi_replay_get_kept)
  // K3 (DPOR)
  bool have_dpor = false;
  uint64_t* d_dext = nullptr;
  uint32_t n_dext = 0;
  void* d_dpor = nullptr;           // one arena for a batch's inputs and outputs
  size_t dpor_bytes = 0;
  demi_violation* d_viol = nullptr; // demi_random_explore_violations: list
  unsigned long long* d_viol_count = nullptr;
  size_t viol_cap = 0;
};

static int fail(demi_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

#define HIP_TRY(ctx, expr)                                                                       \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) return fail(ctx, DEMI_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

extern "C" const char* demi_version(void) { return "demi_gpu 0.1 (gfx950)"; }

extern "C" const char* demi_last_error(const demi_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

extern "C" int demi_ctx_create(int device_ordinal, demi_ctx** out) {
  if (!out) return DEMI_ERR_INVALID_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_ordinal < 0 || device_ordinal >= n)
    return DEMI_ERR_DEVICE;  // the product path has no CPU fallback
  demi_ctx* ctx = new demi_ctx();
  ctx->device = device_ordinal;
  if (hipSetDevice(device_ordinal) != hipSuccess) { delete ctx; return DEMI_ERR_DEVICE; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) { delete ctx; return DEMI_ERR_DEVICE; }
  ctx->num_cu = prop.multiProcessorCount;
  if (hipMalloc(&ctx->d_model, sizeof(DevModel)) != hipSuccess ||
      hipMalloc(&ctx->d_trace, sizeof(uint64_t) * (DEMI_MAX_EXT_EVENTS + 1)) != hipSuccess ||
      hipMalloc(&ctx->d_counter, sizeof(unsigned long long)) != hipSuccess ||
      hipMalloc(&ctx->d_viol_count, sizeof(unsigned long long)) != hipSuccess ||
      hipMalloc(&ctx->d_rec, sizeof(demi_rec_event) * DEMI_MAX_REC_EVENTS) != hipSuccess ||
      hipMalloc(&ctx->d_rec_count, sizeof(uint32_t)) != hipSuccess) {
    demi_ctx_destroy(ctx);
    return DEMI_ERR_DEVICE;
  }
  *out = ctx;
  return DEMI_OK;
}

extern "C" void demi_ctx_destroy(demi_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->d_model) (void)hipFree(ctx->d_model);
  if (ctx->d_trace) (void)hipFree(ctx->d_trace);
  if (ctx->d_counter) (void)hipFree(ctx->d_counter);
  if (ctx->d_out) (void)hipFree(ctx->d_out);
  if (ctx->d_seeds) (void)hipFree(ctx->d_seeds);
  if (ctx->d_rec) (void)hipFree(ctx->d_rec);
  if (ctx->d_rec_count) (void)hipFree(ctx->d_rec_count);
  if (ctx->d_spill) (void)hipFree(ctx->d_spill);
  if (ctx->d_rext) (void)hipFree(ctx->d_rext);
  if (ctx->d_expected) (void)hipFree(ctx->d_expected);
  if (ctx->d_masks) (void)hipFree(ctx->d_masks);
  if (ctx->d_skip) (void)hipFree(ctx->d_skip);
  if (ctx->d_kept) (void)hipFree(ctx->d_kept);
  if (ctx->d_dext) (void)hipFree(ctx->d_dext);
  if (ctx->d_dpor) (void)hipFree(ctx->d_dpor);
  if (ctx->d_viol) (void)hipFree(ctx->d_viol);
  if (ctx->d_viol_count) (void)hipFree(ctx->d_viol_count);
  delete ctx;
}

// ----------------------------------------------------------------------------- model
// Validation of what the adapter lowered (rules are part of the boundary's spec).
static int validate_model(demi_ctx* ctx, const demi_model* m) {
  if (!m) return fail(ctx, DEMI_ERR_INVALID_MODEL, "null model");
  if (m->n_actors < 1 || m->n_actors > DEMI_MAX_ACTORS) return fail(ctx, DEMI_ERR_INVALID_MODEL, "n_actors out of range");
  if (m->n_msg_types < 1 || m->n_msg_types > DEMI_MAX_MSG_TYPES) return fail(ctx, DEMI_ERR_INVALID_MODEL, "n_msg_types out of range");
  if (m->n_classes < 1 || m->n_classes > DEMI_MAX_CLASSES) return fail(ctx, DEMI_ERR_INVALID_MODEL, "n_classes out of range");
  if (m->code_len < 1 || m->code_len > DEMI_MAX_CODE) return fail(ctx, DEMI_ERR_INVALID_MODEL, "code_len out of range");
  if (!m->msg_class || !m->actor_class || !m->handler_start || !m->code || !m->init_state)
    return fail(ctx, DEMI_ERR_INVALID_MODEL, "null table pointer");
  uint32_t timers = 0;
  for (uint32_t t = 0; t < m->n_msg_types; t++) {
    if (m->msg_class[t] > DEMI_MSG_TIMER) return fail(ctx, DEMI_ERR_INVALID_MODEL, "msg_class[%u] invalid", t);
    timers += m->msg_class[t] == DEMI_MSG_TIMER;
  }
  if (timers > DEMI_MAX_TIMER_TYPES) return fail(ctx, DEMI_ERR_INVALID_MODEL, "more than %d timer types", DEMI_MAX_TIMER_TYPES);
  for (uint32_t a = 0; a < m->n_actors; a++)
    if (m->actor_class[a] >= m->n_classes) return fail(ctx, DEMI_ERR_INVALID_MODEL, "actor_class[%u] out of range", a);
  for (uint32_t i = 0; i < m->n_classes * m->n_msg_types; i++)
    if (m->handler_start[i] != 0xFFFF && m->handler_start[i] >= m->code_len)
      return fail(ctx, DEMI_ERR_INVALID_MODEL, "handler_start[%u] out of range", i);
  for (uint32_t pc = 0; pc < m->code_len; pc++) {
    const uint32_t w = m->code[pc], op = w & 0xFF, bimm = (w >> 16) & 1, aux = (w >> 17) & 0x7F, b = w >> 24;
    if (!bimm && b > 15) return fail(ctx, DEMI_ERR_INVALID_MODEL, "row %u: register operand b out of range", pc);
    if (op <= DEMI_OP_MAX && op != 19) continue;
    if (op >= DEMI_OP_SKIPZ && op <= DEMI_OP_SKIP) {
      if (!bimm) return fail(ctx, DEMI_ERR_INVALID_MODEL, "row %u: skip distance must be an immediate", pc);
      if (pc + 1 + b > m->code_len) return fail(ctx, DEMI_ERR_INVALID_MODEL, "row %u: skip past the end of the table", pc);
    } else if (op >= DEMI_OP_IFEQ && op <= DEMI_OP_IFGT) {
      if (pc + 1 + aux > m->code_len) return fail(ctx, DEMI_ERR_INVALID_MODEL, "row %u: skip past the end of the table", pc);
    } else if (op == DEMI_OP_SEND || op == DEMI_OP_BCAST) {
      if (aux >= m->n_msg_types || m->msg_class[aux] != DEMI_MSG_INTERNAL)
        return fail(ctx, DEMI_ERR_INVALID_MODEL, "row %u: SEND/BCAST of a non-internal message type %u", pc, aux);
    } else if (op >= DEMI_OP_TSET && op <= DEMI_OP_TCANCEL) {
      if (aux >= m->n_msg_types || m->msg_class[aux] != DEMI_MSG_TIMER)
        return fail(ctx, DEMI_ERR_INVALID_MODEL, "row %u: timer op on a non-timer message type %u", pc, aux);
    } else {
      return fail(ctx, DEMI_ERR_INVALID_MODEL, "row %u: unknown op %u", pc, op);
    }
  }
  if (m->inv_kind > DEMI_INV_AGREE) return fail(ctx, DEMI_ERR_INVALID_MODEL, "inv_kind invalid");
  if (m->inv_fa > 7 || m->inv_fb > 7 || m->inv_va > 255) return fail(ctx, DEMI_ERR_INVALID_MODEL, "invariant field out of range");
  return DEMI_OK;
}

static int validate_trace(demi_ctx* ctx, const DevModel& hm, const demi_ext_event* ev, uint32_t n) {
  if (n > DEMI_MAX_EXT_EVENTS) return fail(ctx, DEMI_ERR_INVALID_TRACE, "more than %d external events", DEMI_MAX_EXT_EVENTS);
  for (uint32_t i = 0; i < n; i++) {
    const demi_ext_event& e = ev[i];
    switch (e.kind) {
      case DEMI_EV_START: case DEMI_EV_KILL:
        if (e.a >= hm.n_actors) return fail(ctx, DEMI_ERR_INVALID_TRACE, "event %u: actor out of range", i);
        break;
      case DEMI_EV_SEND:
        if (e.a >= hm.n_actors) return fail(ctx, DEMI_ERR_INVALID_TRACE, "event %u: receiver out of range", i);
        if (e.msg_type >= hm.n_msg_types || (hm.meta[e.msg_type] & 0xFF) != DEMI_MSG_EXTERNAL)
          return fail(ctx, DEMI_ERR_INVALID_TRACE, "event %u: Send of a non-external message type", i);
        break;
      case DEMI_EV_PARTITION: case DEMI_EV_UNPARTITION:
        if (e.a >= hm.n_actors || e.b >= hm.n_actors) return fail(ctx, DEMI_ERR_INVALID_TRACE, "event %u: actor out of range", i);
        break;
      case DEMI_EV_WAIT_QUIESCENCE:
        break;
      default:
        return fail(ctx, DEMI_ERR_INVALID_TRACE,
                    "event %u: unsupported external event kind %u (WaitCondition/CodeBlock/HardKill need the JVM scheduler)",
                    i, e.kind);
    }
  }
  return DEMI_OK;
}

extern "C" int demi_model_load(demi_ctx* ctx, const demi_model* m) {
  if (!ctx) return DEMI_ERR_INVALID_ARG;
  int rc = validate_model(ctx, m);
  if (rc) return rc;
  DevModel& h = ctx->hmodel;
  memset(&h, 0, sizeof h);
  h.n_actors = m->n_actors; h.n_msg_types = m->n_msg_types; h.n_classes = m->n_classes; h.code_len = m->code_len;
  h.inv_kind = m->inv_kind; h.inv_fa = m->inv_fa; h.inv_va = m->inv_va; h.inv_fb = m->inv_fb;
  h.fp_match_mask = m->fp_match_mask;
  uint32_t tix = 0;
  for (uint32_t t = 0; t < m->n_msg_types; t++) {
    h.meta[t] = m->msg_class[t] | (tix << 8);
    if (m->msg_class[t] == DEMI_MSG_TIMER) tix++;
  }
  for (uint32_t i = 0; i < DEMI_MAX_CLASSES * DEMI_MAX_MSG_TYPES; i++) h.handler_start[i] = 0xFFFF;
  for (uint32_t i = 0; i < m->n_classes * m->n_msg_types; i++) h.handler_start[i] = m->handler_start[i];
  for (uint32_t a = 0; a < m->n_actors; a++) { h.actor_class[a] = m->actor_class[a]; h.init_state[a] = m->init_state[a]; }
  for (uint32_t d = 1; d <= 128; d++) {
    uint32_t L = 0;
    while ((1u << L) < d) L++;
    const unsigned __int128 num = (unsigned __int128)1 << (31 + L);
    h.divmagic[d] = (uint32_t)((num + d - 1) / d);  // only read for non-powers of two (fits 32 bits)
  }
  for (uint32_t op = 0; op < 64; op++) h.optab[op] = op_control(op);
  memcpy(h.code, m->code, sizeof(uint32_t) * m->code_len);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpy(ctx->d_model, &h, sizeof h, hipMemcpyHostToDevice));
  ctx->have_model = true;
  ctx->have_trace = false;  // a trace is validated against the model it was loaded after
  ctx->have_replay = false;
  ctx->have_dpor = false;
  return DEMI_OK;
}

extern "C" int demi_trace_load(demi_ctx* ctx, const demi_ext_event* events, uint32_t n_events) {
  if (!ctx) return DEMI_ERR_INVALID_ARG;
  if (!ctx->have_model) return fail(ctx, DEMI_ERR_NO_MODEL, "demi_model_load must precede demi_trace_load");
  if (!events && n_events) return fail(ctx, DEMI_ERR_INVALID_ARG, "null events");
  int rc = validate_trace(ctx, ctx->hmodel, events, n_events);
  if (rc) return rc;
  ctx->trace.assign(events, events + n_events);
  ctx->started_mask = 0;
  ctx->n_batches = 1;
  for (uint32_t i = 0; i < n_events; i++) {
    if (events[i].kind == DEMI_EV_START) ctx->started_mask |= 1u << events[i].a;
    if (events[i].kind == DEMI_EV_WAIT_QUIESCENCE) ctx->n_batches++;
  }
  static_assert(sizeof(demi_ext_event) == 8, "event is one 8-byte word");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (n_events)
    HIP_TRY(ctx, hipMemcpy(ctx->d_trace, events, sizeof(demi_ext_event) * n_events, hipMemcpyHostToDevice));
  ctx->have_trace = true;
  return DEMI_OK;
}

// ----------------------------------------------------------------------------- K1 launch
template <bool REC>
static int launch_k1(demi_ctx* ctx, uint32_t p_max, K1Args a, hipStream_t stream) {
  const DevModel& h = ctx->hmodel;
  if (p_max == 0) p_max = 64;
  if (p_max > DEMI_MAX_PENDING) return fail(ctx, DEMI_ERR_INVALID_ARG, "p_max must be 1..%d", DEMI_MAX_PENDING);
  a.p_max = p_max;
  a.n_batches = ctx->n_batches;
  const size_t lds = k1_lds_bytes<REC>(h.code_len, a.n_ev, h.n_classes * h.n_msg_types, h.n_actors, a.n_batches);
  if (lds > 160 * 1024) return fail(ctx, DEMI_ERR_INVALID_ARG, "LDS budget exceeded (%zu bytes)", lds);
  auto kern = k1_random_explore<REC>;
  HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = 0;
  HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, K1_WAVES * 64, lds));
  if (per_cu < 1) per_cu = 1;
  // persistent-style grid: every wave keeps claiming 64-schedule batches until the range is done
  uint64_t blocks = (a.n + (uint64_t)K1_WAVES * 64 - 1) / ((uint64_t)K1_WAVES * 64);
  const uint64_t resident = (uint64_t)ctx->num_cu * (uint64_t)per_cu;
  if (blocks > resident) blocks = resident;
  if (blocks < 1) blocks = 1;
  // HBM scratch for the (rare) pending slots beyond the PEND_HOT kept in LDS
  const size_t need = spill_words(blocks * K1_WAVES * 64) * (REC ? 2 : 1) * sizeof(uint32_t);
  if (ctx->spill_bytes < need) {
    if (ctx->d_spill) (void)hipFree(ctx->d_spill);
    ctx->d_spill = nullptr; ctx->spill_bytes = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_spill, need));
    ctx->spill_bytes = need;
  }
  a.spill = ctx->d_spill;
  HIP_TRY(ctx, hipMemsetAsync(a.work_counter, 0, sizeof(unsigned long long), stream));
#ifdef DEMI_K1_PHASES
  // diagnostic build only (tools/k1_phases.sh): per-phase cycle totals of every wave
  static unsigned long long* d_ph = nullptr;
  const size_t n_ph = (size_t)blocks * K1_WAVES * 8;
  if (!d_ph) HIP_TRY(ctx, hipMalloc(&d_ph, sizeof(unsigned long long) * 8 * 4096 * K1_WAVES));
  HIP_TRY(ctx, hipMemsetAsync(d_ph, 0, sizeof(unsigned long long) * n_ph, stream));
  a.phase_out = d_ph;
#endif
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(K1_WAVES * 64), lds, stream, a);
  HIP_TRY(ctx, hipGetLastError());
#ifdef DEMI_K1_PHASES
  {
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    std::vector<unsigned long long> h(n_ph);
    HIP_TRY(ctx, hipMemcpy(h.data(), d_ph, sizeof(unsigned long long) * n_ph, hipMemcpyDeviceToHost));
    unsigned long long tot[8] = {0};
    for (size_t i = 0; i < n_ph; i++) tot[i & 7] += h[i];
    double all = 0;
    for (int i = 0; i < 6; i++) all += (double)tot[i];
    const char* names[6] = {"refill+finish", "inject", "sched", "rows", "apply", "tail"};
    fprintf(stderr, "[k1 phases] waves=%zu", n_ph / 8);
    for (int i = 0; i < 6; i++) fprintf(stderr, " %s=%.1f%%", names[i], 100.0 * (double)tot[i] / all);
    fprintf(stderr, " iters/wave=%.0f active-lanes/iter=%.1f cycles/iter=%.0f\n", (double)tot[6] / (n_ph / 8),
            (double)tot[7] / (double)tot[6], all / (double)tot[6]);
  }
#endif
  return DEMI_OK;
}

static int make_k1_args(demi_ctx* ctx, uint64_t seed_base, const uint64_t* d_seeds, uint64_t n,
                        const demi_limits* lim, demi_verdict* d_out, K1Args* a) {
  if (!ctx->have_model) return fail(ctx, DEMI_ERR_NO_MODEL, "no model loaded (setInvariant / model_load must precede explore)");
  if (!ctx->have_trace) return fail(ctx, DEMI_ERR_NO_TRACE, "no trace loaded");
  if (!lim) return fail(ctx, DEMI_ERR_INVALID_ARG, "null limits");
  memset(a, 0, sizeof *a);
  a->model = ctx->d_model;
  a->trace = ctx->d_trace;
  a->n_ev = (uint32_t)ctx->trace.size();
  a->exists = lim->populate_all ? ((1u << ctx->hmodel.n_actors) - 1) : ctx->started_mask;
  a->seed_base = seed_base;
  a->seeds = d_seeds;
  a->n = n;
  a->max_messages = lim->max_messages;
  a->interval = lim->invariant_check_interval;
  a->looking_for_valid = lim->looking_for_valid;
  a->looking_for = lim->looking_for;
  a->out = d_out;
  a->work_counter = ctx->d_counter;
  return DEMI_OK;
}

extern "C" int demi_random_explore_dev(demi_ctx* ctx, uint64_t seed_base, const uint64_t* d_seeds, uint64_t n,
                                       const demi_limits* limits, demi_verdict* d_out, void* hip_stream) {
  if (!ctx) return DEMI_ERR_INVALID_ARG;
  if (n == 0) return DEMI_OK;
  if (!d_out) return fail(ctx, DEMI_ERR_INVALID_ARG, "null output");
  K1Args a;
  int rc = make_k1_args(ctx, seed_base, d_seeds, n, limits, d_out, &a);
  if (rc) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  return launch_k1<false>(ctx, limits->p_max, a, static_cast<hipStream_t>(hip_stream));
}

extern "C" int demi_random_explore(demi_ctx* ctx, uint64_t seed_base, const uint64_t* seeds, uint64_t n,
                                   const demi_limits* limits, demi_verdict* out) {
  if (!ctx) return DEMI_ERR_INVALID_ARG;
  if (n == 0) return DEMI_OK;
  if (!out) return fail(ctx, DEMI_ERR_INVALID_ARG, "null output");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->out_cap < n) {
    if (ctx->d_out) (void)hipFree(ctx->d_out);
    ctx->d_out = nullptr; ctx->out_cap = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_out, sizeof(demi_verdict) * n));
    ctx->out_cap = n;
  }
  const uint64_t* d_seeds = nullptr;
  if (seeds) {
    if (ctx->seeds_cap < n) {
      if (ctx->d_seeds) (void)hipFree(ctx->d_seeds);
      ctx->d_seeds = nullptr; ctx->seeds_cap = 0;
      HIP_TRY(ctx, hipMalloc(&ctx->d_seeds, sizeof(uint64_t) * n));
      ctx->seeds_cap = n;
    }
    HIP_TRY(ctx, hipMemcpy(ctx->d_seeds, seeds, sizeof(uint64_t) * n, hipMemcpyHostToDevice));
    d_seeds = ctx->d_seeds;
  }
  int rc = demi_random_explore_dev(ctx, seed_base, d_seeds, n, limits, ctx->d_out, nullptr);
  if (rc) return rc;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(out, ctx->d_out, sizeof(demi_verdict) * n, hipMemcpyDeviceToHost));
  return DEMI_OK;
}

extern "C" int demi_random_get_trace(demi_ctx* ctx, uint64_t seed, const demi_limits* limits, demi_verdict* verdict,
                                     demi_rec_event* out, uint32_t cap, uint32_t* n_out) {
  if (!ctx) return DEMI_ERR_INVALID_ARG;
  if (!verdict || !n_out || (!out && cap)) return fail(ctx, DEMI_ERR_INVALID_ARG, "null output");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->out_cap < 1) {
    HIP_TRY(ctx, hipMalloc(&ctx->d_out, sizeof(demi_verdict)));
    ctx->out_cap = 1;
  }
  K1Args a;
  int rc = make_k1_args(ctx, seed, nullptr, 1, limits, ctx->d_out, &a);
  if (rc) return rc;
  a.rec_out = ctx->d_rec;
  a.rec_count = ctx->d_rec_count;
  a.rec_cap = DEMI_MAX_REC_EVENTS;
  rc = launch_k1<true>(ctx, limits->p_max, a, nullptr);
  if (rc) return rc;
  HIP_TRY(ctx, hipDeviceSynchronize());
  uint32_t n_rec = 0;
  HIP_TRY(ctx, hipMemcpy(verdict, ctx->d_out, sizeof(demi_verdict), hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(&n_rec, ctx->d_rec_count, sizeof n_rec, hipMemcpyDeviceToHost));
  *n_out = n_rec;
  if (n_rec > DEMI_MAX_REC_EVENTS) return fail(ctx, DEMI_ERR_CAPACITY, "execution recorded %u events (> %d)", n_rec, DEMI_MAX_REC_EVENTS);
  if (n_rec > cap) return fail(ctx, DEMI_ERR_CAPACITY, "caller buffer holds %u events, %u recorded", cap, n_rec);
  if (n_rec) HIP_TRY(ctx, hipMemcpy(out, ctx->d_rec, sizeof(demi_rec_event) * n_rec, hipMemcpyDeviceToHost));
  return DEMI_OK;
}

// ----------------------------------------------------------------------------- violation set
extern "C" int demi_collect_violations_dev(demi_ctx* ctx, const demi_verdict* d_verdicts, uint64_t n,
                                           uint64_t index_base, demi_violation* d_out, uint32_t cap,
                                           unsigned long long* d_count, void* hip_stream) {
  if (!ctx) return DEMI_ERR_INVALID_ARG;
  if (!d_count || (!d_verdicts && n) || (!d_out && cap)) return fail(ctx, DEMI_ERR_INVALID_ARG, "null pointer");
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemsetAsync(d_count, 0, sizeof(unsigned long long), stream));
  if (n == 0) return DEMI_OK;
  uint64_t blocks = (n + 255) / 256;
  const uint64_t maxb = (uint64_t)ctx->num_cu * 8;
  if (blocks > maxb) blocks = maxb;
  hipLaunchKernelGGL(k_collect_violations, dim3((unsigned)blocks), dim3(256), 0, stream, d_verdicts, n, index_base, d_out,
                     cap, d_count);
  HIP_TRY(ctx, hipGetLastError());
  return DEMI_OK;
}

// ----------------------------------------------------------------------------- K2: replay
static int ensure_spill(demi_ctx* ctx, size_t lanes, int arrays) {
  const size_t need = spill_words(lanes) * (size_t)arrays * sizeof(uint32_t);
  if (ctx->spill_bytes < need) {
    if (ctx->d_spill) (void)hipFree(ctx->d_spill);
    ctx->d_spill = nullptr; ctx->spill_bytes = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_spill, need));
    ctx->spill_bytes = need;
  }
  return DEMI_OK;
}

extern "C" int demi_replay_load(demi_ctx* ctx, const demi_ext_event* ext, uint32_t n_ext, const demi_rec_event* rec,
                                uint32_t n_rec) {
  if (!ctx) return DEMI_ERR_INVALID_ARG;
  if (!ctx->have_model) return fail(ctx, DEMI_ERR_NO_MODEL, "demi_model_load must precede demi_replay_load");
  if ((!ext && n_ext) || (!rec && n_rec)) return fail(ctx, DEMI_ERR_INVALID_ARG, "null pointer");
  if (n_rec > DEMI_MAX_REC_EVENTS) return fail(ctx, DEMI_ERR_INVALID_TRACE, "more than %d recorded events", DEMI_MAX_REC_EVENTS);
  int rc = validate_trace(ctx, ctx->hmodel, ext, n_ext);
  if (rc) return rc;
  const DevModel& h = ctx->hmodel;
  // id -> index of the Send that enqueued it: filterSends pairs the k-th external MsgSend with the
  // k-th Send; our recorder stores that index explicitly (ext_idx)
  std::vector<uint8_t> send_of_id(2 * DEMI_MAX_REC_EVENTS + 2, 255);
  std::vector<uint64_t> expd;
  std::vector<int32_t> exp_of_rec(n_rec, -1);
  std::vector<uint8_t> rec_is_delivery(n_rec, 0);
  uint32_t spawned = 0;
  for (uint32_t i = 0; i < n_rec; i++) {
    const demi_rec_event& e = rec[i];
    const size_t before = expd.size();
    auto pack = [](uint32_t kind, uint32_t a, uint32_t b, uint32_t type, uint32_t p0, uint32_t p1, uint32_t x) {
      return (uint64_t)kind | ((uint64_t)a << 8) | ((uint64_t)b << 16) | ((uint64_t)type << 24) | ((uint64_t)p0 << 32) |
             ((uint64_t)p1 << 40) | ((uint64_t)x << 48);
    };
    switch (e.kind) {
      case DEMI_REC_SPAWN: case DEMI_REC_KILL:
        if (e.rcv >= h.n_actors) return fail(ctx, DEMI_ERR_INVALID_TRACE, "recorded event %u: actor out of range", i);
        if (e.kind == DEMI_REC_SPAWN) spawned |= 1u << e.rcv;
        expd.push_back(pack(e.kind, e.rcv, 0, 0, 0, 0, e.ext_idx));
        break;
      case DEMI_REC_PARTITION: case DEMI_REC_UNPARTITION:
        if (e.snd >= h.n_actors || e.rcv >= h.n_actors) return fail(ctx, DEMI_ERR_INVALID_TRACE, "recorded event %u: actor out of range", i);
        expd.push_back(pack(e.kind, e.snd, e.rcv, 0, 0, 0, e.ext_idx));
        break;
      case DEMI_REC_MSG_SEND:
        if (e.rcv >= h.n_actors || e.msg_type >= h.n_msg_types) return fail(ctx, DEMI_ERR_INVALID_TRACE, "recorded event %u: bad message", i);
        if (e.flags & 1) {  // external
          if (e.ext_idx >= n_ext || ext[e.ext_idx].kind != DEMI_EV_SEND)
            return fail(ctx, DEMI_ERR_INVALID_TRACE, "recorded event %u: external MsgSend without its Send", i);
          if (e.id < send_of_id.size()) send_of_id[e.id] = e.ext_idx;
          expd.push_back(pack(e.kind, 0, e.rcv, e.msg_type, e.p0, e.p1, e.ext_idx));
        }
        break;
      case DEMI_REC_MSG_EVENT: {
        if (e.rcv >= h.n_actors || e.msg_type >= h.n_msg_types || (e.snd >= h.n_actors && e.snd != DEMI_DEADLETTERS))
          return fail(ctx, DEMI_ERR_INVALID_TRACE, "recorded event %u: bad message", i);
        const uint8_t s = e.id < send_of_id.size() ? send_of_id[e.id] : 255;
        expd.push_back(pack(e.kind, e.snd, e.rcv, e.msg_type, e.p0, e.p1, s));
        break;
      }
      case DEMI_REC_BEGIN_WAIT_QUIESCENCE: case DEMI_REC_QUIESCENCE:
        break;  // nops in advanceReplay (:530-538)
      default:
        return fail(ctx, DEMI_ERR_INVALID_TRACE, "recorded event %u: unknown kind %u", i, e.kind);
    }
    if (expd.size() != before) exp_of_rec[i] = (int32_t)before;
    rec_is_delivery[i] = e.kind == DEMI_REC_MSG_EVENT;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!ctx->d_rext) HIP_TRY(ctx, hipMalloc(&ctx->d_rext, sizeof(uint64_t) * (DEMI_MAX_EXT_EVENTS + 1)));
  if (!ctx->d_expected) HIP_TRY(ctx, hipMalloc(&ctx->d_expected, sizeof(uint64_t) * (DEMI_MAX_REC_EVENTS + 1)));
  if (n_ext) HIP_TRY(ctx, hipMemcpy(ctx->d_rext, ext, sizeof(demi_ext_event) * n_ext, hipMemcpyHostToDevice));
  if (!expd.empty()) HIP_TRY(ctx, hipMemcpy(ctx->d_expected, expd.data(), sizeof(uint64_t) * expd.size(), hipMemcpyHostToDevice));
  ctx->n_rext = n_ext;
  ctx->n_expected = (uint32_t)expd.size();
  ctx->replay_spawned = spawned;
  ctx->exp_of_rec.swap(exp_of_rec);
  ctx->rec_is_delivery.swap(rec_is_delivery);
  ctx->have_replay = true;
  return DEMI_OK;
}

// d_masks == nullptr: every external kept; d_skip / d_kept: see K2Args
static int replay_launch(demi_ctx* ctx, const uint64_t* d_masks, const uint32_t* d_skip, uint8_t* d_kept, uint64_t n,
                         const demi_limits* lim, demi_verdict* d_out, void* hip_stream) {
  if (!ctx->have_model) return fail(ctx, DEMI_ERR_NO_MODEL, "no model loaded");
  if (!ctx->have_replay) return fail(ctx, DEMI_ERR_NO_TRACE, "demi_replay_load must precede demi_replay_batch");
  if (!lim || !d_out) return fail(ctx, DEMI_ERR_INVALID_ARG, "null pointer");
  if (!lim->looking_for_valid) return fail(ctx, DEMI_ERR_INVALID_ARG, "replay needs the target fingerprint (looking_for)");
  uint32_t p_max = lim->p_max ? lim->p_max : 64;
  if (p_max > DEMI_MAX_PENDING) return fail(ctx, DEMI_ERR_INVALID_ARG, "p_max must be 1..%d", DEMI_MAX_PENDING);
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  const DevModel& h = ctx->hmodel;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t lds = k2_lds_bytes(h.code_len, ctx->n_rext, h.n_classes * h.n_msg_types, h.n_actors);
  HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k2_replay), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = 0;
  HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k2_replay, K2_WAVES * 64, lds));
  if (per_cu < 1) per_cu = 1;
  // candidates are few (a DDMin frontier): spread them over as many CUs as possible, one wave each
  uint64_t blocks = (n + 63) / 64;
  const uint64_t resident = (uint64_t)ctx->num_cu * (uint64_t)per_cu;
  if (blocks > resident) blocks = resident;
  int rc = ensure_spill(ctx, blocks * K2_WAVES * 64, 1);
  if (rc) return rc;
  K2Args a;
  memset(&a, 0, sizeof a);
  a.model = ctx->d_model; a.ext = ctx->d_rext; a.n_ext = ctx->n_rext;
  a.exists = lim->populate_all ? ((1u << h.n_actors) - 1) : ctx->replay_spawned;
  a.expected = ctx->d_expected; a.n_exp = ctx->n_expected;
  a.p_max = p_max; a.looking_for = lim->looking_for;
  a.masks = d_masks; a.skip = d_skip; a.kept = d_kept;
  a.n = n; a.out = d_out; a.work_counter = ctx->d_counter; a.spill = ctx->d_spill;
  HIP_TRY(ctx, hipMemsetAsync(a.work_counter, 0, sizeof(unsigned long long), stream));
  hipLaunchKernelGGL(k2_replay, dim3((unsigned)blocks), dim3(K2_WAVES * 64), lds, stream, a);
  HIP_TRY(ctx, hipGetLastError());
  return DEMI_OK;
}

extern "C" int demi_replay_batch_dev(demi_ctx* ctx, const uint64_t* d_masks, uint64_t n, const demi_limits* lim,
                                     demi_verdict* d_out, void* hip_stream) {
  if (!ctx) return DEMI_ERR_INVALID_ARG;
  if (n == 0) return DEMI_OK;
  if (!d_masks) return fail(ctx, DEMI_ERR_INVALID_ARG, "null pointer");
  return replay_launch(ctx, d_masks, nullptr, nullptr, n, lim, d_out, hip_stream);
}

static int ensure_replay_buffers(demi_ctx* ctx, uint64_t n, bool masks) {
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->out_cap < n) {
    if (ctx->d_out) (void)hipFree(ctx->d_out);
    ctx->d_out = nullptr; ctx->out_cap = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_out, sizeof(demi_verdict) * n));
    ctx->out_cap = n;
  }
  if (masks && ctx->masks_cap < n) {
    if (ctx->d_masks) (void)hipFree(ctx->d_masks);
    ctx->d_masks = nullptr; ctx->masks_cap = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_masks, sizeof(uint64_t) * 4 * n));
    ctx->masks_cap = n;
  }
  return DEMI_OK;
}

// recorded-event index of a removal candidate -> index in the lowered trace (must be a MsgEvent)
static int skip_to_expected(demi_ctx* ctx, uint32_t skip, uint32_t* out) {
  if (skip == 0xFFFFFFFFu) { *out = skip; return DEMI_OK; }
  if (skip >= ctx->exp_of_rec.size() || ctx->exp_of_rec[skip] < 0 || !ctx->rec_is_delivery[skip])
    return fail(ctx, DEMI_ERR_INVALID_ARG, "removal candidate %u is not a delivery of the loaded trace", skip);
  *out = (uint32_t)ctx->exp_of_rec[skip];
  return DEMI_OK;
}

extern "C" int demi_replay_removal_batch(demi_ctx* ctx, const uint64_t* masks, const uint32_t* skip, uint64_t n,
                                         const demi_limits* lim, demi_verdict* out) {
  if (!ctx) return DEMI_ERR_INVALID_ARG;
  if (n == 0) return DEMI_OK;
  if (!skip || !out) return fail(ctx, DEMI_ERR_INVALID_ARG, "null pointer");
  if (!ctx->have_replay) return fail(ctx, DEMI_ERR_NO_TRACE, "demi_replay_load must precede demi_replay_removal_batch");
  int rc = ensure_replay_buffers(ctx, n, masks != nullptr);
  if (rc) return rc;
  if (ctx->skip_cap < n) {
    if (ctx->d_skip) (void)hipFree(ctx->d_skip);
    ctx->d_skip = nullptr; ctx->skip_cap = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_skip, sizeof(uint32_t) * n));
    ctx->skip_cap = n;
  }
  std::vector<uint32_t> sk(n);
  for (uint64_t i = 0; i < n; i++) {
    rc = skip_to_expected(ctx, skip[i], &sk[i]);
    if (rc) return rc;
  }
  HIP_TRY(ctx, hipMemcpy(ctx->d_skip, sk.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
  if (masks) HIP_TRY(ctx, hipMemcpy(ctx->d_masks, masks, sizeof(uint64_t) * 4 * n, hipMemcpyHostToDevice));
  rc = replay_launch(ctx, masks ? ctx->d_masks : nullptr, ctx->d_skip, nullptr, n, lim, ctx->d_out, nullptr);
  if (rc) return rc;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(out, ctx->d_out, sizeof(demi_verdict) * n, hipMemcpyDeviceToHost));
  return DEMI_OK;
}

extern "C" int demi_replay_get_kept(demi_ctx* ctx, const uint64_t* mask, uint32_t skip, const demi_limits* lim,
                                    demi_verdict* verdict, uint8_t* kept) {
  if (!ctx) return DEMI_ERR_INVALID_ARG;
  if (!verdict || !kept) return fail(ctx, DEMI_ERR_INVALID_ARG, "null pointer");
  if (!ctx->have_replay) return fail(ctx, DEMI_ERR_NO_TRACE, "demi_replay_load must precede demi_replay_get_kept");
  int rc = ensure_replay_buffers(ctx, 1, true);
  if (rc) return rc;
  if (ctx->skip_cap < 1) {
    HIP_TRY(ctx, hipMalloc(&ctx->d_skip, sizeof(uint32_t)));
    ctx->skip_cap = 1;
  }
  if (!ctx->d_kept) HIP_TRY(ctx, hipMalloc(&ctx->d_kept, DEMI_MAX_REC_EVENTS + 1));
  uint32_t sk = 0;
  rc = skip_to_expected(ctx, skip, &sk);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpy(ctx->d_skip, &sk, sizeof sk, hipMemcpyHostToDevice));
  if (mask) HIP_TRY(ctx, hipMemcpy(ctx->d_masks, mask, sizeof(uint64_t) * 4, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemset(ctx->d_kept, 0, DEMI_MAX_REC_EVENTS + 1));
  rc = replay_launch(ctx, mask ? ctx->d_masks : nullptr, ctx->d_skip, ctx->d_kept, 1, lim, ctx->d_out, nullptr);
  if (rc) return rc;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(verdict, ctx->d_out, sizeof(demi_verdict), hipMemcpyDeviceToHost));
  std::vector<uint8_t> ke(ctx->n_expected + 1);
  if (ctx->n_expected) HIP_TRY(ctx, hipMemcpy(ke.data(), ctx->d_kept, ctx->n_expected, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < ctx->exp_of_rec.size(); i++)
    kept[i] = ctx->exp_of_rec[i] >= 0 ? ke[ctx->exp_of_rec[i]] : 0;
  return DEMI_OK;
}

extern "C" int demi_replay_batch(demi_ctx* ctx, const uint64_t* masks, uint64_t n, const demi_limits* lim,
                                 demi_verdict* out) {
  if (!ctx) return DEMI_ERR_INVALID_ARG;
  if (n == 0) return DEMI_OK;
  if (!masks || !out) return fail(ctx, DEMI_ERR_INVALID_ARG, "null pointer");
  int rc = ensure_replay_buffers(ctx, n, true);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpy(ctx->d_masks, masks, sizeof(uint64_t) * 4 * n, hipMemcpyHostToDevice));
  rc = demi_replay_batch_dev(ctx, ctx->d_masks, n, lim, ctx->d_out, nullptr);
  if (rc) return rc;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(out, ctx->d_out, sizeof(demi_verdict) * n, hipMemcpyDeviceToHost));
  return DEMI_OK;
}

// ----------------------------------------------------------------------------- K3: DPOR
extern "C" int demi_dpor_load(demi_ctx* ctx, const demi_ext_event* ext, uint32_t n_ext) {
  if (!ctx) return DEMI_ERR_INVALID_ARG;
  if (!ctx->have_model) return fail(ctx, DEMI_ERR_NO_MODEL, "demi_model_load must precede demi_dpor_load");
  if (!ext && n_ext) return fail(ctx, DEMI_ERR_INVALID_ARG, "null pointer");
  int rc = validate_trace(ctx, ctx->hmodel, ext, n_ext);
  if (rc) return rc;
  for (uint32_t i = 0; i < n_ext; i++)
    if (ext[i].kind != DEMI_EV_START && ext[i].kind != DEMI_EV_SEND && ext[i].kind != DEMI_EV_WAIT_QUIESCENCE)
      return fail(ctx, DEMI_ERR_INVALID_TRACE, "event %u: unsuported external event for DPOR", i);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!ctx->d_dext) HIP_TRY(ctx, hipMalloc(&ctx->d_dext, sizeof(uint64_t) * (DEMI_MAX_EXT_EVENTS + 1)));
  if (n_ext) HIP_TRY(ctx, hipMemcpy(ctx->d_dext, ext, sizeof(demi_ext_event) * n_ext, hipMemcpyHostToDevice));
  ctx->n_dext = n_ext;
  ctx->have_dpor = true;
  return DEMI_OK;
}

extern "C" int demi_dpor_batch(demi_ctx* ctx, const demi_dpor_trace_entry* prefixes, const uint32_t* prefix_len,
                               uint32_t stride, uint64_t n, const demi_dpor_params* par, demi_verdict* out_verdicts,
                               demi_dpor_trace_entry* out_traces, uint32_t* out_trace_len, demi_dpor_pair* out_pairs,
                               uint32_t* out_n_pairs) {
  if (!ctx) return DEMI_ERR_INVALID_ARG;
  if (n == 0) return DEMI_OK;
  if (!ctx->have_model) return fail(ctx, DEMI_ERR_NO_MODEL, "no model loaded");
  if (!ctx->have_dpor) return fail(ctx, DEMI_ERR_NO_TRACE, "demi_dpor_load must precede demi_dpor_batch");
  if (!par || !prefix_len || !out_verdicts || !out_traces || !out_trace_len || !out_n_pairs || (!prefixes && stride) ||
      (!out_pairs && par->max_pairs))
    return fail(ctx, DEMI_ERR_INVALID_ARG, "null pointer");
  uint32_t p_max = par->p_max ? par->p_max : 64;
  if (p_max > DEMI_MAX_PENDING) return fail(ctx, DEMI_ERR_INVALID_ARG, "p_max must be 1..%d", DEMI_MAX_PENDING);
  for (uint64_t i = 0; i < n; i++)
    if (prefix_len[i] > stride) return fail(ctx, DEMI_ERR_INVALID_ARG, "prefix %llu longer than stride", (unsigned long long)i);
  const DevModel& h = ctx->hmodel;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // arena: prefixes | prefix_len | verdicts | traces | trace_len | pairs | n_pairs
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_pfx = 0, s_pfx = al(sizeof(demi_dpor_trace_entry) * (size_t)stride * n);
  const size_t o_pl = o_pfx + s_pfx, s_pl = al(4 * n);
  const size_t o_v = o_pl + s_pl, s_v = al(sizeof(demi_verdict) * n);
  const size_t o_t = o_v + s_v, s_t = al(sizeof(demi_dpor_trace_entry) * DEMI_DPOR_MAX_TRACE * n);
  const size_t o_tl = o_t + s_t, s_tl = al(4 * n);
  const size_t o_p = o_tl + s_tl, s_p = al(sizeof(demi_dpor_pair) * (size_t)par->max_pairs * n);
  const size_t o_np = o_p + s_p, s_np = al(4 * n);
  const size_t total = o_np + s_np;
  if (ctx->dpor_bytes < total) {
    if (ctx->d_dpor) (void)hipFree(ctx->d_dpor);
    ctx->d_dpor = nullptr; ctx->dpor_bytes = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_dpor, total));
    ctx->dpor_bytes = total;
  }
  unsigned char* base = static_cast<unsigned char*>(ctx->d_dpor);
  if (stride) HIP_TRY(ctx, hipMemcpy(base + o_pfx, prefixes, sizeof(demi_dpor_trace_entry) * (size_t)stride * n, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(base + o_pl, prefix_len, 4 * n, hipMemcpyHostToDevice));
  const size_t lds = k3_lds_bytes(h.code_len, ctx->n_dext, h.n_classes * h.n_msg_types, h.n_actors);
  HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k3_dpor), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = 0;
  HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k3_dpor, K3_WAVES * 64, lds));
  if (per_cu < 1) per_cu = 1;
  uint64_t blocks = (n + 63) / 64;   // a round of the backtrack queue is small: one wave per 64 interleavings
  const uint64_t resident = (uint64_t)ctx->num_cu * (uint64_t)per_cu;
  if (blocks > resident) blocks = resident;
  int rc = ensure_spill(ctx, blocks * K3_WAVES * 64, 2);
  if (rc) return rc;
  K3Args a;
  memset(&a, 0, sizeof a);
  a.model = ctx->d_model; a.ext = ctx->d_dext; a.n_ext = ctx->n_dext;
  a.prefixes = reinterpret_cast<const demi_dpor_trace_entry*>(base + o_pfx);
  a.prefix_len = reinterpret_cast<const uint32_t*>(base + o_pl);
  a.stride = stride; a.n = n;
  a.depth_bound = par->depth_bound; a.max_messages = par->max_messages;
  a.looking_for_valid = par->looking_for_valid; a.looking_for = par->looking_for; a.p_max = p_max;
  a.max_pairs = par->max_pairs;
  a.out = reinterpret_cast<demi_verdict*>(base + o_v);
  a.traces = reinterpret_cast<demi_dpor_trace_entry*>(base + o_t);
  a.trace_len = reinterpret_cast<uint32_t*>(base + o_tl);
  a.pairs = reinterpret_cast<demi_dpor_pair*>(base + o_p);
  a.n_pairs = reinterpret_cast<uint32_t*>(base + o_np);
  a.work_counter = ctx->d_counter; a.spill = ctx->d_spill;
  HIP_TRY(ctx, hipMemsetAsync(a.work_counter, 0, sizeof(unsigned long long), nullptr));
  hipLaunchKernelGGL(k3_dpor, dim3((unsigned)blocks), dim3(K3_WAVES * 64), lds, nullptr, a);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(out_verdicts, base + o_v, sizeof(demi_verdict) * n, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(out_traces, base + o_t, sizeof(demi_dpor_trace_entry) * DEMI_DPOR_MAX_TRACE * n, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(out_trace_len, base + o_tl, 4 * n, hipMemcpyDeviceToHost));
  if (par->max_pairs) HIP_TRY(ctx, hipMemcpy(out_pairs, base + o_p, sizeof(demi_dpor_pair) * (size_t)par->max_pairs * n, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(out_n_pairs, base + o_np, 4 * n, hipMemcpyDeviceToHost));
  return DEMI_OK;
}

// ----------------------------------------------------------------------------- K3: native exploration loop
extern "C" int demi_dpor_explore(demi_ctx* ctx, const demi_dpor_params* par, const demi_dpor_search* srch,
                                 demi_verdict* out_verdicts, uint32_t* out_prefix_len, uint32_t* out_rounds,
                                 demi_dpor_trace_entry* first_violation_trace, uint32_t* first_violation_len,
                                 demi_dpor_stats* stats) {
  if (!ctx) return DEMI_ERR_INVALID_ARG;
  if (!par || !srch || !out_verdicts || !out_prefix_len || !stats) return fail(ctx, DEMI_ERR_INVALID_ARG, "null pointer");
  if (srch->batch < 1 || srch->max_interleavings < 1) return fail(ctx, DEMI_ERR_INVALID_ARG, "batch and max_interleavings must be >= 1");
  auto run = [&](const demi_dpor_trace_entry* pf, const uint32_t* pl, uint32_t stride, uint64_t n, demi_verdict* vd,
                 demi_dpor_trace_entry* tr, uint32_t* tl, demi_dpor_pair* pr, uint32_t* np) {
    return demi_dpor_batch(ctx, pf, pl, stride, n, par, vd, tr, tl, pr, np);
  };
  // pinned result buffers: a round returns up to (4 KB trace + 4 B x max_pairs) per interleaving
  auto pinned_alloc = [](size_t bytes) -> void* {
    void* p = nullptr;
    return hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess ? p : nullptr;
  };
  auto pinned_free = [](void* p) { (void)hipHostFree(p); };
  return demi_host::explore_loop(run, par->max_pairs, srch, out_verdicts, out_prefix_len, out_rounds, first_violation_trace,
                                 first_violation_len, stats, nullptr, pinned_alloc, pinned_free);
}

extern "C" int demi_random_explore_violations(demi_ctx* ctx, uint64_t seed_base, uint64_t n, const demi_limits* limits,
                                              demi_violation* out, uint32_t cap, uint64_t* n_violations) {
  if (!ctx) return DEMI_ERR_INVALID_ARG;
  if (!n_violations || (!out && cap)) return fail(ctx, DEMI_ERR_INVALID_ARG, "null output");
  *n_violations = 0;
  if (n == 0) return DEMI_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->out_cap < n) {
    if (ctx->d_out) (void)hipFree(ctx->d_out);
    ctx->d_out = nullptr; ctx->out_cap = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_out, sizeof(demi_verdict) * n));
    ctx->out_cap = n;
  }
  if (ctx->viol_cap < cap || !ctx->d_viol) {
    if (ctx->d_viol) (void)hipFree(ctx->d_viol);
    ctx->d_viol = nullptr; ctx->viol_cap = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_viol, sizeof(demi_violation) * (size_t)(cap ? cap : 1)));
    ctx->viol_cap = cap;
  }
  unsigned long long* d_count = ctx->d_viol_count;
  int rc = demi_random_explore_dev(ctx, seed_base, nullptr, n, limits, ctx->d_out, nullptr);
  if (rc) return rc;
  rc = demi_collect_violations_dev(ctx, ctx->d_out, n, 0, ctx->d_viol, cap, d_count, nullptr);
  if (rc) return rc;
  unsigned long long cnt = 0;
  HIP_TRY(ctx, hipMemcpy(&cnt, d_count, sizeof cnt, hipMemcpyDeviceToHost));   // synchronises the stream
  *n_violations = cnt;
  const size_t k = cnt < cap ? (size_t)cnt : (size_t)cap;
  if (k) {
    HIP_TRY(ctx, hipMemcpy(out, ctx->d_viol, sizeof(demi_violation) * k, hipMemcpyDeviceToHost));
    std::sort(out, out + k, [](const demi_violation& a, const demi_violation& b) { return a.index < b.index; });
  }
  return DEMI_OK;
}
