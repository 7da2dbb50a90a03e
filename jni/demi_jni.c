/* jni/demi_jni.c — the JNI shim between DEMi's Scala adapter (scala/akka/dispatch/verification/gpu/DemiGpu.scala) and the
 * C ABI of libdemi_gpu.so (include/demi_gpu.h).  One Java_..._DemiGpu_<name> per entry point, nothing but marshalling.
 *
 * Marshalling rules (round 3):
 *   - small structs travel as int / long arrays in field order and are COPIED with Get/Set<Type>ArrayRegion;
 *   - bulk buffers are accessed with Get<Type>ArrayElements / Release<Type>ArrayElements.  NOT GetPrimitiveArrayCritical: an
 *     entry point may run a whole exploration, compile a table with hiprtc, start host thread pools or take part in an
 *     RCCL collective, and a JNI critical region must neither block nor last (it holds off the collector);
 *   - every array length is checked against what the C entry point reads or writes BEFORE the call: a mismatched caller gets
 *     DEMI_ERR_INVALID_ARG instead of a corrupted Java heap;
 *   - a non-null array whose Get<Type>ArrayElements returns NULL (the JVM is out of memory, an exception is pending) is an
 *     error, never "absent": NULL means something to the C ABI (no conjoined atoms, no masks), so passing it on would silently
 *     change the call (LOST below);
 *   - JNI_OnLoad refuses a libdemi_gpu.so of another struct-layout generation (demi_abi_version() != DEMI_ABI_VERSION).
 * Build: make -C jni (needs JAVA_HOME; without a JDK `make -C jni check` compiles against jni/stub/jni.h).
 *
 * Array conventions (little-endian, same layouts as the C structs):
 *   events    byte[8 * n]     demi_ext_event          recorded  byte[16 * n]  demi_rec_event
 *   verdicts  long[2 * n]     demi_verdict (long 0 = flags | fingerprint << 32, long 1 = hash)
 *   masks     long[4 * n]     candidate subsequences  violations long[2 * n]  demi_violation (index, fingerprint | flags << 32)
 *   limits    int[9]          demi_limits             dporParams int[7]  demi_dpor_params     dporSearch int[9]  demi_dpor_search
 *   dporStats long[13]        demi_dpor_stats (kernel_ms as raw double bits; fetches last)                                       */
#include <jni.h>
#include <stdint.h>
#include <string.h>

#include "demi_gpu.h"

#define FN(name) Java_akka_dispatch_verification_gpu_DemiGpu_##name
#define CTX(h) ((demi_ctx*)(intptr_t)(h))
#define LEN(arr) ((arr) ? (int64_t)(*e)->GetArrayLength(e, (arr)) : (int64_t)-1)
/* bulk buffers (may be NULL where the C ABI takes NULL) */
#define BYTES(arr) ((arr) ? (void*)(*e)->GetByteArrayElements(e, (arr), NULL) : NULL)
#define SHORTS(arr) ((arr) ? (void*)(*e)->GetShortArrayElements(e, (arr), NULL) : NULL)
#define INTS(arr) ((arr) ? (void*)(*e)->GetIntArrayElements(e, (arr), NULL) : NULL)
#define LONGS(arr) ((arr) ? (void*)(*e)->GetLongArrayElements(e, (arr), NULL) : NULL)
#define LOST(arr, p) ((arr) != NULL && (p) == NULL)     /* the array exists and its elements could not be obtained */
/* Results go back through Set*ArrayRegion.  When a Get*ArrayElements failed (LOST) an OutOfMemoryError is pending, and with an
 * exception pending JNI allows only a short list of calls - the Release* calls among them, Set*ArrayRegion not (-Xcheck:jni
 * aborts on it): the outputs are skipped then, the caller sees the exception and DEMI_ERR_INVALID_ARG. */
#define SET_LONGS(arr, n, src) do { if (!(*e)->ExceptionCheck(e)) (*e)->SetLongArrayRegion(e, (arr), 0, (n), (src)); } while (0)
#define PUT_BYTES(arr, p, mode) do { if ((arr) && (p)) (*e)->ReleaseByteArrayElements(e, (arr), (jbyte*)(p), (mode)); } while (0)
#define PUT_SHORTS(arr, p, mode) do { if ((arr) && (p)) (*e)->ReleaseShortArrayElements(e, (arr), (jshort*)(p), (mode)); } while (0)
#define PUT_INTS(arr, p, mode) do { if ((arr) && (p)) (*e)->ReleaseIntArrayElements(e, (arr), (jint*)(p), (mode)); } while (0)
#define PUT_LONGS(arr, p, mode) do { if ((arr) && (p)) (*e)->ReleaseLongArrayElements(e, (arr), (jlong*)(p), (mode)); } while (0)

JNIEXPORT jint JNICALL JNI_OnLoad(JavaVM* vm, void* reserved) {
  (void)vm; (void)reserved;
  /* a library built from another generation of demi_gpu.h would read these arrays with other struct layouts */
  return demi_abi_version() == DEMI_ABI_VERSION ? JNI_VERSION_1_6 : JNI_ERR;
}

/* demi_limits from int[9] (a shorter array of an older adapter is refused, not read out of bounds) */
static int limits_of(JNIEnv* e, jintArray limits, demi_limits* x) {
  jint l[9];
  if (LEN(limits) != 9) return DEMI_ERR_INVALID_ARG;
  (*e)->GetIntArrayRegion(e, limits, 0, 9, l);
  x->max_messages = (uint32_t)l[0]; x->invariant_check_interval = (uint32_t)l[1]; x->p_max = (uint32_t)l[2];
  x->looking_for_valid = (uint32_t)l[3]; x->looking_for = (uint32_t)l[4]; x->populate_all = (uint32_t)l[5]; x->strategy = (uint32_t)l[6];
  x->filter_known_absents = (uint32_t)l[7]; x->executions_per_instance = (uint32_t)l[8];
  return DEMI_OK;
}
static int dpor_params_of(JNIEnv* e, jintArray params, demi_dpor_params* x) {
  jint p[7];
  if (LEN(params) != 7) return DEMI_ERR_INVALID_ARG;
  (*e)->GetIntArrayRegion(e, params, 0, 7, p);
  x->depth_bound = (uint32_t)p[0]; x->max_messages = (uint32_t)p[1]; x->looking_for_valid = (uint32_t)p[2]; x->looking_for = (uint32_t)p[3];
  x->p_max = (uint32_t)p[4]; x->max_pairs = (uint32_t)p[5]; x->prioritize_pending = (uint32_t)p[6];
  return DEMI_OK;
}

JNIEXPORT jlong JNICALL FN(ctxCreate)(JNIEnv* e, jclass c, jint device) {
  demi_ctx* ctx = NULL;
  (void)e; (void)c;
  return demi_ctx_create(device, &ctx) == DEMI_OK ? (jlong)(intptr_t)ctx : 0;
}
JNIEXPORT void JNICALL FN(ctxDestroy)(JNIEnv* e, jclass c, jlong h) { (void)e; (void)c; demi_ctx_destroy(CTX(h)); }
JNIEXPORT jstring JNICALL FN(lastError)(JNIEnv* e, jclass c, jlong h) { (void)c; return (*e)->NewStringUTF(e, demi_last_error(CTX(h))); }

/* demi_model_load: inv = { inv_kind, inv_fa, inv_va, inv_fb, fp_match_mask [, flags (DEMI_MODEL_WIDE)] } */
JNIEXPORT jint JNICALL FN(modelLoad)(JNIEnv* e, jclass c, jlong h, jint nActors, jbyteArray msgClass, jbyteArray actorClass,
                                     jint nClasses, jshortArray handlerStart, jintArray code, jlongArray initState, jintArray inv) {
  demi_model m;
  jint iv[6] = {0, 0, 0, 0, 0, 0};
  (void)c;
  memset(&m, 0, sizeof m);
  const int64_t n_inv = LEN(inv), n_types = LEN(msgClass);
  if (n_inv < 5 || n_inv > 6 || n_types < 1 || nActors < 1 || nClasses < 1) return DEMI_ERR_INVALID_ARG;
  (*e)->GetIntArrayRegion(e, inv, 0, (jsize)n_inv, iv);
  m.n_actors = (uint32_t)nActors; m.n_classes = (uint32_t)nClasses;
  m.n_msg_types = (uint32_t)n_types;
  m.code_len = (uint32_t)(LEN(code) < 0 ? 0 : LEN(code));
  m.inv_kind = (uint32_t)iv[0]; m.inv_fa = (uint32_t)iv[1]; m.inv_va = (uint32_t)iv[2]; m.inv_fb = (uint32_t)iv[3];
  m.fp_match_mask = (uint32_t)iv[4];
  m.flags = (uint32_t)iv[5];                             /* a wide model: initState holds two words per actor */
  /* what demi_model_load reads: actor_class[n_actors], handler_start[n_classes * n_msg_types], init_state[n_actors (x 2 wide)] */
  if (LEN(actorClass) < nActors || LEN(handlerStart) < (int64_t)nClasses * n_types ||
      LEN(initState) < (int64_t)nActors * ((m.flags & DEMI_MODEL_WIDE) ? 2 : 1) || LEN(code) < 0)
    return DEMI_ERR_INVALID_ARG;
  m.msg_class = (const uint8_t*)BYTES(msgClass);
  m.actor_class = (const uint8_t*)BYTES(actorClass);
  m.handler_start = (const uint16_t*)SHORTS(handlerStart);
  m.code = (const uint32_t*)INTS(code);
  m.init_state = (const uint64_t*)LONGS(initState);
  jint rc = (LOST(msgClass, m.msg_class) || LOST(actorClass, m.actor_class) || LOST(handlerStart, m.handler_start) || LOST(code, m.code) ||
             LOST(initState, m.init_state)) ? DEMI_ERR_INVALID_ARG : demi_model_load(CTX(h), &m);
  PUT_LONGS(initState, m.init_state, JNI_ABORT);
  PUT_INTS(code, m.code, JNI_ABORT);
  PUT_SHORTS(handlerStart, m.handler_start, JNI_ABORT);
  PUT_BYTES(actorClass, m.actor_class, JNI_ABORT);
  PUT_BYTES(msgClass, m.msg_class, JNI_ABORT);
  return rc;
}
JNIEXPORT jint JNICALL FN(modelSpecialize)(JNIEnv* e, jclass c, jlong h, jboolean enable) {
  (void)e; (void)c;
  return demi_model_specialize(CTX(h), enable ? 1 : 0);
}
JNIEXPORT jint JNICALL FN(traceLoad)(JNIEnv* e, jclass c, jlong h, jbyteArray events) {
  (void)c;
  const int64_t len = LEN(events);
  if (len < 0 || len % 8) return DEMI_ERR_INVALID_ARG;
  void* p = BYTES(events);
  jint rc = LOST(events, p) ? DEMI_ERR_INVALID_ARG : demi_trace_load(CTX(h), (const demi_ext_event*)p, (uint32_t)(len / 8));
  PUT_BYTES(events, p, JNI_ABORT);
  return rc;
}

/* demi_ext_payload_areas: the payload areas of the external events the next traceLoad / dporLoad loads (null: forget) */
JNIEXPORT jint JNICALL FN(extPayloadAreas)(JNIEnv* e, jclass c, jlong h, jlongArray areas) {
  (void)c;
  if (!areas) return demi_ext_payload_areas(CTX(h), NULL, 0);
  const int64_t len = LEN(areas);
  if (len < 0 || len > DEMI_MAX_EXT_EVENTS) return DEMI_ERR_INVALID_ARG;
  void* p = LONGS(areas);
  jint rc = LOST(areas, p) ? DEMI_ERR_INVALID_ARG : demi_ext_payload_areas(CTX(h), (const uint64_t*)p, (uint32_t)len);
  PUT_LONGS(areas, p, JNI_ABORT);
  return rc;
}

/* ---- K1 */
JNIEXPORT jint JNICALL FN(randomExplore)(JNIEnv* e, jclass c, jlong h, jlong seedBase, jlong n, jintArray limits, jlongArray verdicts) {
  demi_limits lim;
  (void)c;
  if (limits_of(e, limits, &lim) || n < 0 || LEN(verdicts) < 2 * (int64_t)n) return DEMI_ERR_INVALID_ARG;
  void* o = LONGS(verdicts);
  jint rc = LOST(verdicts, o) ? DEMI_ERR_INVALID_ARG : demi_random_explore(CTX(h), (uint64_t)seedBase, NULL, (uint64_t)n, &lim, (demi_verdict*)o);
  PUT_LONGS(verdicts, o, 0);
  return rc;
}
/* out: long[2 * cap]; counts: long[2] = { number flagged, lowest flagged index }; flagMask = DEMI_V_* bits */
JNIEXPORT jint JNICALL FN(randomExploreFlagged)(JNIEnv* e, jclass c, jlong h, jlong seedBase, jlong n, jintArray limits, jint flagMask,
                                               jlongArray out, jlongArray counts) {
  demi_limits lim;
  (void)c;
  if (limits_of(e, limits, &lim) || n < 0 || LEN(out) < 0 || LEN(out) % 2 || LEN(counts) != 2) return DEMI_ERR_INVALID_ARG;
  const uint32_t cap = (uint32_t)(LEN(out) / 2);
  uint64_t n_flagged = 0, first = 0;
  void* o = LONGS(out);
  jint rc = LOST(out, o) ? DEMI_ERR_INVALID_ARG
                         : demi_random_explore_flagged(CTX(h), (uint64_t)seedBase, (uint64_t)n, &lim, (uint32_t)flagMask, (demi_violation*)o, cap,
                                                       &n_flagged, &first);
  PUT_LONGS(out, o, 0);
  const jlong cn[2] = {(jlong)n_flagged, (jlong)first};
  SET_LONGS(counts, 2, cn);
  return rc;
}
/* explore() in pieces, two calls in flight in one context.  Returns the ticket (> 0) or a negative demi_status. */
JNIEXPORT jint JNICALL FN(randomExploreSubmit)(JNIEnv* e, jclass c, jlong h, jlong seedBase, jlong n, jintArray limits, jint flagMask) {
  demi_limits lim;
  uint32_t ticket = 0;
  (void)c;
  if (limits_of(e, limits, &lim) || n <= 0) return DEMI_ERR_INVALID_ARG;
  const int rc = demi_random_explore_submit(CTX(h), (uint64_t)seedBase, (uint64_t)n, &lim, (uint32_t)flagMask, 0u, &ticket);
  return rc ? rc : (jint)ticket;
}
/* out: long[2 * cap] (flagged entries, sorted by index); counts: long[2] = { number flagged, lowest flagged index } */
JNIEXPORT jint JNICALL FN(randomExploreWait)(JNIEnv* e, jclass c, jlong h, jint ticket, jlongArray out, jlongArray counts) {
  (void)c;
  if (ticket <= 0 || LEN(out) < 0 || LEN(out) % 2 || LEN(counts) != 2) return DEMI_ERR_INVALID_ARG;
  const uint32_t cap = (uint32_t)(LEN(out) / 2);
  uint64_t n_flagged = 0, first = 0;
  void* o = LONGS(out);
  jint rc = LOST(out, o) ? DEMI_ERR_INVALID_ARG
                         : demi_random_explore_wait(CTX(h), (uint32_t)ticket, NULL, (demi_violation*)o, cap, &n_flagged, &first);
  PUT_LONGS(out, o, 0);
  const jlong cn[2] = {(jlong)n_flagged, (jlong)first};
  SET_LONGS(counts, 2, cn);
  return rc;
}
/* verdict: long[2]; recorded: byte[16 * cap]; returns the number of recorded events, or a negative demi_status */
JNIEXPORT jint JNICALL FN(randomGetTrace)(JNIEnv* e, jclass c, jlong h, jlong seed, jintArray limits, jlongArray verdict, jbyteArray recorded) {
  demi_limits lim;
  demi_verdict v;
  (void)c;
  if (limits_of(e, limits, &lim) || LEN(verdict) != 2 || LEN(recorded) < 0 || LEN(recorded) % (jint)sizeof(demi_rec_event)) return DEMI_ERR_INVALID_ARG;
  const uint32_t cap = (uint32_t)(LEN(recorded) / (jint)sizeof(demi_rec_event));
  uint32_t n_out = 0;
  memset(&v, 0, sizeof v);
  void* r = BYTES(recorded);
  jint rc = LOST(recorded, r) ? DEMI_ERR_INVALID_ARG : demi_random_get_trace(CTX(h), (uint64_t)seed, &lim, &v, (demi_rec_event*)r, cap, &n_out);
  PUT_BYTES(recorded, r, 0);
  SET_LONGS(verdict, 2, (const jlong*)(const void*)&v);
  return rc == DEMI_OK ? (jint)n_out : rc;
}

/* the same for execution number execIndex of the carried-generator instance seeded `seed` */
JNIEXPORT jint JNICALL FN(randomGetTraceCarried)(JNIEnv* e, jclass c, jlong h, jlong seed, jint execIndex, jintArray limits, jlongArray verdict,
                                                jbyteArray recorded) {
  demi_limits lim;
  demi_verdict v;
  (void)c;
  if (limits_of(e, limits, &lim) || execIndex < 0 || LEN(verdict) != 2 || LEN(recorded) < 0 || LEN(recorded) % (jint)sizeof(demi_rec_event)) return DEMI_ERR_INVALID_ARG;
  const uint32_t cap = (uint32_t)(LEN(recorded) / (jint)sizeof(demi_rec_event));
  uint32_t n_out = 0, ran = 0;
  memset(&v, 0, sizeof v);
  void* r = BYTES(recorded);
  jint rc = LOST(recorded, r) ? DEMI_ERR_INVALID_ARG
                              : demi_random_get_trace_carried(CTX(h), (uint64_t)seed, (uint32_t)execIndex, &lim, &v, (demi_rec_event*)r, cap, &n_out, &ran);
  PUT_BYTES(recorded, r, 0);
  SET_LONGS(verdict, 2, (const jlong*)(const void*)&v);
  if (rc == DEMI_OK && ran != (uint32_t)execIndex) return DEMI_ERR_INVALID_ARG;      /* an earlier execution of the chain already violated */
  return rc == DEMI_OK ? (jint)n_out : rc;
}

/* ---- K2 */
JNIEXPORT jint JNICALL FN(replayLoad)(JNIEnv* e, jclass c, jlong h, jbyteArray externals, jbyteArray recorded) {
  (void)c;
  const int64_t le = LEN(externals), lr = LEN(recorded);
  if (le < 0 || le % 8 || lr < 0 || lr % (jint)sizeof(demi_rec_event)) return DEMI_ERR_INVALID_ARG;
  void* x = BYTES(externals);
  void* r = BYTES(recorded);
  jint rc = (LOST(externals, x) || LOST(recorded, r)) ? DEMI_ERR_INVALID_ARG
            : demi_replay_load(CTX(h), (const demi_ext_event*)x, (uint32_t)(le / 8), (const demi_rec_event*)r, (uint32_t)(lr / (jint)sizeof(demi_rec_event)));
  PUT_BYTES(recorded, r, JNI_ABORT);
  PUT_BYTES(externals, x, JNI_ABORT);
  return rc;
}
JNIEXPORT jint JNICALL FN(replayBatch)(JNIEnv* e, jclass c, jlong h, jlongArray masks, jintArray limits, jlongArray verdicts) {
  demi_limits lim;
  (void)c;
  if (limits_of(e, limits, &lim) || LEN(masks) < 0 || LEN(masks) % 4) return DEMI_ERR_INVALID_ARG;
  const uint64_t n = (uint64_t)(LEN(masks) / 4);
  if (LEN(verdicts) < 2 * (int64_t)n) return DEMI_ERR_INVALID_ARG;
  void* m = LONGS(masks);
  void* o = LONGS(verdicts);
  jint rc = (LOST(masks, m) || LOST(verdicts, o)) ? DEMI_ERR_INVALID_ARG
            : demi_replay_batch_sharded(CTX(h), (const uint64_t*)m, n, &lim, (demi_verdict*)o);   /* = demi_replay_batch without a communicator */
  PUT_LONGS(verdicts, o, 0);
  PUT_LONGS(masks, m, JNI_ABORT);
  return rc;
}
JNIEXPORT jint JNICALL FN(replayRemovalBatch)(JNIEnv* e, jclass c, jlong h, jlongArray masksOrNull, jintArray skip, jintArray limits,
                                             jlongArray verdicts) {
  demi_limits lim;
  (void)c;
  if (limits_of(e, limits, &lim) || LEN(skip) < 0) return DEMI_ERR_INVALID_ARG;
  const uint64_t n = (uint64_t)LEN(skip);
  if (LEN(verdicts) < 2 * (int64_t)n || (masksOrNull && LEN(masksOrNull) < 4 * (int64_t)n)) return DEMI_ERR_INVALID_ARG;
  void* m = LONGS(masksOrNull);
  void* s = INTS(skip);
  void* o = LONGS(verdicts);
  jint rc = (LOST(masksOrNull, m) || LOST(skip, s) || LOST(verdicts, o)) ? DEMI_ERR_INVALID_ARG
            : demi_replay_removal_batch(CTX(h), (const uint64_t*)m, (const uint32_t*)s, n, &lim, (demi_verdict*)o);
  PUT_LONGS(verdicts, o, 0);
  PUT_INTS(skip, s, JNI_ABORT);
  PUT_LONGS(masksOrNull, m, JNI_ABORT);
  return rc;
}
/* kept: byte[recorded events of the loaded execution] (demi_replay_recorded_len; a shorter array is refused) */
JNIEXPORT jint JNICALL FN(replayGetKept)(JNIEnv* e, jclass c, jlong h, jlongArray maskOrNull, jint skip, jintArray limits,
                                        jlongArray verdict, jbyteArray kept) {
  demi_limits lim;
  demi_verdict v;
  (void)c;
  if (limits_of(e, limits, &lim) || LEN(verdict) != 2 || LEN(kept) < 0 || (maskOrNull && LEN(maskOrNull) != 4)) return DEMI_ERR_INVALID_ARG;
  if ((uint64_t)LEN(kept) < (uint64_t)demi_replay_recorded_len(CTX(h))) return DEMI_ERR_CAPACITY;
  memset(&v, 0, sizeof v);
  void* m = LONGS(maskOrNull);
  void* k = BYTES(kept);
  jint rc = (LOST(maskOrNull, m) || LOST(kept, k)) ? DEMI_ERR_INVALID_ARG
            : demi_replay_get_kept(CTX(h), (const uint64_t*)m, (uint32_t)skip, &lim, &v, (uint8_t*)k);
  PUT_BYTES(kept, k, 0);
  PUT_LONGS(maskOrNull, m, JNI_ABORT);
  SET_LONGS(verdict, 2, (const jlong*)(const void*)&v);
  return rc;
}

/* ---- DDMin in one call (demi_ddmin).  params: int[4] (demi_ddmin_params); conjoinedOrNull: byte[>= n externals of replayLoad]; mcs: long[4];
 *      consultedOrNull: long[4 * cap] with passedOrNull: byte[cap]; stats: long[5] = consultations, launches, mcs_len, verified, replays */
JNIEXPORT jint JNICALL FN(ddmin)(JNIEnv* e, jclass c, jlong h, jintArray limits, jintArray params, jbyteArray conjoinedOrNull, jlongArray mcs,
                                jlongArray consultedOrNull, jbyteArray passedOrNull, jlongArray stats) {
  demi_limits lim;
  demi_ddmin_params par;
  demi_ddmin_stats st;
  jint pr[4];
  uint64_t out[4] = {0, 0, 0, 0};
  (void)c;
  if (limits_of(e, limits, &lim) || LEN(params) != 4 || LEN(mcs) != 4 || LEN(stats) != 5) return DEMI_ERR_INVALID_ARG;
  (*e)->GetIntArrayRegion(e, params, 0, 4, pr);
  par.depth = (uint32_t)pr[0]; par.max_candidates = (uint32_t)pr[1]; par.check_unmodified = (uint32_t)pr[2]; par.verify_mcs = (uint32_t)pr[3];
  /* demi_ddmin reads conjoined[0 .. n externals of the loaded execution): a shorter array is refused */
  if (conjoinedOrNull && LEN(conjoinedOrNull) < (int64_t)demi_replay_externals_len(CTX(h))) return DEMI_ERR_INVALID_ARG;
  uint32_t cap = 0;
  if (consultedOrNull) {
    if (LEN(consultedOrNull) % 4 || !passedOrNull || LEN(passedOrNull) < LEN(consultedOrNull) / 4) return DEMI_ERR_INVALID_ARG;
    cap = (uint32_t)(LEN(consultedOrNull) / 4);
  }
  memset(&st, 0, sizeof st);
  void* cj = BYTES(conjoinedOrNull);
  void* co = LONGS(consultedOrNull);
  void* pa = BYTES(passedOrNull);
  jint rc = (LOST(conjoinedOrNull, cj) || LOST(consultedOrNull, co) || LOST(passedOrNull, pa)) ? DEMI_ERR_INVALID_ARG
            : demi_ddmin(CTX(h), &lim, &par, (const uint8_t*)cj, out, (uint64_t*)co, (uint8_t*)pa, cap, NULL, 0, &st);
  PUT_BYTES(passedOrNull, pa, 0);
  PUT_LONGS(consultedOrNull, co, 0);
  PUT_BYTES(conjoinedOrNull, cj, JNI_ABORT);
  SET_LONGS(mcs, 4, (const jlong*)(const void*)out);
  jlong o[5];
  o[0] = (jlong)st.consultations; o[1] = (jlong)st.launches; o[2] = (jlong)st.mcs_len; o[3] = (jlong)st.verified; o[4] = (jlong)st.replays;
  SET_LONGS(stats, 5, o);
  return rc;
}

/* ---- RunnerUtils.randomDDMin in one call (demi_random_ddmin) on the externals of traceLoad.  params: int[6] = executions, depth,
 *      max_candidates, check_unmodified, verify_mcs, sequential; conjoinedOrNull: byte[>= n externals]; mcs: long[4]; consultedOrNull:
 *      long[4 * cap] with passedOrNull: byte[cap]; stats: long[5] = consultations, launches, mcs_len, verified, executions run */
JNIEXPORT jint JNICALL FN(randomDDMin)(JNIEnv* e, jclass c, jlong h, jlong seedBase, jintArray limits, jintArray params, jint nExternals,
                                      jbyteArray conjoinedOrNull, jlongArray mcs, jlongArray consultedOrNull, jbyteArray passedOrNull,
                                      jlongArray stats) {
  demi_limits lim;
  demi_random_ddmin_params par;
  demi_ddmin_stats st;
  jint pr[6];
  uint64_t out[4] = {0, 0, 0, 0};
  (void)c;
  if (limits_of(e, limits, &lim) || LEN(params) != 6 || LEN(mcs) != 4 || LEN(stats) != 5 || nExternals < 0) return DEMI_ERR_INVALID_ARG;
  (*e)->GetIntArrayRegion(e, params, 0, 6, pr);
  memset(&par, 0, sizeof par);
  par.executions = (uint32_t)pr[0]; par.depth = (uint32_t)pr[1]; par.max_candidates = (uint32_t)pr[2];
  par.check_unmodified = (uint32_t)pr[3]; par.verify_mcs = (uint32_t)pr[4]; par.sequential = (uint32_t)pr[5];
  /* demi_random_ddmin reads conjoined[0 .. n externals of the LOADED trace): the library says how many that is; a caller that
   * states another count, or passes a shorter array, is refused before anything is read */
  if ((uint32_t)nExternals != demi_trace_len(CTX(h))) return DEMI_ERR_INVALID_ARG;
  if (conjoinedOrNull && LEN(conjoinedOrNull) < (int64_t)demi_trace_len(CTX(h))) return DEMI_ERR_INVALID_ARG;
  uint32_t cap = 0;
  if (consultedOrNull) {
    if (LEN(consultedOrNull) % 4 || !passedOrNull || LEN(passedOrNull) < LEN(consultedOrNull) / 4) return DEMI_ERR_INVALID_ARG;
    cap = (uint32_t)(LEN(consultedOrNull) / 4);
  }
  memset(&st, 0, sizeof st);
  void* cj = BYTES(conjoinedOrNull);
  void* co = LONGS(consultedOrNull);
  void* pa = BYTES(passedOrNull);
  jint rc = (LOST(conjoinedOrNull, cj) || LOST(consultedOrNull, co) || LOST(passedOrNull, pa)) ? DEMI_ERR_INVALID_ARG
            : demi_random_ddmin(CTX(h), (uint64_t)seedBase, &lim, &par, (const uint8_t*)cj, out, (uint64_t*)co, (uint8_t*)pa, cap, NULL, 0, &st);
  PUT_BYTES(passedOrNull, pa, 0);
  PUT_LONGS(consultedOrNull, co, 0);
  PUT_BYTES(conjoinedOrNull, cj, JNI_ABORT);
  SET_LONGS(mcs, 4, (const jlong*)(const void*)out);
  jlong o[5];
  o[0] = (jlong)st.consultations; o[1] = (jlong)st.launches; o[2] = (jlong)st.mcs_len; o[3] = (jlong)st.verified; o[4] = (jlong)st.replays;
  SET_LONGS(stats, 5, o);
  return rc;
}

/* ---- RandomScheduler.test for a batch of subsequences of the loaded trace (demi_random_explore_candidates).  masks: long[4 * n];
 *      verdictsOrNull: long[2 * n * executions]; flags: int[n] (bit 0: some execution violates, bit 1: some execution aborted) */
JNIEXPORT jint JNICALL FN(randomExploreCandidates)(JNIEnv* e, jclass c, jlong h, jlong seedBase, jlongArray masks, jint executions,
                                                  jintArray limits, jlongArray verdictsOrNull, jintArray flags) {
  demi_limits lim;
  (void)c;
  const int64_t n = LEN(masks) / 4;
  if (limits_of(e, limits, &lim) || LEN(masks) % 4 || executions <= 0 || !flags || LEN(flags) < n ||
      (verdictsOrNull && LEN(verdictsOrNull) < 2 * n * (int64_t)executions))
    return DEMI_ERR_INVALID_ARG;
  void* m = LONGS(masks);
  void* v = LONGS(verdictsOrNull);
  void* f = INTS(flags);
  jint rc = (LOST(masks, m) || LOST(verdictsOrNull, v) || LOST(flags, f)) ? DEMI_ERR_INVALID_ARG
            : demi_random_explore_candidates(CTX(h), (uint64_t)seedBase, (const uint64_t*)m, (uint32_t)n, (uint32_t)executions, &lim,
                                             (demi_verdict*)v, (uint32_t*)f);
  PUT_INTS(flags, f, 0);
  PUT_LONGS(verdictsOrNull, v, 0);
  PUT_LONGS(masks, m, JNI_ABORT);
  return rc;
}

/* ---- RunnerUtils.editDistanceDporDDMin in one call (demi_edit_distance_dpor_ddmin).  externals: byte[8 * n]; initialTrace: byte[16 * m]
 *      (demi_dpor_trace_entry); dporParams: int[7]; params: int[7] = max_max_distance, stop_at_size, check_unmodified, ignore_quiescence,
 *      verify_mcs, batch, budget; mcs: long[4]; consultedOrNull: long[4 * cap] with passedOrNull: byte[cap] and distanceOrNull: int[cap];
 *      violationTraceOrNull: byte[16 * 256]; stats: long[8 + 32] = replays, interleavings, consultations, instances, passes, mcs_len,
 *      verified, violation_len, pass_distance[16], pass_mcs_len[16] */
JNIEXPORT jint JNICALL FN(editDistanceDporDDMin)(JNIEnv* e, jclass c, jlong h, jbyteArray externals, jbyteArray initialTrace, jintArray dporParams,
                                                jintArray params, jlongArray mcs, jlongArray consultedOrNull, jbyteArray passedOrNull,
                                                jintArray distanceOrNull, jbyteArray violationTraceOrNull, jlongArray stats) {
  demi_dpor_params par;
  demi_incddmin_params ip;
  demi_incddmin_stats st;
  jint pr[7];
  uint64_t out[4] = {0, 0, 0, 0};
  (void)c;
  const int64_t n_ext = LEN(externals), n_init = LEN(initialTrace);
  if (dpor_params_of(e, dporParams, &par) || LEN(params) != 7 || LEN(mcs) != 4 || LEN(stats) != 40 || n_ext < 0 || n_ext % 8 ||
      n_init <= 0 || n_init % (int64_t)sizeof(demi_dpor_trace_entry))
    return DEMI_ERR_INVALID_ARG;
  (*e)->GetIntArrayRegion(e, params, 0, 7, pr);
  memset(&ip, 0, sizeof ip);
  ip.max_max_distance = (uint32_t)pr[0]; ip.stop_at_size = (uint32_t)pr[1]; ip.check_unmodified = (uint32_t)pr[2];
  ip.ignore_quiescence = (uint32_t)pr[3]; ip.verify_mcs = (uint32_t)pr[4]; ip.batch = (uint32_t)pr[5]; ip.budget = (uint32_t)pr[6];
  uint32_t cap = 0;
  if (consultedOrNull) {
    if (LEN(consultedOrNull) % 4 || !passedOrNull || LEN(passedOrNull) < LEN(consultedOrNull) / 4 ||
        (distanceOrNull && LEN(distanceOrNull) < LEN(consultedOrNull) / 4))
      return DEMI_ERR_INVALID_ARG;
    cap = (uint32_t)(LEN(consultedOrNull) / 4);
  }
  if (violationTraceOrNull && LEN(violationTraceOrNull) < (int64_t)sizeof(demi_dpor_trace_entry) * DEMI_DPOR_MAX_TRACE) return DEMI_ERR_INVALID_ARG;
  memset(&st, 0, sizeof st);
  void* ex = BYTES(externals);
  void* it = BYTES(initialTrace);
  void* co = LONGS(consultedOrNull);
  void* pa = BYTES(passedOrNull);
  void* di = consultedOrNull ? INTS(distanceOrNull) : NULL;
  void* vt = BYTES(violationTraceOrNull);
  jint rc = (LOST(externals, ex) || LOST(initialTrace, it) || LOST(consultedOrNull, co) || LOST(passedOrNull, pa) ||
             (consultedOrNull && LOST(distanceOrNull, di)) || LOST(violationTraceOrNull, vt)) ? DEMI_ERR_INVALID_ARG
            : demi_edit_distance_dpor_ddmin(CTX(h), (const demi_ext_event*)ex, (uint32_t)(n_ext / 8), (const demi_dpor_trace_entry*)it,
                                            (uint32_t)(n_init / (int64_t)sizeof(demi_dpor_trace_entry)), &par, &ip, out, (uint64_t*)co,
                                            (uint8_t*)pa, (uint32_t*)di, cap, (demi_dpor_trace_entry*)vt, &st);
  PUT_BYTES(violationTraceOrNull, vt, 0);
  if (consultedOrNull) PUT_INTS(distanceOrNull, di, 0);
  PUT_BYTES(passedOrNull, pa, 0);
  PUT_LONGS(consultedOrNull, co, 0);
  PUT_BYTES(initialTrace, it, JNI_ABORT);
  PUT_BYTES(externals, ex, JNI_ABORT);
  SET_LONGS(mcs, 4, (const jlong*)(const void*)out);
  jlong o[40];
  o[0] = (jlong)st.replays; o[1] = (jlong)st.interleavings; o[2] = (jlong)st.consultations; o[3] = (jlong)st.instances;
  o[4] = (jlong)st.passes; o[5] = (jlong)st.mcs_len; o[6] = (jlong)st.verified; o[7] = (jlong)st.violation_len;
  for (int i = 0; i < 16; i++) { o[8 + i] = (jlong)st.pass_distance[i]; o[24 + i] = (jlong)st.pass_mcs_len[i]; }
  SET_LONGS(stats, 40, o);
  return rc;
}

/* ---- K3 */
JNIEXPORT jint JNICALL FN(dporLoad)(JNIEnv* e, jclass c, jlong h, jbyteArray externals) {
  (void)c;
  const int64_t len = LEN(externals);
  if (len < 0 || len % 8) return DEMI_ERR_INVALID_ARG;
  void* p = BYTES(externals);
  jint rc = LOST(externals, p) ? DEMI_ERR_INVALID_ARG : demi_dpor_load(CTX(h), (const demi_ext_event*)p, (uint32_t)(len / 8));
  PUT_BYTES(externals, p, JNI_ABORT);
  return rc;
}
/* verdicts: long[2 * max_interleavings]; prefixLen, rounds: int[max_interleavings]; firstViolationTrace: byte[16 * 256];
 * stats: long[13] (the demi_dpor_stats fields in order, `fetches` last; the length of the first violating trace is returned, or a negative
 * demi_status).  A whole exploration runs inside this call: no array is pinned critically (see the header). */
JNIEXPORT jint JNICALL FN(dporExplore)(JNIEnv* e, jclass c, jlong h, jintArray params, jintArray search, jlongArray verdicts,
                                      jintArray prefixLen, jintArray rounds, jbyteArray firstViolationTrace, jlongArray stats) {
  demi_dpor_params par;
  demi_dpor_search srch;
  jint s[9];
  (void)c;
  if (dpor_params_of(e, params, &par) || LEN(search) != 9 || LEN(stats) != 13) return DEMI_ERR_INVALID_ARG;
  (*e)->GetIntArrayRegion(e, search, 0, 9, s);
  srch.batch = (uint32_t)s[0]; srch.max_interleavings = (uint32_t)s[1]; srch.stop_if_violation = (uint32_t)s[2];
  srch.track_history = (uint32_t)s[3]; srch.order = (uint32_t)s[4]; srch.cache_mb = (uint32_t)s[5];
  srch.ordering = (uint32_t)s[6]; srch.max_distance_plus1 = (uint32_t)s[7]; srch.resume = (uint32_t)s[8];
  const int64_t cap = (int64_t)srch.max_interleavings;
  if (LEN(verdicts) < 2 * cap || LEN(prefixLen) < cap || (rounds && LEN(rounds) < cap) ||
      (firstViolationTrace && LEN(firstViolationTrace) < (int64_t)sizeof(demi_dpor_trace_entry) * DEMI_DPOR_MAX_TRACE))
    return DEMI_ERR_INVALID_ARG;
  demi_dpor_stats st;
  uint32_t vlen = 0;
  memset(&st, 0, sizeof st);
  void* v = LONGS(verdicts);
  void* pl = INTS(prefixLen);
  void* rd = INTS(rounds);
  void* ft = BYTES(firstViolationTrace);
  jint rc = (LOST(verdicts, v) || LOST(prefixLen, pl) || LOST(rounds, rd) || LOST(firstViolationTrace, ft)) ? DEMI_ERR_INVALID_ARG
            : demi_dpor_explore(CTX(h), &par, &srch, (demi_verdict*)v, (uint32_t*)pl, (uint32_t*)rd, (demi_dpor_trace_entry*)ft, &vlen, &st);
  PUT_BYTES(firstViolationTrace, ft, 0);
  PUT_INTS(rounds, rd, 0);
  PUT_INTS(prefixLen, pl, 0);
  PUT_LONGS(verdicts, v, 0);
  jlong o[13];
  o[0] = (jlong)st.interleavings; o[1] = (jlong)st.launches; o[2] = (jlong)st.violations; o[3] = (jlong)st.first_violation;
  o[4] = (jlong)st.queue_len; o[5] = (jlong)st.exhausted; o[6] = (jlong)st.executed; o[7] = (jlong)st.cache_misses;
  memcpy(&o[8], &st.kernel_ms, sizeof(jlong));
  o[9] = (jlong)st.h2d_bytes; o[10] = (jlong)st.d2h_bytes; o[11] = (jlong)st.backtrack_points; o[12] = (jlong)st.fetches;
  SET_LONGS(stats, 13, o);
  return rc == DEMI_OK ? (jint)vlen : rc;
}

/* ---- demi_dpor_explored: what interleaving `index` of the last dporExplore was.  nextTrace, trace: byte[16 * 256]
 *      (demi_dpor_trace_entry); lens: long[3] = next-trace length, shared length, executed-trace length */
JNIEXPORT jint JNICALL FN(dporExplored)(JNIEnv* e, jclass c, jlong h, jlong index, jbyteArray nextTrace, jbyteArray trace, jlongArray lens) {
  (void)c;
  const int64_t need = (int64_t)sizeof(demi_dpor_trace_entry) * DEMI_DPOR_MAX_TRACE;
  if (index < 0 || !nextTrace || !trace || LEN(nextTrace) < need || LEN(trace) < need || LEN(lens) != 3) return DEMI_ERR_INVALID_ARG;
  uint32_t nl = 0, sl = 0, tl = 0;
  void* nt = BYTES(nextTrace);
  void* tr = BYTES(trace);
  jint rc = (LOST(nextTrace, nt) || LOST(trace, tr)) ? DEMI_ERR_INVALID_ARG
            : demi_dpor_explored(CTX(h), (uint64_t)index, (demi_dpor_trace_entry*)nt, &nl, &sl, (demi_dpor_trace_entry*)tr, &tl);
  PUT_BYTES(trace, tr, 0);
  PUT_BYTES(nextTrace, nt, 0);
  jlong o[3] = {(jlong)nl, (jlong)sl, (jlong)tl};
  SET_LONGS(lens, 3, o);
  return rc;
}

/* ---- ProvenanceTracker.pruneConcurrentEvents for a batch of traces: traces byte[16 * stride * n] (demi_dpor_trace_entry),
 *      traceLen int[n], affected int[n] (actor bitmasks), keep long[4 * n] */
JNIEXPORT jint JNICALL FN(provenancePrune)(JNIEnv* e, jclass c, jlong h, jbyteArray traces, jintArray traceLen, jintArray affected,
                                          jint stride, jlongArray keep) {
  (void)c;
  const int64_t n = LEN(traceLen);
  if (n < 0 || stride < 0 || LEN(affected) < n || LEN(keep) < 4 * n ||
      LEN(traces) < (int64_t)sizeof(demi_dpor_trace_entry) * (int64_t)stride * n)
    return DEMI_ERR_INVALID_ARG;
  void* t = BYTES(traces);
  void* l = INTS(traceLen);
  void* a = INTS(affected);
  void* k = LONGS(keep);
  jint rc = (LOST(traces, t) || LOST(traceLen, l) || LOST(affected, a) || LOST(keep, k)) ? DEMI_ERR_INVALID_ARG
            : demi_provenance_prune(CTX(h), (const demi_dpor_trace_entry*)t, (const uint32_t*)l, (const uint32_t*)a, (uint32_t)stride,
                                    (uint64_t)n, (uint64_t*)k);
  PUT_LONGS(keep, k, 0);
  PUT_INTS(affected, a, JNI_ABORT);
  PUT_INTS(traceLen, l, JNI_ABORT);
  PUT_BYTES(traces, t, JNI_ABORT);
  return rc;
}

/* ArvindDistanceOrdering.init / setInitialTrace for the following dporExplore calls: originalKeys long[n] (node keys) or null,
 * initialTrace byte[16 * m] (demi_dpor_trace_entry) or null */
JNIEXPORT jint JNICALL FN(dporSetTraces)(JNIEnv* e, jclass c, jlong h, jlongArray originalKeysOrNull, jbyteArray initialTraceOrNull) {
  (void)c;
  const int64_t lk = originalKeysOrNull ? LEN(originalKeysOrNull) : 0, lt = initialTraceOrNull ? LEN(initialTraceOrNull) : 0;
  if (lk < 0 || lt < 0 || lt % (int64_t)sizeof(demi_dpor_trace_entry)) return DEMI_ERR_INVALID_ARG;
  void* k = LONGS(originalKeysOrNull);
  void* t = BYTES(initialTraceOrNull);
  jint rc = (LOST(originalKeysOrNull, k) || LOST(initialTraceOrNull, t)) ? DEMI_ERR_INVALID_ARG
            : demi_dpor_set_traces(CTX(h), (const uint64_t*)k, (uint32_t)lk, (const demi_dpor_trace_entry*)t,
                                   (uint32_t)(lt / (int64_t)sizeof(demi_dpor_trace_entry)));
  PUT_BYTES(initialTraceOrNull, t, JNI_ABORT);
  PUT_LONGS(originalKeysOrNull, k, JNI_ABORT);
  return rc;
}

/* ---- multi-GPU: one JVM (and one ctx) per GPU; rank 0 obtains the id and sends the 128 bytes to the others */
JNIEXPORT jint JNICALL FN(commUniqueId)(JNIEnv* e, jclass c, jbyteArray id128) {
  demi_comm_id id;
  (void)c;
  if (LEN(id128) != (int64_t)sizeof id) return DEMI_ERR_INVALID_ARG;
  jint rc = demi_comm_unique_id(&id);
  if (rc == DEMI_OK) (*e)->SetByteArrayRegion(e, id128, 0, (jsize)sizeof id, (const jbyte*)(const void*)&id);
  return rc;
}
JNIEXPORT jint JNICALL FN(commCreate)(JNIEnv* e, jclass c, jlong h, jbyteArray id128, jint rank, jint world) {
  demi_comm_id id;
  (void)c;
  if (LEN(id128) != (int64_t)sizeof id) return DEMI_ERR_INVALID_ARG;
  (*e)->GetByteArrayRegion(e, id128, 0, (jsize)sizeof id, (jbyte*)(void*)&id);
  return demi_comm_create(CTX(h), &id, rank, world);
}
JNIEXPORT jint JNICALL FN(commDestroy)(JNIEnv* e, jclass c, jlong h) { (void)e; (void)c; return demi_comm_destroy(CTX(h)); }
/* out: long[2 * cap] merged violation set; count: long[1] */
JNIEXPORT jint JNICALL FN(randomExploreSharded)(JNIEnv* e, jclass c, jlong h, jlong seedBase, jlong nTotal, jintArray limits,
                                               jlongArray out, jlongArray count) {
  demi_limits lim;
  (void)c;
  if (limits_of(e, limits, &lim) || nTotal < 0 || LEN(out) < 0 || LEN(out) % 2 || LEN(count) != 1) return DEMI_ERR_INVALID_ARG;
  const uint32_t cap = (uint32_t)(LEN(out) / 2);
  uint64_t n = 0;
  void* o = LONGS(out);
  jint rc = LOST(out, o) ? DEMI_ERR_INVALID_ARG
                         : demi_random_explore_sharded(CTX(h), (uint64_t)seedBase, (uint64_t)nTotal, &lim, (demi_violation*)o, cap, &n);
  PUT_LONGS(out, o, 0);
  const jlong cn = (jlong)n;
  SET_LONGS(count, 1, &cn);
  return rc;
}
