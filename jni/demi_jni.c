/* jni/demi_jni.c — the JNI shim between DEMi's Scala adapter (scala/akka/dispatch/verification/gpu/DemiGpu.scala) and the
 * C ABI of libdemi_gpu.so (include/demi_gpu.h).  One Java_..._DemiGpu_<name> per entry point, nothing but marshalling:
 * primitive arrays are pinned with GetPrimitiveArrayCritical for the duration of the call (no JNI call happens in
 * between), structs travel as int / long arrays in field order.  Build: make -C jni (needs JAVA_HOME; without a JDK
 * `make -C jni check` compiles against jni/stub/jni.h).
 *
 * Array conventions (little-endian, same layouts as the C structs):
 *   events    byte[8 * n]     demi_ext_event          recorded  byte[12 * n]  demi_rec_event
 *   verdicts  long[2 * n]     demi_verdict (long 0 = flags | fingerprint << 32, long 1 = hash)
 *   masks     long[4 * n]     candidate subsequences  violations long[2 * n]  demi_violation (index, fingerprint | flags << 32)
 *   limits    int[8]          demi_limits             dporParams int[7]  demi_dpor_params     dporSearch int[6]  demi_dpor_search
 *   dporStats long[11]        demi_dpor_stats (kernel_ms as raw double bits)                                       */
#include <jni.h>
#include <stdint.h>
#include <string.h>

#include "demi_gpu.h"

#define FN(name) Java_akka_dispatch_verification_gpu_DemiGpu_##name
#define CTX(h) ((demi_ctx*)(intptr_t)(h))
#define PIN(arr) ((arr) ? (*e)->GetPrimitiveArrayCritical(e, (arr), NULL) : NULL)
#define UNPIN(arr, p, mode) do { if (arr) (*e)->ReleasePrimitiveArrayCritical(e, (arr), (p), (mode)); } while (0)

static demi_limits limits_of(const jint* l) {
  demi_limits x;
  x.max_messages = (uint32_t)l[0]; x.invariant_check_interval = (uint32_t)l[1]; x.p_max = (uint32_t)l[2];
  x.looking_for_valid = (uint32_t)l[3]; x.looking_for = (uint32_t)l[4]; x.populate_all = (uint32_t)l[5]; x.strategy = (uint32_t)l[6];
  x.filter_known_absents = (uint32_t)l[7];
  return x;
}
static demi_dpor_params dpor_params_of(const jint* p) {
  demi_dpor_params x;
  x.depth_bound = (uint32_t)p[0]; x.max_messages = (uint32_t)p[1]; x.looking_for_valid = (uint32_t)p[2]; x.looking_for = (uint32_t)p[3];
  x.p_max = (uint32_t)p[4]; x.max_pairs = (uint32_t)p[5]; x.prioritize_pending = (uint32_t)p[6];
  return x;
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
  (void)c;
  memset(&m, 0, sizeof m);
  m.n_actors = (uint32_t)nActors; m.n_classes = (uint32_t)nClasses;
  m.n_msg_types = (uint32_t)(*e)->GetArrayLength(e, msgClass);
  m.code_len = (uint32_t)(*e)->GetArrayLength(e, code);
  const jsize n_inv = (*e)->GetArrayLength(e, inv);     /* (before the critical section: no JNI calls inside one) */
  jint* iv = (jint*)PIN(inv);
  m.inv_kind = (uint32_t)iv[0]; m.inv_fa = (uint32_t)iv[1]; m.inv_va = (uint32_t)iv[2]; m.inv_fb = (uint32_t)iv[3];
  m.fp_match_mask = (uint32_t)iv[4];
  m.flags = n_inv > 5 ? (uint32_t)iv[5] : 0u;            /* a wide model: initState holds two words per actor */
  UNPIN(inv, iv, JNI_ABORT);
  m.msg_class = (const uint8_t*)PIN(msgClass);
  m.actor_class = (const uint8_t*)PIN(actorClass);
  m.handler_start = (const uint16_t*)PIN(handlerStart);
  m.code = (const uint32_t*)PIN(code);
  m.init_state = (const uint64_t*)PIN(initState);
  jint rc = demi_model_load(CTX(h), &m);
  UNPIN(initState, (void*)m.init_state, JNI_ABORT);
  UNPIN(code, (void*)m.code, JNI_ABORT);
  UNPIN(handlerStart, (void*)m.handler_start, JNI_ABORT);
  UNPIN(actorClass, (void*)m.actor_class, JNI_ABORT);
  UNPIN(msgClass, (void*)m.msg_class, JNI_ABORT);
  return rc;
}
JNIEXPORT jint JNICALL FN(modelSpecialize)(JNIEnv* e, jclass c, jlong h, jboolean enable) {
  (void)e; (void)c;
  return demi_model_specialize(CTX(h), enable ? 1 : 0);
}
JNIEXPORT jint JNICALL FN(traceLoad)(JNIEnv* e, jclass c, jlong h, jbyteArray events) {
  (void)c;
  const jsize n = (*e)->GetArrayLength(e, events) / 8;
  void* p = PIN(events);
  jint rc = demi_trace_load(CTX(h), (const demi_ext_event*)p, (uint32_t)n);
  UNPIN(events, p, JNI_ABORT);
  return rc;
}

/* ---- K1 */
JNIEXPORT jint JNICALL FN(randomExplore)(JNIEnv* e, jclass c, jlong h, jlong seedBase, jlong n, jintArray limits, jlongArray verdicts) {
  (void)c;
  jint* l = (jint*)PIN(limits);
  demi_limits lim = limits_of(l);
  UNPIN(limits, l, JNI_ABORT);
  void* o = PIN(verdicts);
  jint rc = demi_random_explore(CTX(h), (uint64_t)seedBase, NULL, (uint64_t)n, &lim, (demi_verdict*)o);
  UNPIN(verdicts, o, 0);
  return rc;
}
/* out: long[2 * cap]; counts: long[2] = { number flagged, lowest flagged index }; flagMask = DEMI_V_* bits */
JNIEXPORT jint JNICALL FN(randomExploreFlagged)(JNIEnv* e, jclass c, jlong h, jlong seedBase, jlong n, jintArray limits, jint flagMask,
                                               jlongArray out, jlongArray counts) {
  (void)c;
  jint* l = (jint*)PIN(limits);
  demi_limits lim = limits_of(l);
  UNPIN(limits, l, JNI_ABORT);
  const uint32_t cap = (uint32_t)((*e)->GetArrayLength(e, out) / 2);
  uint64_t n_flagged = 0, first = 0;
  void* o = PIN(out);
  jint rc = demi_random_explore_flagged(CTX(h), (uint64_t)seedBase, (uint64_t)n, &lim, (uint32_t)flagMask, (demi_violation*)o, cap,
                                        &n_flagged, &first);
  UNPIN(out, o, 0);
  jlong* cn = (jlong*)PIN(counts);
  cn[0] = (jlong)n_flagged; cn[1] = (jlong)first;
  UNPIN(counts, cn, 0);
  return rc;
}
/* verdict: long[2]; recorded: byte[12 * cap]; returns the number of recorded events, or a negative demi_status */
JNIEXPORT jint JNICALL FN(randomGetTrace)(JNIEnv* e, jclass c, jlong h, jlong seed, jintArray limits, jlongArray verdict, jbyteArray recorded) {
  (void)c;
  jint* l = (jint*)PIN(limits);
  demi_limits lim = limits_of(l);
  UNPIN(limits, l, JNI_ABORT);
  const uint32_t cap = (uint32_t)((*e)->GetArrayLength(e, recorded) / 12);
  uint32_t n_out = 0;
  void* v = PIN(verdict);
  void* r = PIN(recorded);
  jint rc = demi_random_get_trace(CTX(h), (uint64_t)seed, &lim, (demi_verdict*)v, (demi_rec_event*)r, cap, &n_out);
  UNPIN(recorded, r, 0);
  UNPIN(verdict, v, 0);
  return rc == DEMI_OK ? (jint)n_out : rc;
}

/* ---- K2 */
JNIEXPORT jint JNICALL FN(replayLoad)(JNIEnv* e, jclass c, jlong h, jbyteArray externals, jbyteArray recorded) {
  (void)c;
  const jsize ne = (*e)->GetArrayLength(e, externals) / 8, nr = (*e)->GetArrayLength(e, recorded) / 12;
  void* x = PIN(externals);
  void* r = PIN(recorded);
  jint rc = demi_replay_load(CTX(h), (const demi_ext_event*)x, (uint32_t)ne, (const demi_rec_event*)r, (uint32_t)nr);
  UNPIN(recorded, r, JNI_ABORT);
  UNPIN(externals, x, JNI_ABORT);
  return rc;
}
JNIEXPORT jint JNICALL FN(replayBatch)(JNIEnv* e, jclass c, jlong h, jlongArray masks, jintArray limits, jlongArray verdicts) {
  (void)c;
  jint* l = (jint*)PIN(limits);
  demi_limits lim = limits_of(l);
  UNPIN(limits, l, JNI_ABORT);
  const uint64_t n = (uint64_t)((*e)->GetArrayLength(e, masks) / 4);
  void* m = PIN(masks);
  void* o = PIN(verdicts);
  jint rc = demi_replay_batch_sharded(CTX(h), (const uint64_t*)m, n, &lim, (demi_verdict*)o);   /* = demi_replay_batch without a communicator */
  UNPIN(verdicts, o, 0);
  UNPIN(masks, m, JNI_ABORT);
  return rc;
}
JNIEXPORT jint JNICALL FN(replayRemovalBatch)(JNIEnv* e, jclass c, jlong h, jlongArray masksOrNull, jintArray skip, jintArray limits,
                                             jlongArray verdicts) {
  (void)c;
  jint* l = (jint*)PIN(limits);
  demi_limits lim = limits_of(l);
  UNPIN(limits, l, JNI_ABORT);
  const uint64_t n = (uint64_t)(*e)->GetArrayLength(e, skip);
  void* m = PIN(masksOrNull);
  void* s = PIN(skip);
  void* o = PIN(verdicts);
  jint rc = demi_replay_removal_batch(CTX(h), (const uint64_t*)m, (const uint32_t*)s, n, &lim, (demi_verdict*)o);
  UNPIN(verdicts, o, 0);
  UNPIN(skip, s, JNI_ABORT);
  UNPIN(masksOrNull, m, JNI_ABORT);
  return rc;
}
JNIEXPORT jint JNICALL FN(replayGetKept)(JNIEnv* e, jclass c, jlong h, jlongArray maskOrNull, jint skip, jintArray limits,
                                        jlongArray verdict, jbyteArray kept) {
  (void)c;
  jint* l = (jint*)PIN(limits);
  demi_limits lim = limits_of(l);
  UNPIN(limits, l, JNI_ABORT);
  void* m = PIN(maskOrNull);
  void* v = PIN(verdict);
  void* k = PIN(kept);
  jint rc = demi_replay_get_kept(CTX(h), (const uint64_t*)m, (uint32_t)skip, &lim, (demi_verdict*)v, (uint8_t*)k);
  UNPIN(kept, k, 0);
  UNPIN(verdict, v, 0);
  UNPIN(maskOrNull, m, JNI_ABORT);
  return rc;
}

/* ---- K3 */
JNIEXPORT jint JNICALL FN(dporLoad)(JNIEnv* e, jclass c, jlong h, jbyteArray externals) {
  (void)c;
  const jsize n = (*e)->GetArrayLength(e, externals) / 8;
  void* p = PIN(externals);
  jint rc = demi_dpor_load(CTX(h), (const demi_ext_event*)p, (uint32_t)n);
  UNPIN(externals, p, JNI_ABORT);
  return rc;
}
/* verdicts: long[2 * max_interleavings]; prefixLen, rounds: int[max_interleavings]; firstViolationTrace: byte[16 * 256];
 * stats: long[11] (the demi_dpor_stats fields in order, first_violation_len in stats[10]'s upper half is not used:
 * the length of the first violating trace is returned, or a negative demi_status)                                    */
JNIEXPORT jint JNICALL FN(dporExplore)(JNIEnv* e, jclass c, jlong h, jintArray params, jintArray search, jlongArray verdicts,
                                      jintArray prefixLen, jintArray rounds, jbyteArray firstViolationTrace, jlongArray stats) {
  (void)c;
  jint* p = (jint*)PIN(params);
  demi_dpor_params par = dpor_params_of(p);
  UNPIN(params, p, JNI_ABORT);
  jint* s = (jint*)PIN(search);
  demi_dpor_search srch;
  srch.batch = (uint32_t)s[0]; srch.max_interleavings = (uint32_t)s[1]; srch.stop_if_violation = (uint32_t)s[2];
  srch.track_history = (uint32_t)s[3]; srch.order = (uint32_t)s[4]; srch.cache_mb = (uint32_t)s[5];
  UNPIN(search, s, JNI_ABORT);
  demi_dpor_stats st;
  uint32_t vlen = 0;
  void* v = PIN(verdicts);
  void* pl = PIN(prefixLen);
  void* rd = PIN(rounds);
  void* ft = PIN(firstViolationTrace);
  jint rc = demi_dpor_explore(CTX(h), &par, &srch, (demi_verdict*)v, (uint32_t*)pl, (uint32_t*)rd, (demi_dpor_trace_entry*)ft, &vlen, &st);
  UNPIN(firstViolationTrace, ft, 0);
  UNPIN(rounds, rd, 0);
  UNPIN(prefixLen, pl, 0);
  UNPIN(verdicts, v, 0);
  jlong* o = (jlong*)PIN(stats);
  o[0] = (jlong)st.interleavings; o[1] = (jlong)st.launches; o[2] = (jlong)st.violations; o[3] = (jlong)st.first_violation;
  o[4] = (jlong)st.queue_len; o[5] = (jlong)st.exhausted; o[6] = (jlong)st.executed; o[7] = (jlong)st.cache_misses;
  memcpy(&o[8], &st.kernel_ms, sizeof(jlong));
  o[9] = (jlong)st.h2d_bytes; o[10] = (jlong)st.d2h_bytes;
  UNPIN(stats, o, 0);
  return rc == DEMI_OK ? (jint)vlen : rc;
}

/* ---- ProvenanceTracker.pruneConcurrentEvents for a batch of traces: traces byte[16 * stride * n] (demi_dpor_trace_entry),
 *      traceLen int[n], affected int[n] (actor bitmasks), keep long[4 * n] */
JNIEXPORT jint JNICALL FN(provenancePrune)(JNIEnv* e, jclass c, jlong h, jbyteArray traces, jintArray traceLen, jintArray affected,
                                          jint stride, jlongArray keep) {
  (void)c;
  const jsize n = (*e)->GetArrayLength(e, traceLen);
  void* t = PIN(traces);
  void* l = PIN(traceLen);
  void* a = PIN(affected);
  void* k = PIN(keep);
  jint rc = demi_provenance_prune(CTX(h), (const demi_dpor_trace_entry*)t, (const uint32_t*)l, (const uint32_t*)a, (uint32_t)stride,
                                  (uint64_t)n, (uint64_t*)k);
  UNPIN(keep, k, 0);
  UNPIN(affected, a, JNI_ABORT);
  UNPIN(traceLen, l, JNI_ABORT);
  UNPIN(traces, t, JNI_ABORT);
  return rc;
}

/* ---- multi-GPU: one JVM (and one ctx) per GPU; rank 0 obtains the id and sends the 128 bytes to the others */
JNIEXPORT jint JNICALL FN(commUniqueId)(JNIEnv* e, jclass c, jbyteArray id128) {
  (void)c;
  void* p = PIN(id128);
  jint rc = demi_comm_unique_id((demi_comm_id*)p);
  UNPIN(id128, p, 0);
  return rc;
}
JNIEXPORT jint JNICALL FN(commCreate)(JNIEnv* e, jclass c, jlong h, jbyteArray id128, jint rank, jint world) {
  (void)c;
  demi_comm_id id;
  void* p = PIN(id128);
  memcpy(&id, p, sizeof id);
  UNPIN(id128, p, JNI_ABORT);
  return demi_comm_create(CTX(h), &id, rank, world);
}
JNIEXPORT jint JNICALL FN(commDestroy)(JNIEnv* e, jclass c, jlong h) { (void)e; (void)c; return demi_comm_destroy(CTX(h)); }
/* out: long[2 * cap] merged violation set; count: long[1] */
JNIEXPORT jint JNICALL FN(randomExploreSharded)(JNIEnv* e, jclass c, jlong h, jlong seedBase, jlong nTotal, jintArray limits,
                                               jlongArray out, jlongArray count) {
  (void)c;
  jint* l = (jint*)PIN(limits);
  demi_limits lim = limits_of(l);
  UNPIN(limits, l, JNI_ABORT);
  const uint32_t cap = (uint32_t)((*e)->GetArrayLength(e, out) / 2);
  uint64_t n = 0;
  void* o = PIN(out);
  jint rc = demi_random_explore_sharded(CTX(h), (uint64_t)seedBase, (uint64_t)nTotal, &lim, (demi_violation*)o, cap, &n);
  UNPIN(out, o, 0);
  jlong* cn = (jlong*)PIN(count);
  cn[0] = (jlong)n;
  UNPIN(count, cn, 0);
  return rc;
}
