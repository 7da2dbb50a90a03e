/*
 * demi_oracle.h — CPU restatement of DEMi's schedule-exploration hot path.  TEST INFRASTRUCTURE.
 *
 * This is the parity oracle and the timed CPU baseline ("port").  It is NOT part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY UNPINNED versus the JVM reference: NetSys/demi has no tests, golden vectors or
 * known-answer fixtures, cannot be built here (no JVM / sbt / AspectJ / Akka 2.3.6), and the
 * transition function it schedules (akka-raft, Spark actors) lives in another repository.  The
 * only externally pinned arithmetic is java.util.Random (JDK javadoc LCG; known answers in
 * tests/golden/jrandom_kat.json).  Everything else restates the Scala sources function by
 * function; each function below cites the file:line it follows (paths relative to
 * /root/reference/src/main/scala/verification/).
 * What stands in for the missing JVM: literal Python transliterations of the Scala schedulers (tests/test_*_transliteration_cpu.py:
 * the reference's own containers and control flow, sharing only the actors' row interpreter with this file).  They have executed
 * every timed workload once - all 2^20 schedules of config 2's step, all 2^20 candidates of config 4, all 60 332 interleavings of
 * config 3, the first 6 000 of config 5's pipeline - and this oracle gives their bytes (tests/golden/ *_transliteration.json, written
 * by tools/check_*_transliteration.py; the CPU suite re-checks the oracle against the records).
 */
#ifndef DEMI_ORACLE_H
#define DEMI_ORACLE_H

#include "../include/demi_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* java.util.Random */
typedef struct { uint64_t s; } orc_jrandom;
void    orc_jrandom_seed(orc_jrandom* r, uint64_t seed);
int32_t orc_jrandom_next(orc_jrandom* r, int bits);
int32_t orc_jrandom_next_int(orc_jrandom* r);
int32_t orc_jrandom_next_int_bound(orc_jrandom* r, int32_t bound);
double  orc_jrandom_next_double(orc_jrandom* r);

/* Model validation; returns 0 or DEMI_ERR_INVALID_MODEL and writes a reason into err. */
int orc_model_validate(const demi_model* m, char* err, size_t err_cap);
int orc_trace_validate(const demi_model* m, const demi_ext_event* ev, uint32_t n, char* err, size_t err_cap);

/* One handler application: delta(state of `me`, message) -> new state + effects appended to `fx`. */
typedef struct {
  uint8_t kind;   /* 0 send, 1 tset, 2 trep, 3 tcancel */
  uint8_t target; /* send: receiver */
  uint8_t msg_type;
  uint16_t p0, p1; /* the SEND row's two operands (8 bits used unless the model is DEMI_MODEL_WIDE) */
  uint64_t area;   /* send: the message's whole payload as demi_rec_event stores it (p0 | p1 << 16 | p_hi << 32): the operands
                      truncated to the model's field width, and the staged fields P2.. of a DEMI_MODEL_PAYLOADS model */
} orc_effect;      /* 16 bytes */
/* state: the actor's one word, or its two words for a DEMI_MODEL_WIDE model */
int orc_vm_run(const demi_model* m, uint32_t me, uint64_t* state, uint8_t msg_type, uint8_t src,
               uint16_t p0, uint16_t p1, uint32_t exists_mask, orc_effect* fx, uint32_t fx_cap, orc_jrandom* app);

/* the same with the delivered message's payload as an area (every field of a DEMI_MODEL_PAYLOADS model); orc_pay_area packs
 * DEMI_MAX_PAYLOADS field values the way a SEND does */
int orc_vm_run_area(const demi_model* m, uint32_t me, uint64_t* state, uint8_t msg_type, uint8_t src,
                    uint64_t area, uint32_t exists_mask, orc_effect* fx, uint32_t fx_cap, orc_jrandom* app);
uint64_t orc_pay_area(const demi_model* m, const uint16_t* fields);

/* The payload areas of the external Sends of the trace the NEXT calls are given (a DEMI_MODEL_PAYLOADS table: demi_ext_event has
 * room for two fields); NULL clears.  Process-wide. */
void orc_set_ext_areas(const uint64_t* areas, uint32_t n);

/* Invariant: returns the fingerprint code (0 = holds). */
uint32_t orc_invariant(const demi_model* m, const uint64_t* states, uint32_t exists_mask);

/* One RandomScheduler execution.  rec may be NULL. */
int orc_random_execute(const demi_model* m, const demi_ext_event* trace, uint32_t n_ev, uint64_t seed,
                       const demi_limits* lim, demi_verdict* out, demi_rec_event* rec, uint32_t rec_cap,
                       uint32_t* n_rec, uint64_t* final_states);

/* n executions; seeds == NULL -> seed_base + i.  n_threads > 1 splits the index range. */
int orc_random_execute_carried(const demi_model* m, const demi_ext_event* trace, uint32_t n_ev, uint64_t seed, uint32_t exec_index,
                               const demi_limits* lim, demi_verdict* out, demi_rec_event* rec, uint32_t rec_cap,
                               uint32_t* n_rec, uint32_t* ran);
int orc_random_explore(const demi_model* m, const demi_ext_event* trace, uint32_t n_ev, uint64_t seed_base,
                       const uint64_t* seeds, uint64_t n, const demi_limits* lim, demi_verdict* out,
                       int n_threads);

/* ---- K2: DDMin's replay oracle, STSScheduler.test without peek (V/schedulers/STSScheduler.scala:199-310).
 * original_ext/original_rec: the externals and the recorded EventTrace of the failing execution.
 * mask: bit i set = external event i is in the candidate subsequence (WaitQuiescence bits are ignored:
 * RunnerUtils.stsSchedDDMin strips them, V/RunnerUtils.scala:680-684).  lim->looking_for is the target. */
int orc_sts_replay(const demi_model* m, const demi_ext_event* original_ext, uint32_t n_ext,
                   const demi_rec_event* original_rec, uint32_t n_rec, const uint64_t mask[4],
                   const demi_limits* lim, demi_verdict* out, uint32_t* n_ignored);
int orc_sts_replay_batch(const demi_model* m, const demi_ext_event* original_ext, uint32_t n_ext,
                         const demi_rec_event* original_rec, uint32_t n_rec, const uint64_t* masks, uint64_t n,
                         const demi_limits* lim, demi_verdict* out, int n_threads);

/* ---- K2 with one removed delivery per candidate: the replay STSSchedMinimizer asks for
 * (V/minification/internal_minimization/ScheduleCheckers.scala:54-57, OneAtATimeRemoval.scala:57-124).
 * skip = index in original_rec of the MsgEvent removed (0xFFFFFFFF = none); masks == NULL keeps every external.
 * kept (optional, [n_rec]): 1 where the recorded event took effect = the executed trace test() returns. */
int orc_sts_removal(const demi_model* m, const demi_ext_event* original_ext, uint32_t n_ext,
                    const demi_rec_event* original_rec, uint32_t n_rec, const uint64_t* mask, uint32_t skip,
                    const demi_limits* lim, demi_verdict* out, uint8_t* kept);
int orc_sts_removal_batch(const demi_model* m, const demi_ext_event* original_ext, uint32_t n_ext,
                          const demi_rec_event* original_rec, uint32_t n_rec, const uint64_t* masks,
                          const uint32_t* skip, uint64_t n, const demi_limits* lim, demi_verdict* out, int n_threads);

/* ---- K3: one DPORwHeuristics interleaving (V/schedulers/DPORwHeuristics.scala:421-942) + the racing-pair
 * analysis of dpor() (:1020-1139).  prefix: nextTrace as node keys.  trace: [DEMI_DPOR_MAX_TRACE]. */
int orc_dpor_execute(const demi_model* m, const demi_ext_event* ext, uint32_t n_ext, const uint64_t* prefix,
                     uint32_t prefix_len, uint32_t shared_len, const demi_dpor_params* par, demi_verdict* out,
                     demi_dpor_trace_entry* trace, uint32_t* trace_len, demi_dpor_pair* pairs, uint32_t* n_pairs);
int orc_dpor_trace_validate(const demi_model* m, const demi_ext_event* ev, uint32_t n, char* err, size_t err_cap);

#ifdef __cplusplus
}
#endif
#endif
