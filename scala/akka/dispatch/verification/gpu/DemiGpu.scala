package akka.dispatch.verification.gpu

/** JNI declarations, 1:1 with jni/demi_jni.c (which is 1:1 with include/demi_gpu.h).  Array layouts are documented at the
 *  top of demi_jni.c.  Every call returns 0 or a negative demi_status unless stated otherwise; lastError(h) has the text. */
object DemiGpu {
  System.loadLibrary("demi_jni")

  // DEMI_V_* verdict flags (include/demi_gpu.h)
  val V_VIOLATION = 0x1; val V_MAXMSG = 0x2; val V_PENDING_OVF = 0x4; val V_QUEUE_OVF = 0x8; val V_DIVERGED = 0x10
  val V_TRACE_OVF = 0x20; val V_PAIRS_OVF = 0x40; val V_SELFMSG = 0x80
  val MAX_PENDING = 128
  val DPOR_ORDER_ROUNDS = 0; val DPOR_ORDER_REFERENCE = 1

  @native def ctxCreate(device: Int): Long
  @native def ctxDestroy(h: Long): Unit
  @native def lastError(h: Long): String
  @native def modelLoad(h: Long, nActors: Int, msgClass: Array[Byte], actorClass: Array[Byte], nClasses: Int,
                        handlerStart: Array[Short], code: Array[Int], initState: Array[Long], inv: Array[Int]): Int
  @native def modelSpecialize(h: Long, enable: Boolean): Int
  @native def traceLoad(h: Long, events: Array[Byte]): Int
  /** demi_ext_payload_areas: the 48-bit payload areas of the external events the NEXT traceLoad / dporLoad loads (a table whose
   *  messages have more than two fields: FlatEvents.packAreas); null forgets a staged array. */
  @native def extPayloadAreas(h: Long, areas: Array[Long]): Int
  @native def randomExplore(h: Long, seedBase: Long, n: Long, limits: Array[Int], verdicts: Array[Long]): Int
  @native def randomExploreFlagged(h: Long, seedBase: Long, n: Long, limits: Array[Int], flagMask: Int,
                                   out: Array[Long], counts: Array[Long]): Int
  /** explore() in pieces, up to three calls outstanding in one context (demi_random_explore_submit / _wait; the loop: submit(k + 2), wait(k)): the ticket (> 0) or a
   *  negative status; wait fills `out` / `counts` as randomExploreFlagged does */
  @native def randomExploreSubmit(h: Long, seedBase: Long, n: Long, limits: Array[Int], flagMask: Int): Int
  @native def randomExploreWait(h: Long, ticket: Int, out: Array[Long], counts: Array[Long]): Int
  /** returns the number of recorded events (16 bytes each in `recorded`: demi_rec_event), or a negative status */
  @native def randomGetTrace(h: Long, seed: Long, limits: Array[Int], verdict: Array[Long], recorded: Array[Byte]): Int
  /** the same for execution number `execIndex` of the instance seeded `seed` (demi_limits.executions_per_instance > 1) */
  @native def randomGetTraceCarried(h: Long, seed: Long, execIndex: Int, limits: Array[Int], verdict: Array[Long], recorded: Array[Byte]): Int
  @native def replayLoad(h: Long, externals: Array[Byte], recorded: Array[Byte]): Int
  @native def replayBatch(h: Long, masks: Array[Long], limits: Array[Int], verdicts: Array[Long]): Int
  @native def replayRemovalBatch(h: Long, masksOrNull: Array[Long], skip: Array[Int], limits: Array[Int], verdicts: Array[Long]): Int
  /** RunnerUtils.stsSchedDDMin in one call (demi_ddmin): params = int[4] (depth, max_candidates, check_unmodified, verify_mcs), mcs = long[4],
   *  stats = long[5] (consultations, launches, mcs_len, verified, replays) */
  @native def ddmin(h: Long, limits: Array[Int], params: Array[Int], conjoinedOrNull: Array[Byte], mcs: Array[Long],
                    consultedOrNull: Array[Long], passedOrNull: Array[Byte], stats: Array[Long]): Int
  /** RunnerUtils.randomDDMin in one call (demi_random_ddmin) on the externals of traceLoad: params = int[6] (executions, depth, max_candidates,
   *  check_unmodified, verify_mcs, sequential), nExternals = how many externals traceLoad was given, mcs = long[4], stats = long[5]
   *  (consultations, launches, mcs_len, verified, executions run) */
  @native def randomDDMin(h: Long, seedBase: Long, limits: Array[Int], params: Array[Int], nExternals: Int, conjoinedOrNull: Array[Byte],
                          mcs: Array[Long], consultedOrNull: Array[Long], passedOrNull: Array[Byte], stats: Array[Long]): Int
  /** RandomScheduler.test for a batch of subsequences of the loaded externals (demi_random_explore_candidates): masks = long[4 * n],
   *  verdictsOrNull = long[2 * n * executions], flags = int[n] (bit 0 = some execution violates, bit 1 = some execution was aborted) */
  @native def randomExploreCandidates(h: Long, seedBase: Long, masks: Array[Long], executions: Int, limits: Array[Int],
                                      verdictsOrNull: Array[Long], flags: Array[Int]): Int
  @native def replayGetKept(h: Long, maskOrNull: Array[Long], skip: Int, limits: Array[Int], verdict: Array[Long], kept: Array[Byte]): Int
  @native def dporLoad(h: Long, externals: Array[Byte]): Int
  /** returns the length of the first violating trace (entries of 16 bytes in firstViolationTrace), or a negative status */
  /** ArvindDistanceOrdering.init(sched, originalTrace) / setInitialTrace for the following dporExplore calls (node keys; 16-byte trace entries) */
  @native def dporSetTraces(h: Long, originalKeysOrNull: Array[Long], initialTraceOrNull: Array[Byte]): Int
  @native def dporExplore(h: Long, params: Array[Int], search: Array[Int], verdicts: Array[Long], prefixLen: Array[Int],
                          rounds: Array[Int], firstViolationTrace: Array[Byte], stats: Array[Long]): Int
  /** What interleaving `index` of the last dporExplore was (demi_dpor_explored): nextTrace / trace = byte[16 * 256] (16-byte trace entries),
   *  lens = long[3] (next-trace length, its shared take() part, executed-trace length).  For diffing an exploration against DPORwHeuristics. */
  @native def dporExplored(h: Long, index: Long, nextTrace: Array[Byte], trace: Array[Byte], lens: Array[Long]): Int
  /** RunnerUtils.editDistanceDporDDMin in one call (demi_edit_distance_dpor_ddmin: IncrementalDDMin over ResumableDPOR, every DPOR
   *  consultation inside the library).  externals = 8 bytes each; initialTrace = 16-byte entries (FlatEvents.dporInitialTrace);
   *  dporParams = int[7]; params = int[7] (max_max_distance, stop_at_size, check_unmodified, ignore_quiescence, verify_mcs, batch, budget);
   *  mcs = long[4]; consultedOrNull = long[4 * cap] with passedOrNull = byte[cap], distanceOrNull = int[cap];
   *  violationTraceOrNull = byte[16 * 256]; stats = long[40] (replays, interleavings, consultations, instances, passes, mcs_len, verified,
   *  violation_len, pass_distance[16], pass_mcs_len[16]) */
  @native def editDistanceDporDDMin(h: Long, externals: Array[Byte], initialTrace: Array[Byte], dporParams: Array[Int], params: Array[Int],
                                    mcs: Array[Long], consultedOrNull: Array[Long], passedOrNull: Array[Byte], distanceOrNull: Array[Int],
                                    violationTraceOrNull: Array[Byte], stats: Array[Long]): Int
  /** ProvenanceTracker.pruneConcurrentEvents for n traces (16-byte entries, `stride` per trace); keep: 4 longs (256 bits) per trace */
  @native def provenancePrune(h: Long, traces: Array[Byte], traceLen: Array[Int], affected: Array[Int], stride: Int, keep: Array[Long]): Int
  @native def commUniqueId(id128: Array[Byte]): Int
  @native def commCreate(h: Long, id128: Array[Byte], rank: Int, world: Int): Int
  @native def commDestroy(h: Long): Int
  @native def randomExploreSharded(h: Long, seedBase: Long, nTotal: Long, limits: Array[Int], out: Array[Long], count: Array[Long]): Int

  def check(h: Long, rc: Int): Int = { if (rc < 0) throw new RuntimeException("demi_gpu error " + rc + ": " + lastError(h)); rc }
  def flags(verdicts: Array[Long], i: Int): Int = (verdicts(2 * i) & 0xFFFFFFFFL).toInt
  def fingerprint(verdicts: Array[Long], i: Int): Int = (verdicts(2 * i) >>> 32).toInt
  def hash(verdicts: Array[Long], i: Int): Long = verdicts(2 * i + 1)
}
