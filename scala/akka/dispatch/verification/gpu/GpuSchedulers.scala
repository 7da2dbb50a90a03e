package akka.dispatch.verification.gpu

import akka.actor.{ActorRef, Cell}
import akka.dispatch.Envelope
import akka.dispatch.verification._
import DemiGpu._

/** What `trait Scheduler` (schedulers/Scheduler.scala:13-104) asks of a class, for schedulers that evaluate their executions on
 *  the GPU.  The reference's drivers assign every scheduler to the Instrumenter before using it (`Instrumenter().scheduler =
 *  sched`: RunnerUtils.scala:89, 666; StatelessTestOracle, TestOracle.scala:79-83), so the GPU schedulers ARE Schedulers.  They
 *  never start an ActorSystem - an execution is simulated from the lowered table - so the Instrumenter has nothing to call
 *  back about: the notification members are no-ops, and the members through which a LIVE system would ask this scheduler what
 *  to deliver next throw (a live system under a GPU scheduler is a wiring error, not something to paper over). */
trait GpuSchedulerBase extends Scheduler {
  protected def notLive(what: String): Nothing =
    throw new IllegalStateException(getClass.getSimpleName + "." + what + ": the GPU schedulers do not drive a live ActorSystem " +
                                    "(executions are simulated from the lowered table); use the JVM scheduler for that")
  // ---- classification (Scheduler.scala:15, 27): nothing is system traffic, there is no failure detector / checkpointer here
  def isSystemCommunication(sender: ActorRef, receiver: ActorRef): Boolean = false
  def isSystemMessage(src: String, dst: String): Boolean = false
  // ---- the Instrumenter's questions to a live scheduler (Scheduler.scala:38-45)
  def start_trace(): Unit = {}                                                   // nothing to restart
  def schedule_new_message(blockedActors: Set[String]): Option[(Cell, Envelope)] = notLive("schedule_new_message")
  def next_event(): Event = notLive("next_event")
  def notify_quiescence(): Unit = {}
  // ---- notifications (Scheduler.scala:48-70)
  def before_receive(cell: Cell): Unit = {}
  def after_receive(cell: Cell): Unit = {}
  def event_produced(event: Event): Unit = {}
  def event_produced(cell: Cell, envelope: Envelope): Unit = notLive("event_produced")
  def event_consumed(event: Event): Unit = {}
  def event_consumed(cell: Cell, envelope: Envelope) { }
  def notify_timer_cancel(receiver: String, msg: Any) { }
  def enqueue_message(sender: Option[ActorRef], receiver: String, msg: Any) { notLive("enqueue_message") }
  // shutdown() is each class's own (it frees the demi_ctx)

  // ---- TestOracle's contract: "Throws an IllegalArgumentException if setInvariant has not been invoked"
  // (TestOracle.scala:45; RandomScheduler.scala:244-246).  The invariant the kernels evaluate is the DESCRIPTOR in
  // lowering.model (closures do not cross the boundary); the closure is kept so that the contract holds and so that a
  // lowering can check on sample checkpoints that the descriptor is the lowering of this very closure.
  protected var invariant: TestOracle.Invariant = null
  def setInvariant(i: TestOracle.Invariant) { invariant = i }
  protected def requireInvariant() {
    if (invariant == null) throw new IllegalArgumentException("Must invoke setInvariant before test / explore")
  }
}

/** sched.depTracker (RunnerUtils.scala:99-100 reads getGraph / getInitialTrace after a violating explore()): the reference's
 *  own DepTracker, fed from the recorded EventTrace with exactly the calls RandomScheduler makes while it runs -
 *  reportNewlyEnabled / reportNewlyEnabledExternal for every produced message (RandomScheduler.scala:291, 303),
 *  reportNewlyDelivered for every delivery (:468), reportKill / reportPartition / reportUnPartition from the orchestrator's
 *  callbacks (:127-129).  Unique ids are the DepTracker's own, as in the reference. */
object GpuDepTracker {
  def fromTrace(schedulerConfig: SchedulerConfig, trace: EventTrace, allActors: Set[String]): DepTracker = {
    val dep = new DepTracker(schedulerConfig)
    val enabled = scala.collection.mutable.HashMap[Int, Unique]()        // Uniq id of the MsgSend -> the DepTracker's node
    for (e <- trace.events) e match {
      case UniqueMsgSend(MsgSend(snd, rcv, msg), id) =>
        enabled(id) = if (EventTypes.isExternal(e)) dep.reportNewlyEnabledExternal(snd, rcv, msg) else dep.reportNewlyEnabled(snd, rcv, msg)
      case UniqueMsgEvent(_, id) => enabled.get(id).foreach(dep.reportNewlyDelivered)
      case KillEvent(n) => dep.reportKill(n, allActors, 0)
      case PartitionEvent((a, b)) => dep.reportPartition(a, b, 0)
      case UnPartitionEvent((a, b)) => dep.reportUnPartition(a, b, 0)
      case _ =>
    }
    dep
  }
}

/** RandomScheduler on the GPU: the same constructor shape as RandomScheduler (RandomScheduler.scala:41-44) plus the
 *  lowering.  Execution i of explore() is one full execution with `new FullyRandom(seed + i)`, i.e. the shape of
 *  RunnerUtils.fuzz, which builds a fresh scheduler and strategy per execution (RunnerUtils.scala:75-90).
 *
 *  One RandomScheduler instance never reseeds between its executions (RandomScheduler.scala:584, 649-651: execution k + 1
 *  continues the generator where execution k left it).  carriedGenerator = true reproduces exactly that -
 *  `new RandomScheduler(config, max_executions, ..., new FullyRandom(seed))`: one lane runs the chain of executions, the
 *  first violating one is returned - and is what a JVM run with the same seed can be compared with execution by execution
 *  (sequential by nature: use it for that comparison, not for throughput).  The default is "N independent executions with
 *  seeds seed, seed + 1, ...", the lowest violating index returned: RunnerUtils.fuzz's shape, and the parallel one. */
object GpuRandomScheduler {
  /** executions per device call of explore() (BASELINE config 2's step: 2^20 schedules) */
  val CHUNK = 1L << 20
}

class GpuRandomScheduler(val schedulerConfig: SchedulerConfig, max_executions: Int = 1, invariant_check_interval: Int = 0,
                         seed: Long = System.currentTimeMillis(), lowering: TableLowering, device: Int = 0,
                         srcDstFifo: Boolean = false, pMax: Int = 64, carriedGenerator: Boolean = false)
    extends GpuSchedulerBase with TestOracle {
  private val h = ctxCreate(device)
  if (h == 0) throw new IllegalStateException("no MI355X visible: use RandomScheduler")
  private var maxMessages = Int.MaxValue
  private var modelLoaded = false
  var stats: MinimizationStats = null
  /** filled by a violating explore(): what RunnerUtils.fuzz reads as sched.depTracker.getGraph / getInitialTrace (:99-100) */
  var depTracker = new DepTracker(schedulerConfig)
  def getName = "GpuRandomScheduler"
  def setMaxMessages(m: Int) { maxMessages = m }

  private def limits(lookingFor: Option[ViolationFingerprint], p: Int = pMax) = Array(
    if (maxMessages == Int.MaxValue) 0 else maxMessages, math.max(0, invariant_check_interval), p,
    if (lookingFor.isDefined) 1 else 0, lookingFor.map(lowering.fingerprintCode).getOrElse(0), 0, if (srcDstFifo) 1 else 0, 0,
    // demi_limits.executions_per_instance: carriedGenerator = exactly `new RandomScheduler(config, max_executions)` with
    // `new FullyRandom(seed)` - ONE generator through all executions, the first violating one returned
    if (carriedGenerator) max_executions else 1)

  private def prepare(trace: Seq[ExternalEvent]) {
    if (!modelLoaded) {
      val m = lowering.model
      check(h, modelLoad(h, m.nActors, m.msgClass, m.actorClass, m.nClasses, m.handlerStart, m.code, m.initState,
                         Array(m.invKind, m.invFa, m.invVa, m.invFb, m.fpMatchMask, m.flags)))
      if (m.compiledOnly) check(h, modelSpecialize(h, true))     // a wide table, or one with arrays, runs only as compiled code
      else if (max_executions >= (1 << 16)) modelSpecialize(h, true)   // optional: a failure keeps the table interpreter
      modelLoaded = true
    }
    if (lowering.model.payloads > 2) check(h, extPayloadAreas(h, FlatEvents.packAreas(trace, lowering)))   // external Sends with all their fields
    check(h, traceLoad(h, FlatEvents.pack(trace, lowering)))
  }

  /** explore (RandomScheduler.scala:234-272): Some((trace, fingerprint)) of the lowest violating execution.  Executions
   *  that were aborted on an engine capacity are re-run alone with the largest pending set before a higher index is
   *  believed; if one still does not fit, the JVM scheduler has to decide it (UnsupportedOnGpu). */
  def explore(trace: Seq[ExternalEvent], lookingFor: Option[ViolationFingerprint] = None)
      : Option[(EventTrace, ViolationFingerprint)] = {
    requireInvariant()
    prepare(trace)
    if (stats != null) (1 to max_executions).foreach(_ => stats.increment_replays())
    val OVF = V_PENDING_OVF | V_QUEUE_OVF
    var start = 0L
    // Carried generator: the executions of the instance are ONE chain.  An execution aborted on a capacity leaves the generator
    // where the reference's instance never is, so every verdict behind it is meaningless (the kernel runs the chain on; the
    // Python mirror raises CapacityExceeded): the whole chain - exploration AND the recording re-run - is then repeated with the
    // largest pending set, and if an execution still does not fit the JVM scheduler has to decide the instance.
    var chainPMax = pMax
    // Fresh generator per execution (the shape of RunnerUtils.fuzz): the executions go to the device in calls of CHUNK; while the
    // answer of call k is waited for, calls k + 1 and k + 2 are already submitted (the library overlaps the tail of every launch
    // with the start of the next), and nothing beyond the calls in flight behind the first violating one is ever submitted.
    val pipelined = !carriedGenerator
    val ahead = new scala.collection.mutable.Queue[(Long, Long, Int)]   // submitted calls, in order: (first execution, executions, ticket)
    def topUp(from: Long) {
      var nxt = if (ahead.isEmpty) from else ahead.last._1 + ahead.last._2
      while (ahead.size < 3 && nxt < max_executions) {
        val n = math.min(GpuRandomScheduler.CHUNK, max_executions - nxt)
        ahead.enqueue((nxt, n, check(h, randomExploreSubmit(h, seed + nxt, n, limits(lookingFor, pMax), V_VIOLATION | OVF))))
        nxt += n
      }
    }
    def drain() {
      while (ahead.nonEmpty) randomExploreWait(h, ahead.dequeue()._3, new Array[Long](0), new Array[Long](2))
    }
    try { while (start < max_executions) {
      val out = new Array[Long](2 * 65536); val counts = new Array[Long](2)
      var span = max_executions - start                    // executions [start, start + span) are decided by this pass
      if (pipelined) {
        if (ahead.nonEmpty && ahead.head._1 != start) drain()   // (a truncated list sent the search back into the middle of a call)
        topUp(start)
        val cur = ahead.dequeue()
        span = cur._2
        check(h, randomExploreWait(h, cur._3, out, counts))
      } else
        check(h, randomExploreFlagged(h, seed + start, max_executions - start, limits(lookingFor, chainPMax), V_VIOLATION | OVF, out, counts))
      if (counts(0) == 0 && !pipelined) return None
      // (index, flags) in index order; a truncated list is an arbitrary subset: then only the lowest index is certain
      val cand0 = if (counts(0) <= 65536) (0 until counts(0).toInt).map(i => (out(2 * i), ((out(2 * i + 1) >>> 32) & 0xFF).toInt))
                  else Seq((counts(1), -1))
      // (carried: only what precedes the chain's first violating execution was run at all; an OVF there poisons the rest)
      val chainOvf = carriedGenerator && cand0.exists { case (_, fl) => fl < 0 || (fl & OVF) != 0 }
      if (chainOvf && chainPMax == MAX_PENDING)
        throw new UnsupportedOnGpu("an execution of the carried-generator instance exceeds the engine's capacities")
      val cand = if (chainOvf) { chainPMax = MAX_PENDING; Seq.empty[(Long, Int)] } else cand0
      var next = if (chainOvf) 0L else start + span                   // (0: the same chain again, with the largest pending set)
      for ((idx, fl) <- cand) {
        var lim = limits(lookingFor, chainPMax)
        if (fl < 0 || (fl & OVF) != 0) lim = limits(lookingFor, MAX_PENDING)
        val v = new Array[Long](2); val rec = new Array[Byte](FlatEvents.REC_BYTES * 16384)
        // (carried generator: `idx` is the execution number within the one instance; the recording kernel re-runs the chain)
        val n = if (carriedGenerator) check(h, randomGetTraceCarried(h, seed, (start + idx).toInt, lim, v, rec))
                else check(h, randomGetTrace(h, seed + start + idx, lim, v, rec))
        val f = flags(v, 0)
        if ((f & OVF) != 0) throw new UnsupportedOnGpu("schedule " + (start + idx) + " exceeds the engine's capacities")
        if ((f & V_VIOLATION) != 0) {
          val used = trace.take((f >> 8) & 0xFF)        // checkIfBugFound prunes the externals never injected (:160-163)
          val found = FlatEvents.toEventTrace(rec, n, used, lowering)
          depTracker = GpuDepTracker.fromTrace(schedulerConfig, found, (0 until lowering.model.nActors).map(lowering.actorName).toSet)
          return Some((found, lowering.fingerprintOf(fingerprint(v, 0))))
        }
        if (fl < 0) next = start + idx + 1
      }
      if (counts(0) <= 65536 && !chainOvf && start + span >= max_executions) return None
      start = next
    } } finally drain()
    None
  }

  /** TestOracle.test (RandomScheduler.scala:597-612). */
  def test(events: Seq[ExternalEvent], fp: ViolationFingerprint, _stats: MinimizationStats,
           init: Option[() => Any] = None): Option[EventTrace] = {
    stats = _stats
    explore(events, Some(fp)).map(_._1)
  }
  def shutdown() { ctxDestroy(h) }
}

/** STSScheduler(schedulerConfig, original_trace, allowPeek = false) as DDMin's oracle (STSScheduler.scala:199-310). */
class GpuSTSScheduler(val schedulerConfig: SchedulerConfig, original_trace: EventTrace, lowering: TableLowering,
                      device: Int = 0, pMax: Int = 64) extends GpuSchedulerBase with TestOracle {
  private val h = ctxCreate(device)
  if (h == 0) throw new IllegalStateException("no MI355X visible: use STSScheduler")
  // what RunnerUtils.stsSchedDDMin sets on its scheduler before minimising (RunnerUtils.scala:655-667): the callbacks run around
  // every test() as in STSScheduler.test (:206-214, 300-305); the actor Props are what populateActorSystem would create - the
  // lowering already names every actor, so they are only remembered
  private var preTest: Option[STSScheduler.PreTestCallback] = None
  private var postTest: Option[STSScheduler.PostTestCallback] = None
  def setPreTestCallback(c: STSScheduler.PreTestCallback) { preTest = Some(c) }
  def setPostTestCallback(c: STSScheduler.PostTestCallback) { postTest = Some(c) }
  var actorNamePropPairs: Seq[Tuple2[akka.actor.Props, String]] = Seq.empty
  def setActorNamePropPairs(pairs: Seq[Tuple2[akka.actor.Props, String]]) { actorNamePropPairs = pairs }
  private val externals = original_trace.original_externals
  private val indexOf = externals.zipWithIndex.map { case (e, i) => e._id -> i }.toMap      // ExternalEvent._id (ExternalEvents.scala:14-31)
  private val recorded = FlatEvents.packRecorded(original_trace, lowering)
  locally {
    val m = lowering.model
    check(h, modelLoad(h, m.nActors, m.msgClass, m.actorClass, m.nClasses, m.handlerStart, m.code, m.initState,
                       Array(m.invKind, m.invFa, m.invVa, m.invFb, m.fpMatchMask, m.flags)))
    if (m.compiledOnly) check(h, modelSpecialize(h, true))
    check(h, replayLoad(h, FlatEvents.pack(externals, lowering), recorded))
  }
  def getName = "GpuSTSSchedNoPeek"
  // the last entry is demi_limits.filter_known_absents: SchedulerConfig.filterKnownAbsents, as the reference computes it
  private def limits(fp: ViolationFingerprint, p: Int = pMax) =
    Array(0, 0, p, 1, lowering.fingerprintCode(fp), 0, 0, if (schedulerConfig.filterKnownAbsents) 1 else 0, 1)
  private def mask(subseq: Seq[ExternalEvent]): Array[Long] = {
    val m = new Array[Long](4)
    for (e <- subseq) { val i = indexOf(e._id); m(i >> 6) |= 1L << (i & 63) }
    m
  }

  /** one launch for a whole DDMin frontier: element i is Some(executed trace) iff subseqs(i) reproduces the violation */
  def testBatch(subseqs: Seq[Seq[ExternalEvent]], fp: ViolationFingerprint, stats: MinimizationStats): Seq[Option[EventTrace]] = {
    requireInvariant()
    if (stats != null) subseqs.foreach(_ => stats.increment_replays())
    val masks = subseqs.flatMap(mask).toArray
    val v = new Array[Long](2 * subseqs.size)
    check(h, replayBatch(h, masks, limits(fp), v))
    val OVF = V_PENDING_OVF | V_QUEUE_OVF
    subseqs.indices.map { i =>
      var f = flags(v, i)
      if ((f & OVF) != 0) {                                 // aborted on a capacity: once more with the largest pending set
        val v1 = new Array[Long](2)
        check(h, replayBatch(h, mask(subseqs(i)), limits(fp, MAX_PENDING), v1))
        f = flags(v1, 0)
        if ((f & OVF) != 0) throw new UnsupportedOnGpu("a candidate replay exceeds the engine's capacities")
      }
      if ((f & V_VIOLATION) != 0) Some(executed(subseqs(i), fp)) else None
    }
  }

  /** RunnerUtils.stsSchedDDMin (:642-707) in ONE native call (demi_ddmin): DDMin.minimize / ddmin2 over the loaded execution's
   *  external events with this scheduler as the oracle - the atoms, the splits, the speculative frontiers and their launches all
   *  inside the library.  Returns (the MCS as a subsequence of the original externals, the trace as stsSchedDDMin returns it);
   *  the consultations are added to `stats` as DDMin would.  maxCandidates = 0 leaves the launch width to the library.  Same MCS
   *  as `new DDMin(this).minimize(...)` (the same decision tree, consulted through cached verdicts).
   *  conjoined: UnmodifiedEventDag.conjoinAtoms as one partner index per external event (-1 / 255 = none), or None.
   *  As in the reference (:689-706): an MCS shorter than the externals is verified - Some(executed trace) if it reproduces the
   *  violation, None otherwise - and an MCS that removed nothing returns Some(original trace) without a verification.
   *  Where this differs from the reference's loop, by construction: the preTest / postTest callbacks stsSchedDDMin installs run
   *  ONCE around the whole native search, not around each of its consultations (those happen inside one library call; a
   *  callback that must see every replay needs `new DDMin(this)` over test()). */
  def ddmin(fp: ViolationFingerprint, stats: MinimizationStats, checkUnmodified: Boolean = true,
            maxCandidates: Int = 0, conjoined: Option[Array[Byte]] = None): (Seq[ExternalEvent], Option[EventTrace]) = {
    requireInvariant()
    val mcs = new Array[Long](4); val st = new Array[Long](5)
    val ext = original_trace.original_externals
    conjoined.foreach(c => require(c.length >= ext.size, "one conjoined-partner index per external event"))
    preTest.foreach(_())
    try check(h, DemiGpu.ddmin(h, limits(fp), Array(0, maxCandidates, if (checkUnmodified) 1 else 0, 1), conjoined.orNull, mcs, null, null, st))
    finally postTest.foreach(_())
    if (stats != null) (0L until st(0)).foreach(_ => stats.increment_replays())
    val kept = ext.indices.filter(i => ((mcs(i >> 6) >>> (i & 63)) & 1L) != 0).map(ext)
    val externalsSize = ext.count { case WaitQuiescence() => false; case WaitCondition(_) => false; case _ => true }
    if (kept.length < externalsSize) (kept, if (st(3) != 0) Some(executed(kept, fp)) else None)
    else (kept, Some(original_trace))
  }

  /** the EventTrace test() returns on success (:286-292): the recorded events that took effect in the replay */
  private def executed(subseq: Seq[ExternalEvent], fp: ViolationFingerprint): EventTrace = {
    val v = new Array[Long](2); val kept = new Array[Byte](original_trace.events.size)
    check(h, replayGetKept(h, mask(subseq), -1, limits(fp, MAX_PENDING), v, kept))
    val t = new EventTrace(subseq)
    for ((e, k) <- original_trace.events.zip(kept) if k != 0) t += e
    t
  }

  def test(subseq: Seq[ExternalEvent], fp: ViolationFingerprint, stats: MinimizationStats,
           init: Option[() => Any] = None): Option[EventTrace] = {
    preTest.foreach(_())
    val r = testBatch(Seq(subseq), fp, stats).head
    postTest.foreach(_())
    r
  }
  def shutdown() { ctxDestroy(h) }
}

/** RunnerUtils.testWithStsSched (RunnerUtils.scala:913-943) for STSSchedMinimizer: the candidates are "the last failing
 *  trace minus the delivery at index skip(i)" (OneAtATimeRemoval.scala:57-124). */
class GpuStsRemovalOracle(schedulerConfig: SchedulerConfig, mcs: Seq[ExternalEvent], lowering: TableLowering,
                          device: Int = 0, pMax: Int = 64) {
  private val h = ctxCreate(device)
  private var loaded: EventTrace = null
  private var modelLoaded = false
  // the last entry is demi_limits.filter_known_absents: SchedulerConfig.filterKnownAbsents, as the reference computes it
  private def limits(fp: ViolationFingerprint, p: Int = pMax) =
    Array(0, 0, p, 1, lowering.fingerprintCode(fp), 0, 0, if (schedulerConfig.filterKnownAbsents) 1 else 0, 1)
  private def load(trace: EventTrace) {
    if (!modelLoaded) {
      val m = lowering.model
      check(h, modelLoad(h, m.nActors, m.msgClass, m.actorClass, m.nClasses, m.handlerStart, m.code, m.initState,
                         Array(m.invKind, m.invFa, m.invVa, m.invFb, m.fpMatchMask, m.flags)))
      if (m.compiledOnly) check(h, modelSpecialize(h, true))
      modelLoaded = true
    }
    if (!(loaded eq trace)) { check(h, replayLoad(h, FlatEvents.pack(mcs, lowering), FlatEvents.packRecorded(trace, lowering))); loaded = trace }
  }
  /** one launch for the whole proposal sequence of a RemovalStrategy (each proposal assumes the previous one failed) */
  def testRemovals(trace: EventTrace, skip: Array[Int], fp: ViolationFingerprint): Array[Boolean] = {
    load(trace)
    val out = new Array[Long](2 * skip.length)
    check(h, replayRemovalBatch(h, null, skip, limits(fp, MAX_PENDING), out))
    Array.tabulate(skip.length) { i =>
      if ((flags(out, i) & (V_PENDING_OVF | V_QUEUE_OVF)) != 0) throw new UnsupportedOnGpu("a removal candidate exceeds the engine's capacities")
      (flags(out, i) & V_VIOLATION) != 0
    }
  }
  /** Some(executed trace) iff the candidate still triggers the violation */
  def executed(trace: EventTrace, skip: Int, fp: ViolationFingerprint): Option[EventTrace] = {
    load(trace)
    val v = new Array[Long](2); val kept = new Array[Byte](trace.events.size)
    check(h, replayGetKept(h, null, skip, limits(fp, MAX_PENDING), v, kept))
    if ((flags(v, 0) & V_VIOLATION) == 0) None
    else { val t = new EventTrace(trace.original_externals); for ((e, k) <- trace.events.zip(kept) if k != 0) t += e; Some(t) }
  }
  def shutdown() { ctxDestroy(h) }
}

/** RunnerUtils.boundedDPOR (RunnerUtils.scala:881-911) / DPORwHeuristics.test over the whole exploration in the library:
 *  backtrack queue, ExploredTacker and getNext() run natively, the explored set and the traces stay on the GPU.
 *  referenceOrder = true commits the interleavings in DPORwHeuristics' own one-at-a-time order (same sequence, same
 *  "first violation found"); false explores in rounds of `batch` (a slightly different explored set, much faster). */
class GpuDPOR(val schedulerConfig: SchedulerConfig, lowering: TableLowering, depthBound: Int = 0, batch: Int = 4096,
              stopIfViolationFound: Boolean = true, referenceOrder: Boolean = true, maxInterleavings: Int = 1 << 17,
              device: Int = 0) extends GpuSchedulerBase with TestOracle {
  private val h = ctxCreate(device)
  if (h == 0) throw new IllegalStateException("no MI355X visible: use DPORwHeuristics")
  def getName = "GpuDPORwHeuristics"
  // what RunnerUtils.boundedDPOR sets before test() (RunnerUtils.scala:889-900)
  private var maxMessagesToSchedule = 0
  def setMaxMessagesToSchedule(m: Int) { maxMessagesToSchedule = m }                       // DPORwHeuristics.scala:136-139
  var actorNameProps: Seq[Tuple2[akka.actor.Props, String]] = Seq.empty
  def setActorNameProps(pairs: Seq[Tuple2[akka.actor.Props, String]]) { actorNameProps = pairs }
  def test(events: Seq[ExternalEvent], fp: ViolationFingerprint, stats: MinimizationStats,
           init: Option[() => Any] = None): Option[EventTrace] = {
    requireInvariant()
    val m = lowering.model
    check(h, modelLoad(h, m.nActors, m.msgClass, m.actorClass, m.nClasses, m.handlerStart, m.code, m.initState,
                       Array(m.invKind, m.invFa, m.invVa, m.invFb, m.fpMatchMask, m.flags)))
    if (m.compiledOnly) check(h, modelSpecialize(h, true))
    else modelSpecialize(h, true)
    if (lowering.model.payloads > 2) check(h, extPayloadAreas(h, FlatEvents.packAreas(events, lowering)))
    check(h, dporLoad(h, FlatEvents.pack(events, lowering)))      // Start / Send / WaitQuiescence only (DPORwHeuristics.scala:692-710)
    val params = Array(depthBound, maxMessagesToSchedule, 1, lowering.fingerprintCode(fp), 64, 4096, 0)
    val search = Array(batch, maxInterleavings, if (stopIfViolationFound) 1 else 0, 1,
                       if (referenceOrder) DPOR_ORDER_REFERENCE else DPOR_ORDER_ROUNDS, 0,
                       0 /* DefaultBacktrackOrdering; 1 = ArvindDistanceOrdering after dporSetTraces */, 0 /* no distance cap */,
                       0 /* resume: 1 = continue from the queue the previous ordered exploration of this context left */)
    val verdicts = new Array[Long](2 * maxInterleavings); val plen = new Array[Int](maxInterleavings)
    val rounds = new Array[Int](maxInterleavings); val vt = new Array[Byte](16 * 256); val st = new Array[Long](13)
    val vlen = check(h, dporExplore(h, params, search, verdicts, plen, rounds, vt, st))
    if (stats != null) (0L until st(0)).foreach(_ => stats.increment_replays())
    if (st(2) == 0) None
    else Some(GpuDPOR.traceOf(vt, vlen, events, lowering))
  }
  def shutdown() { ctxDestroy(h) }
}

/** RunnerUtils.editDistanceDporDDMin (RunnerUtils.scala:810-879) in ONE native call (demi_edit_distance_dpor_ddmin): IncrementalDDMin
 *  (IncrementalDeltaDebugging.scala:20-92) over ResumableDPOR (:94-122) - every consultation a bounded DPORwHeuristics exploration with
 *  ArvindDistanceOrdering, the original execution as initial trace and a distance cap that doubles per pass; a subsequence consulted
 *  again continues from its backtrack queue.  Returns what the reference returns: (mcs externals, stats, the reproducing trace if the
 *  MCS is smaller than the externals and verifies, the violation).  The preTest / postTest hooks of the reference's schedulers have no
 *  counterpart: the consultations happen inside one library call. */
object GpuEditDistanceDporDDMin {
  def apply(schedulerConfig: SchedulerConfig, lowering: TableLowering, trace: EventTrace, violation: ViolationFingerprint,
            ignoreQuiescence: Boolean = true, stats: Option[MinimizationStats] = None, stopAtSize: Int = 6, maxMaxDistance: Int = 8,
            batch: Int = 256, device: Int = 0): (Seq[ExternalEvent], MinimizationStats, Option[EventTrace], ViolationFingerprint) = {
    if (schedulerConfig.invariant_check.isEmpty) throw new IllegalArgumentException("Must invoke setInvariant before test()")
    val h = ctxCreate(device)
    if (h == 0) throw new IllegalStateException("no MI355X visible: use RunnerUtils.editDistanceDporDDMin")
    try {
      val m = lowering.model
      check(h, modelLoad(h, m.nActors, m.msgClass, m.actorClass, m.nClasses, m.handlerStart, m.code, m.initState,
                         Array(m.invKind, m.invFa, m.invVa, m.invFb, m.fpMatchMask, m.flags)))
      if (m.compiledOnly) check(h, modelSpecialize(h, true))     // a wide table, or one with arrays, runs only as compiled code
      else modelSpecialize(h, true)                              // optional: a failure keeps the table interpreter
      val ext = trace.original_externals
      val init = FlatEvents.dporInitialTrace(trace, lowering)
      // what the reference's dporConstructor sets (:827-839): prioritizePendingUponDivergence, setMaxMessagesToSchedule(initialTrace.size)
      val dporParams = Array(0, init.length / 16, 1, lowering.fingerprintCode(violation), 64, 4096, 1)
      val params = Array(maxMaxDistance, stopAtSize, 0, if (ignoreQuiescence) 1 else 0, 1, batch, 0)
      val mcs = new Array[Long](4); val vt = new Array[Byte](16 * 256); val st = new Array[Long](40)
      check(h, DemiGpu.editDistanceDporDDMin(h, FlatEvents.pack(ext, lowering), init, dporParams, params, mcs, null, null, null, vt, st))
      val out = stats.getOrElse(new MinimizationStats)
      (0L until st(0)).foreach(_ => out.increment_replays())
      val kept = ext.indices.filter(i => ((mcs(i >> 6) >>> (i & 63)) & 1L) != 0).map(ext)
      // RunnerUtils.scala:841-878: an MCS no smaller than the externals that took part is not verified at all and the ORIGINAL trace
      // is returned (st(6) == -1); otherwise Some(reproducing trace) when the MCS verifies (1), None when it does not (0)
      val verified = st(6) match {
        case 1L => Some(GpuDPOR.traceOf(vt, st(7).toInt, kept, lowering))
        case 0L => None
        case _  => Some(trace)
      }
      (kept, out, verified, violation)
    } finally ctxDestroy(h)
  }
}

/** RunnerUtils.randomDDMin (RunnerUtils.scala:601-623) in ONE native call (demi_random_ddmin): DDMin whose TestOracle is the RandomScheduler
 *  itself - a candidate subsequence "fails" iff one of `maxExecutions` random interleavings of it reproduces `violation` (the reference
 *  builds its scheduler with max_executions = 1; SURVEY 8d's config 4 asks for 100).  The decision tree, its speculative frontier and the
 *  launches - (frontier candidates x maxExecutions) executions each, a workgroup per candidate - run inside the library.  Returns what the
 *  reference returns: (mcs externals, stats, Some(a reproducing trace) if the MCS verifies - the original trace when nothing was removed -,
 *  the violation).  `sched.setMaxMessages(trace.size)` as in the reference (:608). */
object GpuRandomDDMin {
  def apply(schedulerConfig: SchedulerConfig, lowering: TableLowering, trace: EventTrace, violation: ViolationFingerprint,
            maxExecutions: Int = 100, seed: Long = 0L, stats: Option[MinimizationStats] = None, device: Int = 0, pMax: Int = 64,
            maxCandidates: Int = 256): (Seq[ExternalEvent], MinimizationStats, Option[EventTrace], ViolationFingerprint) = {
    if (schedulerConfig.invariant_check.isEmpty) throw new IllegalArgumentException("Must invoke setInvariant before test()")
    val h = ctxCreate(device)
    if (h == 0) throw new IllegalStateException("no MI355X visible: use RunnerUtils.randomDDMin")
    try {
      val m = lowering.model
      check(h, modelLoad(h, m.nActors, m.msgClass, m.actorClass, m.nClasses, m.handlerStart, m.code, m.initState,
                         Array(m.invKind, m.invFa, m.invVa, m.invFb, m.fpMatchMask, m.flags)))
      if (m.compiledOnly) check(h, modelSpecialize(h, true)) else modelSpecialize(h, true)
      val ext = trace.original_externals
      if (lowering.model.payloads > 2) check(h, extPayloadAreas(h, FlatEvents.packAreas(ext, lowering)))
      check(h, traceLoad(h, FlatEvents.pack(ext, lowering)))
      // demi_limits: max_messages = trace.size (:608), no periodic invariant check (RandomScheduler(config, 1, 0)), lookingFor = violation
      val limits = Array(trace.events.size, 0, pMax, 1, lowering.fingerprintCode(violation), 0, 0, 0, 1)
      val params = Array(maxExecutions, 0, maxCandidates, 0 /* checkUnmodifed = false (:609) */, 1, 0)
      val mcs = new Array[Long](4); val st = new Array[Long](5)
      check(h, DemiGpu.randomDDMin(h, seed, limits, params, ext.size, null, mcs, null, null, st))
      val out = stats.getOrElse(new MinimizationStats)
      (0L until st(0) * maxExecutions).foreach(_ => out.increment_replays())      // RandomScheduler counts a replay per execution (:203-205)
      val kept = ext.indices.filter(i => ((mcs(i >> 6) >>> (i & 63)) & 1L) != 0).map(ext)
      // (:611-622) the MCS is validated only when it is smaller than the externals; the reproducing trace itself comes from the
      // recording kernel: the lowest violating execution of the MCS, as RandomScheduler.test returns it
      val verified: Option[EventTrace] =
        if (kept.size >= ext.size) Some(trace)
        else if (st(3) != 1L) None
        else {
          val sched = new GpuRandomScheduler(schedulerConfig, maxExecutions, 0, seed, lowering, device, false, pMax)
          try { sched.setInvariant(schedulerConfig.invariant_check.get); sched.setMaxMessages(trace.events.size); sched.test(kept, violation, out) }
          finally sched.shutdown()
        }
      (kept, out, verified, violation)
    } finally ctxDestroy(h)
  }
}

object GpuDPOR {
  /** demi_dpor_trace_entry[] (key 8, word 4, parent, qperiod, depth, kind) -> the MsgEvents of the violating interleaving */
  def traceOf(vt: Array[Byte], n: Int, externals: Seq[ExternalEvent], lo: TableLowering): EventTrace = {
    val t = new EventTrace(externals)
    for (i <- 0 until n) {
      val o = 16 * i
      val kind = vt(o + 15) & 0xFF
      if (kind == 1) {
        val w = (vt(o + 8) & 0xFF) | ((vt(o + 9) & 0xFF) << 8) | ((vt(o + 10) & 0xFF) << 16) | ((vt(o + 11) & 0xFF) << 24)
        // (the header in the layout of the lowering's table: 3 + 4 bits of receiver / sender, or 4 + 5 for more than 8 actors; a
        // wide table's entry reports type, receiver, sender and p0 - p1 is not in the low half of its word)
        val (ty, dst, src) = lo.model.header(w)
        val (p0, p1) = if (lo.model.wide) ((w >>> 16) & 0xFFFF, 0) else ((w >> 16) & 255, (w >>> 24) & 255)
        t += MsgEvent(if (src == lo.model.deadLetters) "deadLetters" else lo.actorName(src), lo.actorName(dst), lo.decode(ty, p0, p1))
      } else if (kind == 2) t += Quiescence
    }
    t
  }
}
