package akka.dispatch.verification.gpu

import akka.dispatch.verification._

/** The flat arrays of demi_model (include/demi_gpu.h): the application's actors as a transition table. */
case class FlatModel(nActors: Int, msgClass: Array[Byte], actorClass: Array[Byte], nClasses: Int,
                     handlerStart: Array[Short], code: Array[Int], initState: Array[Long],
                     invKind: Int, invFa: Int, invVa: Int, invFb: Int, fpMatchMask: Int = 0xFFFFFFFF,
                     wide: Boolean = false, arrayLen: Int = 0, payloads: Int = 2) {
  /** demi_model.flags.  wide = DEMI_MODEL_WIDE: 16-bit state fields and payloads (terms / log indices above 255); initState then
   *  holds two words per actor (F0..F3, F4..F7).  arrayLen = DEMI_MODEL_ARRAY(n): every actor owns an array of n elements
   *  beside its eight fields (rows LDX / STX: a replicated log, a vote table), empty at the start (include/demi_gpu.h). */
  def flags: Int = (if (wide) 1 else 0) | ((arrayLen & 0xFF) << 8) | (if (payloads > 2) payloads << 16 else 0)
  /** payloads = DEMI_MODEL_PAYLOADS(n), n = 3..6 (wide tables only): a message carries n payload fields, each payloadBits wide
   *  (16, 12, 9, 8 bits), in the 48-bit payload area of the 64-bit message word; demi_rec_event holds the area as
   *  p0 | p1 << 16 | p_hi << 32.  A table without the option has two fields: 8 bits each, 16 when wide. */
  def payloadBits: Int = if (!wide) 8 else if (payloads <= 3) 16 else 48 / payloads
  def area(fields: Seq[Int]): Long = {
    val m = (1L << payloadBits) - 1
    if (!wide) (fields(0) & m) | ((fields(1) & m) << 16)
    else fields.take(payloads).zipWithIndex.map { case (v, k) => (v.toLong & m) << (k * payloadBits) }.foldLeft(0L)(_ | _)
  }
  def fieldsOf(area: Long): Seq[Int] = {
    val m = (1L << payloadBits) - 1
    if (!wide) Seq((area & m).toInt, ((area >> 16) & m).toInt) else (0 until payloads).map(k => ((area >> (k * payloadBits)) & m).toInt)
  }
  /** More than 8 actors (9 .. 16): the BIG layout of include/demi_gpu.h - such a table must be wide; its message word has a 4-bit
   *  receiver and a 5-bit sender field, its deadLetters id is 31, its fingerprints carry 16-bit actor masks. */
  def big: Boolean = nActors > 8
  def deadLetters: Int = if (big) 31 else 15
  /** the message word (include/demi_gpu.h): 32 bits, or the 64 bits of a wide table */
  def word(msgType: Int, src: Int, dst: Int, area: Long): Long =
    if (!wide) (msgType | (dst << 5) | (src << 8)).toLong | ((area & 0xFF) << 16) | (((area >> 16) & 0xFF) << 24)
    else (msgType | (dst << 5) | (src << (if (big) 9 else 8))).toLong | (area << 16)
  /** (type, receiver, sender) of the low half of a message word (demi_dpor_trace_entry.word) */
  def header(w: Int): (Int, Int, Int) = if (big) (w & 31, (w >> 5) & 15, (w >> 9) & 31) else (w & 31, (w >> 5) & 7, (w >> 8) & 15)
  /** Such a table has no interpreter on the device: every scheduler compiles it right after loading it. */
  def compiledOnly: Boolean = wide || arrayLen > 0
}

/** What an application supplies next to its MessageFingerprinter (MessageFingerprints.scala:14-32): how its actors,
 *  messages and invariant lower to the table.  The row vocabulary is DEMI_OP_* in include/demi_gpu.h; demi_amd/model.py
 *  (raft_model) is a complete example in the Python mirror.  Closures cannot cross the boundary (Props, message
 *  constructors, Invariant: SURVEY 8b), which is exactly what this trait replaces. */
trait TableLowering {
  def model: FlatModel
  def actorId(name: String): Int                      // 0 .. nActors - 1
  def actorName(id: Int): String
  def encode(msg: Any): (Int, Int, Int)               // (msg_type, p0, p1) = the message's fingerprint
  def decode(msgType: Int, p0: Int, p1: Int): Any     // a message equal to the original under the app's fingerprinter
  /** Messages with more than two fields (model.payloads > 2): every field, P0 first.  The defaults serve two-field tables. */
  def encodeFields(msg: Any): (Int, Seq[Int]) = { val (t, p0, p1) = encode(msg); (t, Seq(p0, p1)) }
  def decodeFields(msgType: Int, fields: Seq[Int]): Any = decode(msgType, fields(0), fields(1))
  def fingerprintCode(fp: ViolationFingerprint): Int  // 32-bit code of demi_verdict.fingerprint
  def fingerprintOf(code: Int): ViolationFingerprint
}

class UnsupportedOnGpu(what: String) extends RuntimeException(what + " cannot run on the GPU path: use the JVM scheduler")

/** ExternalEvent <-> demi_ext_event (8 bytes) and demi_rec_event (16 bytes) <-> the EventTrace records. */
object FlatEvents {
  val EV_START = 0; val EV_KILL = 1; val EV_SEND = 2; val EV_PARTITION = 3; val EV_UNPARTITION = 4; val EV_WAIT_QUIESCENCE = 5
  val REC_SPAWN = 0; val REC_KILL = 1; val REC_PARTITION = 2; val REC_UNPARTITION = 3; val REC_BEGIN_WAIT_QUIESCENCE = 4
  val REC_QUIESCENCE = 5; val REC_MSG_SEND = 6; val REC_MSG_EVENT = 7
  val REC_BYTES = 16           // sizeof(demi_rec_event)
  val DEADLETTERS = 15         // (tables of up to 8 actors; FlatModel.deadLetters is the id of the lowering's own layout)

  def pack(trace: Seq[ExternalEvent], lo: TableLowering): Array[Byte] = {
    val out = new Array[Byte](8 * trace.size)
    for ((ev, i) <- trace.zipWithIndex) {
      val (kind, a, b, t, p0, p1) = ev match {
        case Start(_, name) => (EV_START, lo.actorId(name), 0, 0, 0, 0)
        case Kill(name) => (EV_KILL, lo.actorId(name), 0, 0, 0, 0)
        case Send(name, ctor) => val (t, p0, p1) = lo.encode(ctor()); (EV_SEND, lo.actorId(name), 0, t, p0, p1)
        case Partition(x, y) => (EV_PARTITION, lo.actorId(x), lo.actorId(y), 0, 0, 0)
        case UnPartition(x, y) => (EV_UNPARTITION, lo.actorId(x), lo.actorId(y), 0, 0, 0)
        case WaitQuiescence() => (EV_WAIT_QUIESCENCE, 0, 0, 0, 0, 0)
        case other => throw new UnsupportedOnGpu(other.getClass.getSimpleName)   // WaitCondition, CodeBlock, HardKill
      }
      val o = 8 * i
      out(o) = kind.toByte; out(o + 1) = a.toByte; out(o + 2) = b.toByte; out(o + 3) = t.toByte
      out(o + 4) = p0.toByte; out(o + 5) = p1.toByte
      out(o + 6) = (p0 >> 8).toByte; out(o + 7) = (p1 >> 8).toByte       // 16-bit payloads: wide models only (else 0)
    }
    out
  }

  /** The payload areas of the trace's Sends (demi_ext_payload_areas), for a lowering whose messages have more than two fields:
   *  every field of the constructor's message, packed as FlatModel.area does; 0 for the events that are not Sends. */
  def packAreas(trace: Seq[ExternalEvent], lo: TableLowering): Array[Long] =
    trace.map { case Send(_, ctor) => lo.model.area(lo.encodeFields(ctor())._2); case _ => 0L }.toArray

  private def name(id: Int, lo: TableLowering) = if (id == lo.model.deadLetters) "deadLetters" else lo.actorName(id)

  /** demi_rec_event[] -> the records an EventTrace holds (EventTrace.scala:16-18, AuxilaryTypes.scala:34-69). */
  def toEventTrace(rec: Array[Byte], nRec: Int, externals: Seq[ExternalEvent], lo: TableLowering): EventTrace = {
    val trace = new EventTrace(externals)
    for (i <- 0 until nRec) {
      // kind, snd, rcv, msg_type, p0 (u16), p1 (u16), flags, ext_idx, p_hi (u16), id (u32): include/demi_gpu.h
      val o = REC_BYTES * i
      def u(k: Int) = rec(o + k) & 0xFF
      val id = (u(12)) | (u(13) << 8) | (u(14) << 16) | (u(15) << 24)
      val area = (u(4) | (u(5) << 8)).toLong | ((u(6) | (u(7) << 8)).toLong << 16) | ((u(10) | (u(11) << 8)).toLong << 32)   // DEMI_REC_AREA
      def msg = lo.decodeFields(u(3), lo.model.fieldsOf(area))
      u(0) match {
        case REC_SPAWN => externals(u(9)) match { case Start(ctor, n) => trace += SpawnEvent("", ctor(), n, null) }
        case REC_KILL => trace += KillEvent(lo.actorName(u(2)))
        case REC_PARTITION => trace += PartitionEvent((lo.actorName(u(1)), lo.actorName(u(2))))
        case REC_UNPARTITION => trace += UnPartitionEvent((lo.actorName(u(1)), lo.actorName(u(2))))
        case REC_BEGIN_WAIT_QUIESCENCE => trace += BeginWaitQuiescence
        case REC_QUIESCENCE => trace += Quiescence
        case REC_MSG_SEND => trace += UniqueMsgSend(MsgSend(name(u(1), lo), lo.actorName(u(2)), msg), id)
        case REC_MSG_EVENT => trace += UniqueMsgEvent(MsgEvent(name(u(1), lo), lo.actorName(u(2)), msg), id)
      }
    }
    trace
  }

  /** DepTracker.getInitialTrace (DepTracker.scala:130-133, 173) of a recorded execution as demi_dpor_trace_entry[] (key u64, word u32,
   *  parent, qperiod, depth, kind - 16 bytes): the root, then every delivery, each identified by the hash chain of its causal path
   *  (include/demi_gpu.h: key(child) = (key(parent) ^ word) * FNV prime; the parent is the delivery during which it was sent, the
   *  root for external messages).  What demi_amd/incremental_ddmin.py dpor_initial_trace computes from the same records. */
  val DPOR_ROOT_KEY = 0xCBF29CE484222325L
  val DPOR_PRIME = 0x100000001B3L
  def dporInitialTrace(trace: EventTrace, lo: TableLowering): Array[Byte] = {
    def actor(n: String) = if (n == "deadLetters" || n == "Timer") lo.model.deadLetters else lo.actorId(n)
    // (a wide table's node keys hash the 64-bit word; the entry reports its low half)
    def word(s: String, r: String, m: Any): Long = { val (t, f) = lo.encodeFields(m); lo.model.word(t, actor(s), lo.actorId(r), lo.model.area(f)) }
    val keyOfId = scala.collection.mutable.Map[Int, (Long, Int)]()        // Uniq id -> (node key, trace index of its producer)
    val entries = scala.collection.mutable.ArrayBuffer[(Long, Long, Int, Int, Int)]((DPOR_ROOT_KEY, 0L, 0, 0, 0))    // key, word, parent, depth, kind
    var curKey = DPOR_ROOT_KEY; var curIdx = 0
    for (e <- trace.events) e match {
      case u @ UniqueMsgSend(MsgSend(s, r, m), id) =>
        val (pk, pi) = if (EventTypes.isExternal(u)) (DPOR_ROOT_KEY, 0) else (curKey, curIdx)
        keyOfId(id) = (((pk ^ word(s, r, m)) * DPOR_PRIME), pi)
      case UniqueMsgEvent(MsgEvent(s, r, m), id) =>
        val (k, pi) = keyOfId(id)
        entries += ((k, word(s, r, m), pi & 0xFF, (entries(pi)._4 + 1) & 0xFF, 1))
        curKey = k; curIdx = entries.size - 1
      case _ =>
    }
    val out = new Array[Byte](16 * entries.size)
    for (((k, w, parent, depth, kind), i) <- entries.zipWithIndex) {
      val o = 16 * i
      for (b <- 0 until 8) out(o + b) = (k >>> (8 * b)).toByte
      for (b <- 0 until 4) out(o + 8 + b) = (w >>> (8 * b)).toByte
      out(o + 12) = parent.toByte; out(o + 13) = 0; out(o + 14) = depth.toByte; out(o + 15) = kind.toByte
    }
    out
  }

  /** EventTrace -> demi_rec_event[] (for replayLoad): the inverse of toEventTrace for the records the GPU path produces. */
  def packRecorded(trace: EventTrace, lo: TableLowering): Array[Byte] = {
    val evs = trace.events.toSeq
    val out = new Array[Byte](REC_BYTES * evs.size)
    val sends = trace.original_externals.zipWithIndex.collect { case (Send(_, _), i) => i }.iterator
    val spawns = scala.collection.mutable.Map[String, Int]() ++
      trace.original_externals.zipWithIndex.collect { case (Start(_, n), i) => n -> i }
    def put(i: Int, kind: Int, snd: Int, rcv: Int, t: Int, area: Long, fl: Int, ext: Int, id: Int) {
      val o = REC_BYTES * i
      out(o) = kind.toByte; out(o + 1) = snd.toByte; out(o + 2) = rcv.toByte; out(o + 3) = t.toByte
      out(o + 4) = area.toByte; out(o + 5) = (area >> 8).toByte; out(o + 6) = (area >> 16).toByte; out(o + 7) = (area >> 24).toByte   // p0, p1
      out(o + 8) = fl.toByte; out(o + 9) = ext.toByte; out(o + 10) = (area >> 32).toByte; out(o + 11) = (area >> 40).toByte            // p_hi
      out(o + 12) = id.toByte; out(o + 13) = (id >> 8).toByte; out(o + 14) = (id >> 16).toByte; out(o + 15) = (id >> 24).toByte
    }
    def actor(n: String) = if (n == "deadLetters" || n == "Timer") lo.model.deadLetters else lo.actorId(n)
    for ((e, i) <- evs.zipWithIndex) e match {
      case SpawnEvent(_, _, n, _) => put(i, REC_SPAWN, 0, lo.actorId(n), 0, 0L, 0, spawns.getOrElse(n, 255), 0)
      case KillEvent(n) => put(i, REC_KILL, 0, lo.actorId(n), 0, 0L, 0, 255, 0)
      case PartitionEvent((a, b)) => put(i, REC_PARTITION, lo.actorId(a), lo.actorId(b), 0, 0L, 0, 255, 0)
      case UnPartitionEvent((a, b)) => put(i, REC_UNPARTITION, lo.actorId(a), lo.actorId(b), 0, 0L, 0, 255, 0)
      case BeginWaitQuiescence => put(i, REC_BEGIN_WAIT_QUIESCENCE, 0, 0, 0, 0L, 0, 255, 0)
      case Quiescence => put(i, REC_QUIESCENCE, 0, 0, 0, 0L, 0, 255, 0)
      case UniqueMsgSend(MsgSend(s, r, m), id) =>
        val (t, f) = lo.encodeFields(m)
        val external = EventTypes.isExternal(e)       // the k-th external MsgSend belongs to the k-th Send (EventTrace.scala:382-452)
        put(i, REC_MSG_SEND, actor(s), lo.actorId(r), t, lo.model.area(f), if (external) 1 else 0, if (external) sends.next() else 255, id)
      case UniqueMsgEvent(MsgEvent(s, r, m), id) =>
        val (t, f) = lo.encodeFields(m)
        put(i, REC_MSG_EVENT, actor(s), lo.actorId(r), t, lo.model.area(f), 0, 255, id)
      case other => throw new UnsupportedOnGpu(other.getClass.getSimpleName)
    }
    out
  }
}
