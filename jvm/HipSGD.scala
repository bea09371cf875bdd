// HipSGD.scala — the Scala side of the drop-in: `HipSGD extends FMLearn`, to be added to SparkFM next to
// ALS (src/main/scala/io/edstud/spark/fm/lib/).  SOURCE ONLY: there is no JVM, Scala compiler or sbt in
// the build image, so this file has never been compiled; what IS checked here (tests/test_host_cpu.py) is that
// every `@native def` below has a JNI function of the same name and arity in jvm/fmhip_jni.c and that every
// fmhip_* symbol that file calls is declared in include/fmhip.h.  The tested binding of the same C symbols is
// sparkfm_amd/_ffi.py (ctypes).
//
// Plug-in point mirrored:  abstract class FMLearn { def learn(fm: FMModel, dataset: DataSet): FMModel }
//                          (S/fm/FMLearn.scala:10-12), selected with FM(...).learnWith(HipSGD.run(...))
//                          exactly like ALS.run() (S/fm/lib/ALS.scala:202-208, S/driver.scala:106-110) and
//                          called by the fit loop once per iteration (S/fm/impl/FactorizationMachines.scala:45).
//
// Where the rows live.  The reference's learner pulls the WHOLE transposed dataset to the driver
// (`collectAsMap`, S/fm/lib/ALS.scala:34) — the limitation SURVEY.md a8 records.  This learner does not:
//   gpus == 1   the rows are flattened on the driver and trained on one GPU of the driver's host (spark-local, C1-C3);
//   gpus  > 1   `learn` runs one Spark job of `gpus` tasks: task r is rank r of the data-parallel job, flattens the
//               rows of ITS partition only, keeps them on GPU r % gpusPerHost for as long as the executor lives, and the
//               ranks exchange gradients among themselves inside the native library (RCCL over xGMI,
//               fmhip_dp_epoch).  The driver ships the 128-byte communicator id and the parameters (a broadcast)
//               and reads rank 0's parameters back: no row ever travels to the driver.
//               Needs `gpus` task slots running AT THE SAME TIME (local[N] with N >= gpus, or one executor core per
//               GPU): the ranks wait for each other inside the collectives.
package io.edstud.spark.fm.lib

import java.util.concurrent.ConcurrentHashMap

import breeze.linalg.SparseVector
import io.edstud.spark.DataSet
import io.edstud.spark.fm.{FMLearn, FMModel}

/** MI355X mini-batch SGD learner.  One `learn` call = one epoch over the cached rows (all mini-batches).
  * The gradient exchange of a multi-GPU run happens inside the native library, as the reference's learner does
  * its own reduction inside `learn` (S/fm/lib/ALS.scala:153). */
class HipSGD protected (eta: Double, reg0: Double, regw: Double, regv: Double, batchRows: Long, gpus: Int,
                        gpusPerHost: Int) extends FMLearn {

  /** relabel feature ids by descending frequency before the upload (off: ids go to the GPU as the loader produced them) */
  var relabel: Boolean = false
  /** what a data-parallel step exchanges: 0 = the dense gradient, all-reduced in slices under the backward, every rank
    * applying the identical update (default: the faster of the two dense modes in every schedule measured so far —
    * DESIGN.md section 7; bench.py times both on the node it runs on); 2 = the same reduce-scattered, every rank updating
    * its share; 1 = only the rows some rank touched (models far wider than a global batch); 3 = the dense exchange with
    * consecutive steps overlapped (the coldest slice travels beside the next position's forward: 14 % faster than 0 against
    * emulated 8 x 300 GB/s collectives, slower where the step is compute-bound — bench.py times it with the others) */
  var exchange: Int = 0
  /** visit the mini-batches in a seeded random order, a fresh permutation per epoch (None = ascending).  Data-parallel: every
    * rank draws the SAME permutation of the epoch's positions (fmhip_dp_epoch_order) */
  var shuffleSeed: Option[Long] = None
  private var epoch: Long = 0L
  /** cuts of the backward for the overlapped exchange (fmhip_dp_plan) */
  var upperFractions: Array[Double] = Array(0.05, 0.15, 0.3, 0.55)

  // one job key per learner instance: the executors keep their rank's device state under (jobKey, rank)
  private val jobKey: Long = HipSGD.nextJobKey()
  @transient private var local: HipSGD.RankState = null     // gpus == 1: the driver-side state
  @transient private var cached: DataSet = null
  @transient private var uniqueId: Array[Byte] = null

  override def learn(fm: FMModel, dataset: DataSet): FMModel = {
    if (gpus <= 1) learnOnDriver(fm, dataset) else learnOnExecutors(fm, dataset)
  }

  // ---- one GPU, in the driver's process (spark-local) -------------------------------------------------------------
  private def learnOnDriver(fm: FMModel, dataset: DataSet): FMModel = {
    if (cached ne dataset) {                   // first call: flatten the RDD rows to CSR and upload them once
      if (local != null) local.close()
      local = HipSGD.upload(dataset.rdd.collect().iterator, fm.num_attribute, fm.num_factor, batchRows, 0, relabel, null)
      cached = dataset
    }
    local.setParams(fm.w0, fm.w.data, fm.v.data)
    HipSGD.sgdEpoch(local.model, local.data, eta, reg0, regw, regv)
    fm.w0 = local.getParams(fm.w.data, fm.v.data)   // mutate in place and return, as ALS does (:27,:40,:64,:74)
    fm
  }

  // ---- `gpus` ranks on the executors --------------------------------------------------------------------------------
  private def learnOnExecutors(fm: FMModel, dataset: DataSet): FMModel = {
    val sc = dataset.rdd.sparkContext
    val world = gpus
    // the rows of rank r = partition r of the cached RDD cut into `world` partitions (a narrow coalesce of the cached
    // parent: the same rows reach the same rank on every call, which is what lets a rank keep its upload)
    val parts = if (dataset.rdd.partitions.length == world) dataset.rdd else dataset.rdd.coalesce(world, shuffle = false)
    if (uniqueId == null) {
      // rank 0's executor creates the communicator id (its RCCL bootstrap listener must live where rank 0 lives)
      uniqueId = parts.mapPartitionsWithIndex((r, _) => if (r == 0) Iterator(HipSGD.commUniqueId()) else Iterator.empty).collect().head
    }
    // frequency ranks are a property of the WHOLE dataset: counted per partition, summed on the driver, shipped back
    val idRank: Array[Int] =
      if (!relabel) null
      else {
        val n1 = fm.num_attribute + 1
        val counts = parts.mapPartitions { rows =>
          // A native call per CHUNK of at most 2^24 ids (not per row: every call copies the n1-long counts array in and
          // out; not per partition: a partition may hold more than 2^31 - 1 ids, and flattening it whole doubles its
          // memory; 64 MB of staging whatever the partition's size).  featureCounts accumulates into `c`, so the chunks
          // just follow each other.
          val c = new Array[Long](n1)
          val chunk = new Array[Int](1 << 24)
          var o = 0
          def flush(): Unit = if (o > 0) { HipSGD.featureCounts(java.util.Arrays.copyOf(chunk, o), n1, c); o = 0 }
          rows.foreach { case (_, sv) =>
            var i = 0
            while (i < sv.used) {                  // a row longer than the chunk is cut across calls
              val n = math.min(sv.used - i, chunk.length - o)
              System.arraycopy(sv.index, i, chunk, o, n)
              o += n; i += n
              if (o == chunk.length) flush()
            }
          }
          flush()
          Iterator(c)
        }.reduce { (a, b) => var i = 0; while (i < a.length) { a(i) += b(i); i += 1 }; a }
        val rank = new Array[Int](n1); val byRank = new Array[Int](n1)
        HipSGD.rankFromCounts(counts, rank, byRank)
        rank
      }
    val params = sc.broadcast((fm.w0, fm.w.data, fm.v.data, uniqueId, idRank))
    val (key, nAttr, nFac, bRows, perHost, xchg, fracs) = (jobKey, fm.num_attribute, fm.num_factor, batchRows, gpusPerHost, exchange, upperFractions)
    val (e, r0, rw, rv) = (eta, reg0, regw, regv)
    val orderSeed: Option[Long] = shuffleSeed.map(_ + epoch)
    epoch += 1
    val out = parts.mapPartitionsWithIndex { (rank, rows) =>
      val (w0, w, v, id, ranks) = params.value
      val st = HipSGD.rankState(key, rank) {
        // first task of this rank on this executor: upload the partition's rows, join the communicator, plan the exchange
        val s = HipSGD.upload(rows, nAttr, nFac, bRows, rank % perHost, ranks != null, ranks)
        s.comm = HipSGD.commCreate(s.model, id, rank, world)
        HipSGD.dpExchange(s.comm, xchg)
        HipSGD.dpPlan(s.model, s.data, s.comm, fracs)
        s
      }
      st.setParams(w0, w, v)
      orderSeed match {      // every rank the same number of steps, in the same order; replicas identical
        case None => HipSGD.dpEpoch(st.model, st.data, st.comm, e, r0, rw, rv)
        case Some(seed) =>
          val steps = HipSGD.dpPlanSteps(st.comm).toInt                  // agreed over all ranks by dpPlan
          val order = new scala.util.Random(seed).shuffle((0 until steps).toList).map(_.toLong).toArray
          HipSGD.dpEpochOrder(st.model, st.data, st.comm, e, r0, rw, rv, order)
      }
      if (rank == 0) {
        val wOut = new Array[Double](w.length); val vOut = new Array[Double](v.length)
        val w0Out = st.getParams(wOut, vOut)
        Iterator((w0Out, wOut, vOut))
      } else Iterator.empty
    }.collect().head
    cached = dataset
    fm.w0 = out._1
    System.arraycopy(out._2, 0, fm.w.data, 0, out._2.length)
    System.arraycopy(out._3, 0, fm.v.data, 0, out._3.length)
    params.unpersist()
    fm
  }

  /** Model.computeRMSE(dataset) (S/Model.scala:13-19) on held-out rows: a scoring-only upload on GPU 0 of this host. */
  def computeRMSE(fm: FMModel, rowPtr: Array[Long], col: Array[Int], value: Array[Double], y: Array[Double]): Double = {
    val model = HipSGD.modelCreate(0, fm.num_attribute, fm.num_factor)
    try {
      HipSGD.setParams(model, fm.w0, fm.w.data, fm.v.data)
      val rows = HipSGD.rowsCreate(0, y.length, rowPtr, col, value, y)
      try HipSGD.rmse(model, rows) finally HipSGD.datasetDestroy(rows)
    } finally HipSGD.modelDestroy(model)
  }

  /** Frees the device state: the driver's, and (one job) every rank's on its executor. */
  def close(dataset: DataSet = cached): Unit = {
    if (local != null) { local.close(); local = null }
    if (gpus > 1 && dataset != null) {
      val key = jobKey
      dataset.rdd.coalesce(gpus, shuffle = false).mapPartitionsWithIndex { (rank, _) => HipSGD.dropRankState(key, rank); Iterator.empty }.count()
    }
    cached = null
  }
}

object HipSGD {
  System.loadLibrary("fmhip_jni")                                   // jvm/fmhip_jni.c, links libfmhip.so

  /** Mirrors ALS.run() (S/fm/lib/ALS.scala:202-208): one GPU of the driver's host. */
  def run(eta: Double = 0.05, reg0: Double = 0, regw: Double = 0, regv: Double = 0, batchRows: Long = 250000L): HipSGD =
    new HipSGD(eta, reg0, regw, regv, batchRows, 1, 1)

  /** Data-parallel over `gpus` GPUs (one Spark task = one rank = one GPU; `gpusPerHost` GPUs per executor host). */
  def runDistributed(gpus: Int, gpusPerHost: Int = 8, eta: Double = 0.05, reg0: Double = 0, regw: Double = 0,
                     regv: Double = 0, batchRows: Long = 625000L): HipSGD =
    new HipSGD(eta, reg0, regw, regv, batchRows, gpus, gpusPerHost)

  /** Device state of one rank: handles of the native library plus the relabelling tables of its upload. */
  final class RankState(val model: Long, val data: Long, val k: Int, val idByRank: Array[Int]) {
    var comm: Long = 0L
    /** breeze column-major `v.data` == the ABI's layout (f + i*k); relabelled uploads move rows to the internal numbering */
    def setParams(w0: Double, w: Array[Double], v: Array[Double]): Unit = {
      if (idByRank == null) HipSGD.setParams(model, w0, w, v)
      else {
        val n1 = idByRank.length; val wi = new Array[Double](n1); val vi = new Array[Double](n1 * k)
        var r = 0
        while (r < n1) { wi(r) = w(idByRank(r)); System.arraycopy(v, idByRank(r) * k, vi, r * k, k); r += 1 }
        HipSGD.setParams(model, w0, wi, vi)
      }
    }
    /** fills `w`, `v` (the caller's numbering) and returns w0 */
    def getParams(w: Array[Double], v: Array[Double]): Double = {
      val w0 = new Array[Double](1)
      if (idByRank == null) HipSGD.getParams(model, w0, w, v)
      else {
        val n1 = idByRank.length; val wi = new Array[Double](n1); val vi = new Array[Double](n1 * k)
        HipSGD.getParams(model, w0, wi, vi)
        var r = 0
        while (r < n1) { w(idByRank(r)) = wi(r); System.arraycopy(vi, r * k, v, idByRank(r) * k, k); r += 1 }
      }
      w0(0)
    }
    def close(): Unit = {
      if (comm != 0L) HipSGD.commDestroy(comm)
      HipSGD.datasetDestroy(data)
      HipSGD.modelDestroy(model)
    }
  }

  /** Flattens (label, SparseVector) rows to CSR — breeze SparseVector: the first `used` entries of index/data, stored
    * order (S/DataSet.scala:42) — and uploads them to `device`.  `idRank` given: ids are relabelled with it;
    * `relabelLocally`: the ranks are computed from these rows alone (single-GPU runs). */
  def upload(rows: Iterator[(Double, SparseVector[Double])], numAttribute: Int, numFactor: Int, batchRows: Long, device: Int,
             relabelLocally: Boolean, idRank: Array[Int]): RankState = {
    val buf = rows.toArray
    val rowPtr = new Array[Long](buf.length + 1)
    var p = 0L; var r = 0
    while (r < buf.length) { p += buf(r)._2.used; r += 1; rowPtr(r) = p }
    val col = new Array[Int](p.toInt); val value = new Array[Double](p.toInt); val y = new Array[Double](buf.length)
    var o = 0; r = 0
    while (r < buf.length) {
      val sv = buf(r)._2
      System.arraycopy(sv.index, 0, col, o, sv.used); System.arraycopy(sv.data, 0, value, o, sv.used)
      y(r) = buf(r)._1; o += sv.used; r += 1
    }
    val n1 = numAttribute + 1
    var rank = idRank
    var byRank: Array[Int] = null
    if (rank == null && relabelLocally) {
      val counts = new Array[Long](n1)
      featureCounts(col, n1, counts)
      rank = new Array[Int](n1); byRank = new Array[Int](n1)
      rankFromCounts(counts, rank, byRank)
    } else if (rank != null) {
      byRank = new Array[Int](n1)
      var i = 0
      while (i < n1) { byRank(rank(i)) = i; i += 1 }
    }
    if (rank != null) relabelColumns(col, rank)
    val data = datasetCreate(device, buf.length, rowPtr, col, value, y, batchRows)
    val model = modelCreate(device, numAttribute, numFactor)
    new RankState(model, data, numFactor, byRank)
  }

  // executor-side registry: the device state of (job, rank) outlives the task that created it
  private val states = new ConcurrentHashMap[(Long, Int), RankState]()
  private val jobKeys = new java.util.concurrent.atomic.AtomicLong(System.nanoTime())
  def nextJobKey(): Long = jobKeys.incrementAndGet()
  def rankState(job: Long, rank: Int)(create: => RankState): RankState = states.synchronized {
    var s = states.get((job, rank))
    if (s == null) { s = create; states.put((job, rank), s) }
    s
  }
  def dropRankState(job: Long, rank: Int): Unit = states.synchronized {
    val s = states.remove((job, rank))
    if (s != null) s.close()
  }

  // ---- include/fmhip.h, one native per entry point used (a non-zero status becomes a RuntimeException
  //      carrying fmhip_last_error())
  @native def modelCreate(device: Int, numAttribute: Long, numFactor: Int): Long
  @native def modelDestroy(h: Long): Unit
  @native def setParams(h: Long, w0: Double, w: Array[Double], v: Array[Double]): Unit
  @native def getParams(h: Long, w0: Array[Double], w: Array[Double], v: Array[Double]): Unit
  @native def datasetCreate(device: Int, nRows: Long, rowPtr: Array[Long], col: Array[Int],
                            value: Array[Double], y: Array[Double], batchRows: Long): Long
  @native def rowsCreate(device: Int, nRows: Long, rowPtr: Array[Long], col: Array[Int],
                         value: Array[Double], y: Array[Double]): Long
  @native def datasetDestroy(h: Long): Unit
  @native def sgdEpoch(model: Long, data: Long, eta: Double, reg0: Double, regw: Double, regv: Double): Unit
  @native def rmse(model: Long, data: Long): Double
  @native def predict(model: Long, data: Long, yhat: Array[Double]): Unit
  @native def predictRows(model: Long, nRows: Long, rowPtr: Array[Long], col: Array[Int], value: Array[Double],
                          yhat: Array[Double]): Unit
  @native def deviceCount(): Int
  @native def commUniqueId(): Array[Byte]
  @native def commCreate(model: Long, uniqueId: Array[Byte], rank: Int, world: Int): Long
  @native def commDestroy(h: Long): Unit
  @native def dpPlan(model: Long, data: Long, comm: Long, upperFractions: Array[Double]): Unit
  @native def dpEpoch(model: Long, data: Long, comm: Long, eta: Double, reg0: Double, regw: Double, regv: Double): Unit
  /** 0 = dense all-reduce, 1 = touched rows only (Criteo-width models), 2 = reduce-scatter + sharded update + all-gather;
    * every rank, before dpPlan */
  @native def dpExchange(comm: Long, mode: Int): Unit
  /** the lock-step steps of an epoch agreed by dpPlan (the largest batch count of any rank) */
  @native def dpPlanSteps(comm: Long): Long
  /** dpEpoch with the positions visited in `order` — a permutation of 0 until dpPlanSteps, the same on every rank */
  @native def dpEpochOrder(model: Long, data: Long, comm: Long, eta: Double, reg0: Double, regw: Double, regv: Double,
                           order: Array[Long]): Unit
  /** [lo, hi) of `rank`, balanced by stored nonzeros (fmhip_shard_rows). */
  @native def shardRows(rowPtr: Array[Long], world: Int, rank: Int): Array[Long]
  // feature relabelling by frequency (a pure renaming; ids that arrive hashed or in dictionary order cost ~20 % of the forward):
  // counts accumulate into `counts` (sum the tables of all ranks before ranking), relabelColumns rewrites `col` in place
  @native def featureCounts(col: Array[Int], n1: Long, counts: Array[Long]): Unit
  @native def rankFromCounts(counts: Array[Long], rank: Array[Int], byRank: Array[Int]): Unit
  @native def relabelColumns(col: Array[Int], rank: Array[Int]): Unit
}
