// HipSGD.scala — the Scala side of the drop-in: `HipSGD extends FMLearn`, to be added to SparkFM next to
// ALS (src/main/scala/io/edstud/spark/fm/lib/).  SOURCE ONLY: there is no JVM, Scala compiler or sbt in
// the build image, so this file has never been compiled; the tested binding of the same C symbols is
// sparkfm_amd/_ffi.py (ctypes).  Every native below maps 1:1 onto include/fmhip.h through jvm/fmhip_jni.c.
//
// Plug-in point mirrored:  abstract class FMLearn { def learn(fm: FMModel, dataset: DataSet): FMModel }
//                          (S/fm/FMLearn.scala:10-12), selected with FM(...).learnWith(HipSGD.run(...))
//                          exactly like ALS.run() (S/fm/lib/ALS.scala:202-208, S/driver.scala:106-110).
package io.edstud.spark.fm.lib

import io.edstud.spark.DataSet
import io.edstud.spark.fm.{FMLearn, FMModel}

/** MI355X mini-batch SGD learner.  One `learn` call = one epoch over the cached rows (all mini-batches).
  * world > 1: data-parallel over the GPUs of one node — this instance is one rank (one executor / one GPU),
  * `uniqueId` the 128 bytes rank 0 obtained from `HipSGD.commUniqueId()` and the driver broadcast; the
  * gradient exchange (RCCL over xGMI) happens inside the native library, as the reference's learner does
  * its own reduction inside `learn` (S/fm/lib/ALS.scala:153). */
class HipSGD protected (eta: Double, reg0: Double, regw: Double, regv: Double, batchRows: Long, device: Int,
                        rank: Int, world: Int, uniqueId: Array[Byte]) extends FMLearn {

  @transient private var model: Long = 0L     // fmhip_model_t
  @transient private var data: Long = 0L      // fmhip_dataset_t
  @transient private var comm: Long = 0L      // fmhip_comm_t (world > 1)
  @transient private var cached: DataSet = null
  /** relabel feature ids by descending frequency before the upload (off: ids go to the GPU as the loader produced them) */
  var relabel: Boolean = false
  /** exchange only the gradient rows some rank touched instead of the dense gradient (models far wider than a global batch) */
  var touchedRowsExchange: Boolean = false
  @transient private var rank: Array[Int] = null
  @transient private var byRank: Array[Int] = null

  override def learn(fm: FMModel, dataset: DataSet): FMModel = {
    if (cached ne dataset) {                   // first call: flatten the RDD rows to CSR and upload them once
      // (label, SparseVector) — in local[*] this is in-process; on a cluster every executor passes the rows of
      // ITS partitions (rdd.mapPartitions) to its own HipSGD rank.  Rows of rank r: HipSGD.shardRows(...)
      val rows = dataset.rdd.collect()
      val rowPtr = new Array[Long](rows.length + 1)
      var p = 0L; var r = 0
      while (r < rows.length) { p += rows(r)._2.used; r += 1; rowPtr(r) = p }
      val col = new Array[Int](p.toInt); val value = new Array[Double](p.toInt); val y = new Array[Double](rows.length)
      var o = 0; r = 0
      while (r < rows.length) {
        val sv = rows(r)._2                    // breeze SparseVector: first `used` entries of index/data, stored order
        System.arraycopy(sv.index, 0, col, o, sv.used); System.arraycopy(sv.data, 0, value, o, sv.used)
        y(r) = rows(r)._1; o += sv.used; r += 1
      }
      if (relabel) {                           // ids by descending frequency: internal row r = the caller's feature byRank(r)
        val n1 = fm.num_attribute + 1
        val counts = new Array[Long](n1)
        HipSGD.featureCounts(col, n1, counts)  // world > 1: sum `counts` over the ranks here (rdd.treeReduce / allreduce)
        rank = new Array[Int](n1); byRank = new Array[Int](n1)
        HipSGD.rankFromCounts(counts, rank, byRank)
        HipSGD.relabelColumns(col, rank)
      }
      if (data != 0L) HipSGD.datasetDestroy(data)
      data = HipSGD.datasetCreate(device, rows.length, rowPtr, col, value, y, batchRows)
      if (model == 0L) model = HipSGD.modelCreate(device, fm.num_attribute, fm.num_factor)
      if (world > 1 && comm == 0L) {
        comm = HipSGD.commCreate(model, uniqueId, rank, world)
        if (touchedRowsExchange) HipSGD.dpExchange(comm, 1)
        HipSGD.dpPlan(model, data, comm, Array(0.05, 0.15, 0.3, 0.55))   // cuts of the backward for the overlapped all-reduce
      }
      cached = dataset
    }
    val k = fm.num_factor
    if (!relabel) HipSGD.setParams(model, fm.w0, fm.w.data, fm.v.data)   // breeze column-major == ABI layout (f + i*k)
    else {                                                          // the same arrays in the internal numbering
      val n1 = byRank.length; val wi = new Array[Double](n1); val vi = new Array[Double](n1 * k)
      var r = 0
      while (r < n1) { wi(r) = fm.w.data(byRank(r)); System.arraycopy(fm.v.data, byRank(r) * k, vi, r * k, k); r += 1 }
      HipSGD.setParams(model, fm.w0, wi, vi)
    }
    if (world > 1) HipSGD.dpEpoch(model, data, comm, eta, reg0, regw, regv)
    else HipSGD.sgdEpoch(model, data, eta, reg0, regw, regv)
    val w0 = new Array[Double](1)
    if (!relabel) HipSGD.getParams(model, w0, fm.w.data, fm.v.data)  // mutate in place and return, as ALS does (:27,:40,:64,:74)
    else {
      val n1 = byRank.length; val wi = new Array[Double](n1); val vi = new Array[Double](n1 * k)
      HipSGD.getParams(model, w0, wi, vi)
      var r = 0
      while (r < n1) { fm.w.data(byRank(r)) = wi(r); System.arraycopy(vi, r * k, fm.v.data, byRank(r) * k, k); r += 1 }
    }
    fm.w0 = w0(0)
    fm
  }

  /** Model.computeRMSE(dataset) (S/Model.scala:13-19) on held-out rows: a scoring-only upload. */
  def computeRMSE(fm: FMModel, rowPtr: Array[Long], col: Array[Int], value: Array[Double], y: Array[Double]): Double = {
    if (model == 0L) model = HipSGD.modelCreate(device, fm.num_attribute, fm.num_factor)
    HipSGD.setParams(model, fm.w0, fm.w.data, fm.v.data)
    val rows = HipSGD.rowsCreate(device, y.length, rowPtr, col, value, y)
    try HipSGD.rmse(model, rows) finally HipSGD.datasetDestroy(rows)
  }

  def close(): Unit = {
    if (comm != 0L) { HipSGD.commDestroy(comm); comm = 0L }
    if (data != 0L) { HipSGD.datasetDestroy(data); data = 0L }
    if (model != 0L) { HipSGD.modelDestroy(model); model = 0L }
    cached = null
  }
}

object HipSGD {
  System.loadLibrary("fmhip_jni")                                   // jvm/fmhip_jni.c, links libfmhip.so

  /** Mirrors ALS.run() (S/fm/lib/ALS.scala:202-208). */
  def run(eta: Double = 0.05, reg0: Double = 0, regw: Double = 0, regv: Double = 0,
          batchRows: Long = 250000L, device: Int = 0): HipSGD =
    new HipSGD(eta, reg0, regw, regv, batchRows, device, 0, 1, null)

  /** One rank of a data-parallel job: `uniqueId` from `commUniqueId()` on rank 0, broadcast by the driver. */
  def runDistributed(rank: Int, world: Int, uniqueId: Array[Byte], eta: Double = 0.05, reg0: Double = 0,
                     regw: Double = 0, regv: Double = 0, batchRows: Long = 625000L): HipSGD =
    new HipSGD(eta, reg0, regw, regv, batchRows, rank, rank, world, uniqueId)

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
  @native def commUniqueId(): Array[Byte]
  @native def commCreate(model: Long, uniqueId: Array[Byte], rank: Int, world: Int): Long
  @native def commDestroy(h: Long): Unit
  @native def dpPlan(model: Long, data: Long, comm: Long, upperFractions: Array[Double]): Unit
  @native def dpEpoch(model: Long, data: Long, comm: Long, eta: Double, reg0: Double, regw: Double, regv: Double): Unit
  /** 0 = dense packed gradient (default), 1 = touched rows only (Criteo-width models); every rank, before dpPlan */
  @native def dpExchange(comm: Long, mode: Int): Unit
  /** [lo, hi) of `rank`, balanced by stored nonzeros (fmhip_shard_rows). */
  @native def shardRows(rowPtr: Array[Long], world: Int, rank: Int): Array[Long]
  // feature relabelling by frequency (a pure renaming; ids that arrive hashed or in dictionary order cost ~20 % of the forward):
  // counts accumulate into `counts` (sum the tables of all ranks before ranking), relabelColumns rewrites `col` in place
  @native def featureCounts(col: Array[Int], n1: Long, counts: Array[Long]): Unit
  @native def rankFromCounts(counts: Array[Long], rank: Array[Int], byRank: Array[Int]): Unit
  @native def relabelColumns(col: Array[Int], rank: Array[Int]): Unit
}
