// sparkfm.hpp — a header-only C++17 mirror of SparkFM's host classes over the C ABI of libfmhip.so (include/fmhip.h).
//
// SparkFM is compiled JVM code; no JVM exists where this was built, so beside the Python mirror (sparkfm_amd/) and the
// Scala / JNI sources a maintainer would add (jvm/), this is the same surface for a COMPILED host: same class names, same
// argument meaning, same call order and error behaviour as the reference —
//
//     sparkfm::DataSet      S/DataSet.scala:42-62            rows of (label, SparseVector), cache / unpersist, size, dimension
//     sparkfm::FMModel      S/fm/FMModel.scala:9-63          num_attribute, num_factor, public w0 / w / v, reg0 / regw / regv, predict
//                           S/Model.scala:13-19              computeRMSE
//     sparkfm::FMLearn      S/fm/FMLearn.scala:10-12         the plug-in point: learn(fm, dataset): FMModel
//     sparkfm::HipSGD       (build-defined; SparkFM ships ALS only) one learn = one epoch of mini-batch SGD on the GPU
//     sparkfm::HipALS       S/fm/lib/ALS.scala:15-75,202-208 the reference's own learner in fp64 on the GPU
//     sparkfm::FM           S/fm/FM.scala:25-33, S/fm/impl/FactorizationMachines.scala:30-51   the fit loop
//
// Nothing but include/fmhip.h (the product header) is used.  The reference throws JVM exceptions (S/DataCollection.scala:36);
// here a non-zero status of the C ABI becomes sparkfm::Error carrying fmhip_last_error().  Parameters live on the host as in
// the reference (public, mutable: `fm.w0`, `fm.w`, `fm.v` with v[f + i*k] = breeze's column-major DenseMatrix(k, n+1)); every
// call that needs them on the device uploads them first, as jvm/HipSGD.scala does (the fit loop calls `learn` once per
// iteration: S/fm/impl/FactorizationMachines.scala:45).  There is no CPU fallback: every computation is a call into libfmhip.so.
#ifndef SPARKFM_HPP
#define SPARKFM_HPP

#include <cmath>
#include <cstdint>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "fmhip.h"

namespace sparkfm {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &what) : std::runtime_error("fmhip error " + std::to_string(c) + ": " + what), code(c) {}
};
inline void check(int rc) {
    if (rc != FMHIP_OK) throw Error(rc, fmhip_last_error());
}

// breeze.linalg.SparseVector[Double] as the reference uses it: index / data in stored order (need not be sorted)
struct SparseVector {
    std::vector<int32_t> index;
    std::vector<double> data;
};

// S/DataSet.scala:42-62 — the rows live on the host until cache() uploads them (mini-batches of batch_rows rows; 0 = one batch,
// what HipALS needs); unpersist() drops the device copy (S/fm/impl/FactorizationMachines.scala:36,48)
class DataSet {
  public:
    explicit DataSet(const std::vector<std::pair<double, SparseVector>> &rows, int64_t batch_rows = 0, int device = 0)
        : batch_rows_(batch_rows), device_(device) {
        row_ptr_.push_back(0);
        for (const auto &r : rows) {
            if (r.second.index.size() != r.second.data.size()) throw Error(FMHIP_ERR_INVALID, "index / data length mismatch");
            col_.insert(col_.end(), r.second.index.begin(), r.second.index.end());
            val_.insert(val_.end(), r.second.data.begin(), r.second.data.end());
            y_.push_back(r.first);
            row_ptr_.push_back((int64_t)col_.size());
        }
    }
    DataSet(std::vector<int64_t> row_ptr, std::vector<int32_t> col, std::vector<double> val, std::vector<double> y, int64_t batch_rows = 0, int device = 0)
        : row_ptr_(std::move(row_ptr)), col_(std::move(col)), val_(std::move(val)), y_(std::move(y)), batch_rows_(batch_rows), device_(device) {
        if (row_ptr_.size() != y_.size() + 1) throw Error(FMHIP_ERR_INVALID, "row_ptr must have one entry more than there are labels");
    }
    DataSet(const DataSet &) = delete;
    DataSet &operator=(const DataSet &) = delete;
    ~DataSet() { (void)fmhip_dataset_destroy(h_); }

    DataSet &cache() {      // dataset.cache() + transposeInput (S/DataSet.scala:48-62)
        if (!h_) check(fmhip_dataset_create(device_, (int64_t)y_.size(), row_ptr_.data(), col_.data(), val_.data(), y_.data(), batch_rows_, &h_));
        return *this;
    }
    DataSet &unpersist() {
        check(fmhip_dataset_destroy(h_));
        h_ = nullptr;
        return *this;
    }
    int64_t size() const { return (int64_t)y_.size(); }                 // S/DataSet.scala:23-25
    int64_t dimension() const {                                           // S/DataSet.scala:27-29: the largest feature index
        int64_t d = 0;
        for (int32_t c : col_) d = c > d ? c : d;
        return d;
    }
    int64_t n_batches() {
        int64_t nb = 0;
        check(fmhip_dataset_info(cache().h_, nullptr, nullptr, nullptr, nullptr, &nb));
        return nb;
    }
    const std::vector<double> &labels() const { return y_; }
    fmhip_dataset_t handle() { return cache().h_; }
    int device() const { return device_; }

  private:
    std::vector<int64_t> row_ptr_;
    std::vector<int32_t> col_;
    std::vector<double> val_, y_;
    int64_t batch_rows_;
    int device_;
    fmhip_dataset_t h_ = nullptr;
};

// S/fm/FMModel.scala:9-63 — `new FMModel(num_attribute, num_factor)`: w0 = 0, w = 0, v ~ N(mean, stdev) (:17-22; the reference's
// draw is unseeded — quirk Q2 — so parity runs assign w0 / w / v explicitly)
class FMModel {
  public:
    const int64_t num_attribute;
    const int32_t num_factor;
    double w0 = 0.0;
    std::vector<double> w, v;                       // w[n+1]; v[k * (n+1)], element (f, i) at f + i*k
    double reg0 = 0.0, regw = 0.0, regv = 10.0;     // S/fm/FMModel.scala:29-31 (ALS ridge terms; HipSGD carries its own)

    FMModel(int64_t numAttribute, int32_t numFactor, double mean = 0.0, double stdev = 0.01, uint64_t seed = 0, int device = 0)
        : num_attribute(numAttribute), num_factor(numFactor), w((size_t)numAttribute + 1, 0.0), v((size_t)numFactor * ((size_t)numAttribute + 1)), device_(device) {
        std::mt19937_64 gen(seed);
        std::normal_distribution<double> dist(mean, stdev);
        for (double &x : v) x = dist(gen);
    }
    FMModel(const FMModel &) = delete;
    FMModel &operator=(const FMModel &) = delete;
    FMModel(FMModel &&o) noexcept
        : num_attribute(o.num_attribute), num_factor(o.num_factor), w0(o.w0), w(std::move(o.w)), v(std::move(o.v)), reg0(o.reg0), regw(o.regw), regv(o.regv),
          device_(o.device_), h_(o.h_) {
        o.h_ = nullptr;
    }
    ~FMModel() { (void)fmhip_model_destroy(h_); }

    double predict(const SparseVector &features) {                        // S/fm/FMModel.scala:34-55
        const int64_t rp[2] = {0, (int64_t)features.index.size()};
        double yhat = 0.0;
        check(fmhip_predict_rows(upload(), 1, rp, features.index.data(), features.data.data(), &yhat));
        return yhat;
    }
    std::vector<double> predict(DataSet &dataset) {                       // dataset.rdd.mapValues(predict), S/Model.scala:14
        std::vector<double> yhat((size_t)dataset.size());
        check(fmhip_predict(upload(), dataset.handle(), yhat.data()));
        return yhat;
    }
    double computeRMSE(DataSet &dataset) {                                // S/Model.scala:13-19
        double rmse = 0.0;
        check(fmhip_rmse(upload(), dataset.handle(), &rmse, nullptr));
        return rmse;
    }

    // the device replica: created on first use, refreshed from the host fields before every use (they are public and mutable)
    fmhip_model_t upload() {
        if (!h_) check(fmhip_model_create(device_, num_attribute, num_factor, nullptr, &h_));
        check(fmhip_model_set_params(h_, w0, w.data(), v.data()));
        return h_;
    }
    void download() { check(fmhip_model_get_params(h_, &w0, w.data(), v.data())); }

  private:
    int device_;
    fmhip_model_t h_ = nullptr;
};

// S/fm/FMLearn.scala:10-12 — the plug-in point; the model is mutated in place and returned (S/fm/lib/ALS.scala:27,40,64,74)
class FMLearn {
  public:
    virtual ~FMLearn() = default;
    virtual FMModel &learn(FMModel &fm, DataSet &dataset) = 0;
};

// One learn = one epoch of mini-batch SGD over the dataset's batches (ascending order) — fmhip_sgd_epoch.
// theta <- theta - eta * (sum_{r in batch} e_r h_r(theta) / |batch| + reg * theta), e and h from S/fm/lib/ALS.scala:142-144, :56-58 / :40 / :21
class HipSGD : public FMLearn {
  public:
    double eta, reg0, regw, regv;
    fmhip_stats last_stats{};
    explicit HipSGD(double eta_ = 0.05, double reg0_ = 0.0, double regw_ = 0.0, double regv_ = 0.0) : eta(eta_), reg0(reg0_), regw(regw_), regv(regv_) {}
    static HipSGD run(double eta = 0.05, double reg0 = 0.0, double regw = 0.0, double regv = 0.0) { return HipSGD(eta, reg0, regw, regv); }   // cf. ALS.run(), S/fm/lib/ALS.scala:202-208
    FMModel &learn(FMModel &fm, DataSet &dataset) override {
        check(fmhip_sgd_epoch(fm.upload(), dataset.handle(), eta, reg0, regw, regv, nullptr, &last_stats));
        fm.download();
        return fm;
    }
};

// The reference's own learner: one learn = one ALS.learn pass (S/fm/lib/ALS.scala:15-75) in fp64 on the GPU, with the MODEL's
// regularisers as the reference uses them (:21, :40, :56).  Needs a one-batch DataSet (batch_rows = 0).
class HipALS : public FMLearn {
  public:
    static HipALS run() { return HipALS(); }
    FMModel &learn(FMModel &fm, DataSet &dataset) override {
        check(fmhip_als_epoch(fm.upload(), dataset.handle(), fm.reg0, fm.regw, fm.regv));
        fm.download();
        return fm;
    }
};

// FM(dataset, numFactor, maxIteration).learnWith(learner) — S/fm/FM.scala:25-33 and the fit loop of
// S/fm/impl/FactorizationMachines.scala:30-51: cache; new FMModel(dimension, numFactor); maxIteration x { computeRMSE (logged);
// fm = fml.learn(fm, dataset) }; unpersist
class FM {
  public:
    std::vector<double> rmse_history;
    FM(DataSet &dataset, int32_t numFactor, int maxIteration = 100, uint64_t seed = 0) : dataset_(dataset), numFactor_(numFactor), maxIteration_(maxIteration), seed_(seed) {}
    // init (optional): called on the fresh model before the first iteration — parity runs inject w0 / w / v here (quirk Q2)
    template <class Init>
    FMModel learnWith(FMLearn &fml, Init init) {
        DataSet &ds = dataset_.cache();                                          // :36
        FMModel fm(ds.dimension(), numFactor_, 0.0, 0.01, seed_, ds.device());   // :39
        init(fm);
        for (int i = 1; i <= maxIteration_; ++i) {                               // :42
            rmse_history.push_back(fm.computeRMSE(ds));                          // :43 (logged and discarded in the reference)
            fml.learn(fm, ds);                                                   // :45
        }
        ds.unpersist();                                                          // :48
        return fm;
    }
    FMModel learnWith(FMLearn &fml) {
        return learnWith(fml, [](FMModel &) {});
    }

  private:
    DataSet &dataset_;
    int32_t numFactor_;
    int maxIteration_;
    uint64_t seed_;
};

}  // namespace sparkfm
#endif  // SPARKFM_HPP
