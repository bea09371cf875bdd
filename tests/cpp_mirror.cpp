// cpp_mirror.cpp — a consumer of include/sparkfm.hpp (the header-only C++ mirror of SparkFM's host classes over the C ABI).
//   cpp_mirror host            no GPU: the mirror compiles against the PRODUCT header alone, and library errors arrive as
//                              sparkfm::Error carrying fmhip_last_error() (the reference throws JVM exceptions)
//   cpp_mirror gpu <out.bin>   the reference's own flow — FM(dataset, k, maxIteration).learnWith(learner), then predict /
//                              computeRMSE (S/driver.scala:100-112) — for HipSGD (mini-batches) and HipALS (one batch) on a
//                              problem made of exact rationals; writes every result as raw doubles so that
//                              tests/test_gpu_configs.py can compare them BIT FOR BIT with the same flow through the Python mirror
#include <cstdio>
#include <cstring>
#include <string>

#include "sparkfm.hpp"

#define CHECK(cond)                                                       \
    do {                                                                  \
        if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

using namespace sparkfm;

// the problem: N rows over n1 = 97 features (prime: the strided ids of a row are distinct), everything an exact rational
static std::vector<std::pair<double, SparseVector>> make_rows(int n_rows, int n1) {
    std::vector<std::pair<double, SparseVector>> rows;
    for (int r = 0; r < n_rows; ++r) {
        SparseVector sv;
        const int nnz = 3 + r % 5;
        for (int j = 0; j < nnz; ++j) {
            sv.index.push_back((r * 7 + j * 31) % n1);
            sv.data.push_back(0.25 + (double)((r + 3 * j) % 8) / 8.0);
        }
        rows.emplace_back((double)((r * 37) % 11) / 5.0 - 1.0, sv);
    }
    return rows;
}
static void inject(FMModel &fm) {
    const int k = fm.num_factor;
    fm.w0 = 0.125;
    for (int64_t i = 0; i <= fm.num_attribute; ++i) {
        fm.w[(size_t)i] = (double)((i * 29) % 17 - 8) / 160.0;
        for (int f = 0; f < k; ++f) fm.v[(size_t)(f + i * k)] = (double)((f * 7 + i * 13) % 23 - 11) / 220.0;
    }
}
static void put(FILE *o, const std::vector<double> &x) { fwrite(x.data(), sizeof(double), x.size(), o); }
static void put(FILE *o, double x) { fwrite(&x, sizeof x, 1, o); }

int main(int argc, char **argv) {
    const std::string mode = argc > 1 ? argv[1] : "host";
    if (mode == "host") {
        bool threw = false;
        try {
            FMModel fm(10, FMHIP_MAX_FACTORS + 1);
            fm.upload();
        } catch (const Error &e) {
            threw = e.code == FMHIP_ERR_UNSUPPORTED && strstr(e.what(), "FMHIP_MAX_FACTORS") != nullptr;
        }
        CHECK(threw);
        threw = false;
        try {
            DataSet bad({0, 2, 1}, {0, 1}, {1.0, 1.0}, {0.0, 1.0});
            bad.cache();
        } catch (const Error &e) {
            threw = e.code == FMHIP_ERR_INVALID && strstr(e.what(), "row_ptr decreases") != nullptr;
        }
        CHECK(threw);
        auto rows = make_rows(20, 97);
        DataSet ds(rows);
        CHECK(ds.size() == 20 && ds.dimension() > 0 && ds.dimension() <= 96);
        FMModel fm(ds.dimension(), 4);
        CHECK(fm.w.size() == (size_t)ds.dimension() + 1 && fm.v.size() == 4 * fm.w.size() && fm.w0 == 0.0 && fm.regv == 10.0);
        double s = 0.0;
        for (double x : fm.v) s += x * x;
        CHECK(s > 0.0 && std::sqrt(s / (double)fm.v.size()) < 0.02);        // v ~ N(0, 0.01) (S/fm/FMModel.scala:19-22)
        HipSGD sgd = HipSGD::run(0.05, 0.0, 1e-3, 1e-3);
        FMLearn &as_plugin = sgd;                                            // the plug-in point is the abstract learner
        (void)as_plugin;
        printf("cpp_mirror host: checks ok\n");
        return 0;
    }
    CHECK(mode == "gpu" && argc > 2);
    FILE *o = fopen(argv[2], "wb");
    CHECK(o != nullptr);
    const int n_rows = 3000, n1 = 97, k = 8;
    auto rows = make_rows(n_rows, n1);
    try {
        // ---- HipSGD: 3 iterations over mini-batches of 700 rows
        {
            DataSet ds(rows, 700);
            HipSGD sgd = HipSGD::run(0.05, 0.0, 1e-3, 1e-3);
            FM fit(ds, k, 3);
            FMModel fm = fit.learnWith(sgd, inject);
            CHECK(fit.rmse_history.size() == 3 && fit.rmse_history[2] < fit.rmse_history[0]);
            put(o, fit.rmse_history);
            put(o, fm.w0);
            put(o, fm.w);
            put(o, fm.v);
            DataSet test(rows, 0);                                           // held-out style scoring (S/driver.scala:100-112)
            put(o, fm.computeRMSE(test));
            put(o, fm.predict(test));
            put(o, fm.predict(rows[5].second));
            put(o, (double)sgd.last_stats.rows);
        }
        // ---- HipALS: the reference's own learner, 2 iterations on a one-batch dataset
        {
            DataSet ds(rows, 0);
            HipALS als = HipALS::run();
            FM fit(ds, k, 2);
            FMModel fm = fit.learnWith(als, inject);
            put(o, fit.rmse_history);
            put(o, fm.w0);
            put(o, fm.w);
            put(o, fm.v);
            put(o, fm.computeRMSE(ds));
        }
    } catch (const Error &e) {
        fprintf(stderr, "sparkfm::Error: %s\n", e.what());
        return 2;
    }
    fclose(o);
    printf("cpp_mirror gpu: flow ok\n");
    return 0;
}
