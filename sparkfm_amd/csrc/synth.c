/*
 * synth.c — seeded synthetic sparse regression data for bench.py and the tests
 * (BASELINE.md §3 / SURVEY.md §8(d): rows with U{lo..hi} distinct Zipf(s) feature ids,
 * values 1.0 w.p. 0.5 else U(0.1,1), labels from a planted FM (k_true factors) + noise).
 *
 * Host-only helper (plain C, OpenMP).  Every row owns its own PCG64 stream derived from
 * (seed, global row index), so any shard [row_begin, row_begin + n_rows) of the same
 * virtual dataset can be generated independently (one shard per GPU rank) and the result
 * does not depend on the thread count.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef __uint128_t u128;

typedef struct { u128 state, inc; } pcg64;

static const u128 PCG_MULT = ((u128)0x2360ED051FC65DA4ULL << 64) | 0x4385DF649FCCF645ULL;

static uint64_t splitmix64(uint64_t *x) {
    uint64_t z = (*x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static uint64_t pcg_next(pcg64 *r) {
    r->state = r->state * PCG_MULT + r->inc;
    uint64_t hi = (uint64_t)(r->state >> 64), lo = (uint64_t)r->state;
    uint64_t x = hi ^ lo;
    unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((64 - rot) & 63));
}

static void pcg_seed(pcg64 *r, uint64_t seed, uint64_t stream) {
    uint64_t s = seed ^ (stream * 0xD1342543DE82EF95ULL);
    uint64_t a = splitmix64(&s), b = splitmix64(&s), c = splitmix64(&s), d = splitmix64(&s);
    r->inc = ((((u128)a << 64) | b) << 1) | 1;
    r->state = ((u128)c << 64) | d;
    (void)pcg_next(r);
}

static double pcg_unif(pcg64 *r) { return (double)(pcg_next(r) >> 11) * (1.0 / 9007199254740992.0); }

static double pcg_normal(pcg64 *r) {
    double u1 = pcg_unif(r), u2 = pcg_unif(r);
    if (u1 < 1e-300) u1 = 1e-300;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

/* planted model: w0* = 0.1, w* ~ N(0, 0.1), V* ~ N(0, 0.1)  (k_true x n, feature-major) */
static void planted(uint64_t seed, int64_t n, int k_true, double *w, double *v) {
    pcg64 r;
    pcg_seed(&r, seed, 0xFFFFFFFFFFFFULL);
    for (int64_t i = 0; i < n; ++i) {
        w[i] = 0.1 * pcg_normal(&r);
        for (int f = 0; f < k_true; ++f) v[i * k_true + f] = 0.1 * pcg_normal(&r);
    }
}

static int64_t zipf_draw(const double *cdf, int64_t n, double u) {
    int64_t lo = 0, hi = n - 1;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (cdf[mid] < u) lo = mid + 1; else hi = mid;
    }
    return lo;
}

static int row_nnz(pcg64 *r, int nnz_lo, int nnz_hi, int64_t n) {
    int m = nnz_lo + (int)(pcg_next(r) % (uint64_t)(nnz_hi - nnz_lo + 1));
    return (int64_t)m > n ? (int)n : m;
}

/* pass 1: row_ptr[n_rows + 1] */
void fms_zipf_count(uint64_t seed, int64_t row_begin, int64_t n_rows, int64_t n_features, int nnz_lo, int nnz_hi,
                    int64_t *row_ptr) {
    row_ptr[0] = 0;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n_rows; ++r) {
        pcg64 g;
        pcg_seed(&g, seed, (uint64_t)(row_begin + r));
        row_ptr[r + 1] = row_nnz(&g, nnz_lo, nnz_hi, n_features);
    }
    for (int64_t r = 0; r < n_rows; ++r) row_ptr[r + 1] += row_ptr[r];
}

/* pass 2: col/val/y for the row_ptr of pass 1.  zipf_s <= 0 means uniform ids. */
int fms_zipf_fill(uint64_t seed, int64_t row_begin, int64_t n_rows, int64_t n_features, int nnz_lo, int nnz_hi,
                  double zipf_s, int k_true, double noise, const int64_t *row_ptr, int32_t *col, float *val,
                  float *y) {
    double *cdf = (double *)malloc((size_t)n_features * sizeof(double));
    double *w = (double *)malloc((size_t)n_features * sizeof(double));
    double *v = (double *)malloc((size_t)n_features * (size_t)(k_true > 0 ? k_true : 1) * sizeof(double));
    if (!cdf || !w || !v) { free(cdf); free(w); free(v); return -1; }
    double tot = 0.0;
    for (int64_t i = 0; i < n_features; ++i) {
        tot += zipf_s > 0 ? pow((double)(i + 1), -zipf_s) : 1.0;
        cdf[i] = tot;
    }
    for (int64_t i = 0; i < n_features; ++i) cdf[i] /= tot;
    planted(seed, n_features, k_true, w, v);
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t r = 0; r < n_rows; ++r) {
        pcg64 g;
        pcg_seed(&g, seed, (uint64_t)(row_begin + r));
        const int m = row_nnz(&g, nnz_lo, nnz_hi, n_features);
        int32_t *c = col + row_ptr[r];
        float *x = val + row_ptr[r];
        int have = 0;
        while (have < m) {
            int32_t id = (int32_t)zipf_draw(cdf, n_features, pcg_unif(&g));
            int dup = 0;
            for (int j = 0; j < have; ++j) if (c[j] == id) { dup = 1; break; }
            if (dup) continue;
            c[have++] = id;
        }
        double yy = 0.1, q[16], s[16];
        for (int f = 0; f < k_true && f < 16; ++f) { q[f] = 0.0; s[f] = 0.0; }
        for (int j = 0; j < m; ++j) {
            float xv = (pcg_next(&g) & 1) ? 1.0f : (float)(0.1 + 0.9 * pcg_unif(&g));
            x[j] = xv;
            yy += w[c[j]] * xv;
            for (int f = 0; f < k_true && f < 16; ++f) {
                double t = v[(int64_t)c[j] * k_true + f] * xv;
                q[f] += t;
                s[f] += t * t;
            }
        }
        for (int f = 0; f < k_true && f < 16; ++f) yy += 0.5 * (q[f] * q[f] - s[f]);
        y[r] = (float)(yy + noise * pcg_normal(&g));
    }
    free(cdf); free(w); free(v);
    return 0;
}
