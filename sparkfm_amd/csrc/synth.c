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
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef __uint128_t u128;

/* launchers export OMP_NUM_THREADS=1 to their ranks (torch.distributed.run does): the caller states the thread count */
void fms_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

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

/* ------------------------------------------------------------------------------------------------
 * C5: Criteo-shaped rows (BASELINE.json config 5).  39 fields per row, each
 * missing w.p. 0.1: 13 numeric fields (one fixed slot each, value log1p-like in [0,8)) and 26
 * categorical fields (value 1.0, one id out of the field's own vocabulary, popularity Zipf(s_f),
 * s_f in 1.1 .. 1.3), all hashed into `n_hash` slots — the one-hot layout the reference's ETL
 * produces before training (S/fm/util/StandardVectorizor.scala:62-87), with the hashing trick in
 * place of its dictionary.  Vocabulary sizes are the Criteo display-ads cardinalities (sum ~33.8 M).
 * Two fields can hash to one slot, so a row may hold the same index twice (legal for the reference:
 * its loader neither sorts nor de-duplicates, S/fm/FMUtils.scala:28-36).
 * Labels: planted FM whose parameters are a hash of the slot id (no n_hash-sized tables).
 * ------------------------------------------------------------------------------------------------ */
#define C5_NUM 13
#define C5_CAT 26
#define C5_FIELDS (C5_NUM + C5_CAT)

static const int64_t c5_vocab[C5_CAT] = {1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194,
                                         27, 14992, 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572};

static uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xFF51AFD7ED558CCDULL; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ULL; x ^= x >> 33;
    return x;
}

static int32_t c5_slot(int field, int64_t value, int64_t n_hash) {
    return (int32_t)(mix64(((uint64_t)(field + 1) << 40) ^ (uint64_t)value ^ 0x9E3779B97F4A7C15ULL) % (uint64_t)n_hash);
}

/* Zipf(s) over {1..n} by rejection-inversion (Hörmann & Derflinger 1996): O(1) per draw, no table */
typedef struct { double s, n, h_x1, h_n, cut; } zipf_ri;
static double zri_h(const zipf_ri *z, double x) { return exp((1.0 - z->s) * log(x)) / (1.0 - z->s); }
static double zri_hinv(const zipf_ri *z, double x) { return exp(log((1.0 - z->s) * x) / (1.0 - z->s)); }
static void zri_init(zipf_ri *z, double s, int64_t n) {
    z->s = s; z->n = (double)n;
    z->h_x1 = zri_h(z, 1.5) - 1.0;
    z->h_n = zri_h(z, z->n + 0.5);
    z->cut = 2.0 - zri_hinv(z, zri_h(z, 2.5) - exp(-s * log(2.0)));
}
static int64_t zri_draw(const zipf_ri *z, pcg64 *r) {
    for (;;) {
        const double u = z->h_n + pcg_unif(r) * (z->h_x1 - z->h_n);
        const double x = zri_hinv(z, u);
        double k = floor(x + 0.5);
        if (k < 1.0) k = 1.0;
        if (k > z->n) k = z->n;
        if (k - x <= z->cut || u >= zri_h(z, k + 0.5) - exp(-z->s * log(k))) return (int64_t)k;
    }
}

/* planted parameters of slot i: w* ~ N(0, 0.1), V*[f] ~ N(0, 0.1), from a hash of (seed, i) */
static void c5_planted(uint64_t seed, int32_t i, int k_true, double *w, double *v) {
    pcg64 r;
    pcg_seed(&r, seed ^ 0xC5C5C5C5ULL, 0x100000000ULL + (uint64_t)i);
    *w = 0.1 * pcg_normal(&r);
    for (int f = 0; f < k_true; ++f) v[f] = 0.1 * pcg_normal(&r);
}

static int c5_row(uint64_t seed, int64_t row, int64_t n_hash, const zipf_ri *zf, int32_t *c, float *x) {
    pcg64 g;
    pcg_seed(&g, seed, (uint64_t)row);
    int m = 0;
    for (int f = 0; f < C5_FIELDS; ++f) {
        const int present = pcg_unif(&g) >= 0.1;
        if (f < C5_NUM) {
            /* count-like numeric: log1p of a heavy-tailed count, clipped below 8 */
            const double cnt = exp(6.0 * pcg_unif(&g) * pcg_unif(&g)) - 1.0;
            double val = log1p(cnt) * 1.3;
            if (val >= 7.999) val = 7.999;
            if (val < 0.05) val = 0.05;              /* a stored zero would be an explicit zero entry */
            if (present && c) { c[m] = c5_slot(f, 0, n_hash); x[m] = (float)val; }
        } else {
            const int64_t id = zri_draw(&zf[f - C5_NUM], &g);
            if (present && c) { c[m] = c5_slot(f, id, n_hash); x[m] = 1.0f; }
        }
        m += present;
    }
    return m;
}

static void c5_zipfs(zipf_ri *zf) {
    for (int j = 0; j < C5_CAT; ++j) zri_init(&zf[j], 1.1 + 0.2 * (double)((j * 7) % C5_CAT) / (double)(C5_CAT - 1), c5_vocab[j]);
}

void fms_criteo_count(uint64_t seed, int64_t row_begin, int64_t n_rows, int64_t n_hash, int64_t *row_ptr) {
    zipf_ri zf[C5_CAT];
    c5_zipfs(zf);
    row_ptr[0] = 0;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n_rows; ++r) row_ptr[r + 1] = c5_row(seed, row_begin + r, n_hash, zf, NULL, NULL);
    for (int64_t r = 0; r < n_rows; ++r) row_ptr[r + 1] += row_ptr[r];
}

int fms_criteo_fill(uint64_t seed, int64_t row_begin, int64_t n_rows, int64_t n_hash, int k_true, double noise,
                    const int64_t *row_ptr, int32_t *col, float *val, float *y) {
    zipf_ri zf[C5_CAT];
    c5_zipfs(zf);
    if (k_true > 16) k_true = 16;
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t r = 0; r < n_rows; ++r) {
        int32_t *c = col + row_ptr[r];
        float *x = val + row_ptr[r];
        const int m = c5_row(seed, row_begin + r, n_hash, zf, c, x);
        double yy = 0.1, q[16], s[16], wi, vi[16];
        for (int f = 0; f < k_true; ++f) { q[f] = 0.0; s[f] = 0.0; }
        for (int j = 0; j < m; ++j) {
            c5_planted(seed, c[j], k_true, &wi, vi);
            /* numeric values reach 8: scale their share so the label stays O(1) */
            const double xv = x[j] > 1.0f ? x[j] * 0.125 : x[j];
            yy += wi * xv;
            for (int f = 0; f < k_true; ++f) { const double t = vi[f] * xv; q[f] += t; s[f] += t * t; }
        }
        for (int f = 0; f < k_true; ++f) yy += 0.5 * (q[f] * q[f] - s[f]);
        pcg64 g;
        pcg_seed(&g, seed ^ 0x5EEDULL, (uint64_t)(row_begin + r));
        y[r] = (float)(yy + noise * pcg_normal(&g));
    }
    return 0;
}
