/*
 * c_abi_smoke.c — a plain-C99 consumer of include/fmhip.h ALONE (the product header: no measurement or experiment entry
 * point is needed to bind the learner), compiled with `gcc -std=c99 -Iinclude` and linked against libfmhip.so by
 * tests/test_host_cpu.py: proves the header is C (not just C++) and that the entry points that need no GPU behave
 * (status codes, fmhip_last_error, host-side sharding).  tests/c_abi_experimental_smoke.c: the same for fmhip_experimental.h.
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "fmhip.h"

#define CHECK(cond)                                                       \
    do {                                                                  \
        if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

int main(void) {
    fmhip_model_t m = NULL;
    fmhip_dataset_t d = NULL;
    fmhip_stats st;
    int64_t row_ptr[6] = {0, 100, 101, 102, 103, 400};
    int64_t bad_ptr[3] = {0, 2, 1};
    int64_t lo = -1, hi = -1, covered = 0;
    int r;

    memset(&st, 0, sizeof st);
    CHECK(fmhip_version() == FMHIP_VERSION);
    CHECK(sizeof(st.sse) == 8 && sizeof(st.steps) == 8);
    /* argument validation happens before any device call */
    CHECK(fmhip_model_create(0, 10, 4, NULL, NULL) == FMHIP_ERR_INVALID);
    CHECK(fmhip_model_create(0, -1, 4, NULL, &m) == FMHIP_ERR_INVALID && m == NULL);
    CHECK(fmhip_model_create(0, 10, FMHIP_MAX_FACTORS + 1, NULL, &m) == FMHIP_ERR_UNSUPPORTED);
    CHECK(strstr(fmhip_last_error(), "FMHIP_MAX_FACTORS") != NULL);
    CHECK(fmhip_dataset_create(0, 2, bad_ptr, NULL, NULL, NULL, 0, &d) == FMHIP_ERR_INVALID && d == NULL);
    CHECK(strstr(fmhip_last_error(), "row_ptr decreases") != NULL);
    CHECK(fmhip_rows_create(0, 2, bad_ptr, NULL, NULL, NULL, &d) == FMHIP_ERR_INVALID);
    {
        fmhip_dataset_opts o;
        memset(&o, 0, sizeof o);
        CHECK(fmhip_dataset_create_opts(0, 2, bad_ptr, NULL, NULL, NULL, &o, &d) == FMHIP_ERR_INVALID);   /* struct_size unset */
        CHECK(strstr(fmhip_last_error(), "struct_size") != NULL);
        o.struct_size = (int32_t)sizeof o;
        o.hot_block = -1;
        o.row_block_rows = -1;
        CHECK(fmhip_dataset_create_opts(0, 2, bad_ptr, NULL, NULL, NULL, &o, &d) == FMHIP_ERR_INVALID);
        CHECK(strstr(fmhip_last_error(), "row_ptr decreases") != NULL);
    }
    CHECK(fmhip_predict(NULL, NULL, NULL) == FMHIP_ERR_INVALID);
    CHECK(fmhip_dp_step(NULL, NULL, 0, NULL, 0.1, 0, 0, 0) == FMHIP_ERR_INVALID);
    CHECK(fmhip_comm_create(NULL, NULL, 0, 1, NULL) == FMHIP_ERR_INVALID);
    CHECK(fmhip_model_destroy(NULL) == FMHIP_OK && fmhip_dataset_destroy(NULL) == FMHIP_OK && fmhip_comm_destroy(NULL) == FMHIP_OK);
    /* nnz-balanced contiguous shards: rows 0 and 4 hold nearly everything */
    for (r = 0; r < 3; ++r) {
        CHECK(fmhip_shard_rows(5, row_ptr, 3, r, &lo, &hi) == FMHIP_OK);
        CHECK(lo == covered && hi >= lo && hi <= 5);
        covered = hi;
    }
    CHECK(covered == 5);
    CHECK(fmhip_shard_rows(5, row_ptr, 3, 0, &lo, &hi) == FMHIP_OK && lo == 0 && hi == 4);   /* 103 of 400 nonzeros is nearest to a third */
    CHECK(fmhip_shard_rows(5, row_ptr, 0, 0, &lo, &hi) == FMHIP_ERR_INVALID);
    /* feature relabelling by frequency: host arithmetic (descending count, ties by ascending id) */
    {
        const int32_t col[9] = {4, 1, 4, 2, 4, 1, 0, 2, 5};
        int32_t out[9], rank[6], by_rank[6];
        int64_t counts[6] = {0, 0, 0, 0, 0, 0};
        const int32_t want_by_rank[6] = {4, 1, 2, 0, 5, 3};      /* counts 1,2,2,0,3,1 */
        int i;
        CHECK(fmhip_feature_counts(5, col, 6, counts) == FMHIP_OK);
        CHECK(fmhip_feature_counts(4, col + 5, 6, counts) == FMHIP_OK);    /* a second partition accumulates */
        CHECK(counts[4] == 3 && counts[1] == 2 && counts[3] == 0);
        CHECK(fmhip_rank_from_counts(6, counts, rank, by_rank) == FMHIP_OK);
        for (i = 0; i < 6; ++i) CHECK(by_rank[i] == want_by_rank[i] && rank[by_rank[i]] == i);
        CHECK(fmhip_relabel_columns(9, col, 6, rank, out) == FMHIP_OK);
        for (i = 0; i < 9; ++i) CHECK(by_rank[out[i]] == col[i]);
        CHECK(fmhip_relabel_columns(9, col, 5, rank, out) == FMHIP_ERR_INVALID);   /* id 5 outside [0, 5) */
    }
    /* communicator-side argument checks */
    CHECK(fmhip_dp_exchange(NULL, FMHIP_EXCHANGE_TOUCHED) == FMHIP_ERR_INVALID);
    CHECK(FMHIP_COLL_ALLGATHER_I32 == 3 && FMHIP_COLL_ALLGATHER_F32 == 5 && FMHIP_EXCHANGE_SHARDED == 2 && FMHIP_EXCHANGE_PIPELINED == 3);
    CHECK(fmhip_dp_steps(NULL, NULL, NULL, 0, NULL, 0.1, 0, 0, 0) == FMHIP_ERR_INVALID);
    printf("c_abi_smoke ok (fmhip %d)\n", fmhip_version());
    return 0;
}
