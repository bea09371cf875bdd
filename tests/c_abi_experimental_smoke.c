/*
 * c_abi_experimental_smoke.c — a plain-C99 consumer of include/fmhip_experimental.h (the measurement / experiment surface:
 * named tuning keys, profiling, emulation, layout queries, a transport of the caller's own), compiled and run by
 * tests/test_host_cpu.py beside c_abi_smoke.c: the header is C, includes the product header, and the entry points that need
 * no GPU validate their arguments.
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "fmhip_experimental.h"

#define CHECK(cond)                                                       \
    do {                                                                  \
        if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

int main(void) {
    fmhip_profile pr;
    fmhip_comm_profile cp;
    fmhip_tune_key key = FMHIP_TUNE_FLAT_ADDRESS;
    memset(&pr, 0, sizeof pr);
    memset(&cp, 0, sizeof cp);
    CHECK(fmhip_version() == FMHIP_VERSION);
    CHECK(sizeof(pr.ms) == 8 * FMHIP_K_COUNT && sizeof(cp.exposed_ms) == 8);
    CHECK(fmhip_ablation_mask() == 0);
    CHECK((int)key == 8 && FMHIP_TUNE_HOT_PAGES == 12 && FMHIP_TUNE_KEY_COUNT == 13);
    CHECK(fmhip_tune(FMHIP_TUNE_KEY_COUNT, 1) == FMHIP_ERR_INVALID && fmhip_tune(-1, 1) == FMHIP_ERR_INVALID);
    CHECK(fmhip_model_tune(NULL, FMHIP_TUNE_XCD_PLACEMENT, 2) == FMHIP_ERR_INVALID);
    CHECK(fmhip_comm_create_external(NULL, 0, 1, NULL, NULL, NULL) == FMHIP_ERR_INVALID);
    CHECK(fmhip_dataset_hot_pages(NULL, NULL, NULL, NULL, NULL) == FMHIP_ERR_INVALID);
    CHECK(FMHIP_HOT_PAGES * 16 <= 128);
    CHECK(fmhip_dataset_partition_rows(NULL, 0) == FMHIP_ERR_INVALID);
    CHECK(fmhip_comm_emulate(NULL, 1.0) == FMHIP_ERR_INVALID && fmhip_comm_profile_begin(NULL) == FMHIP_ERR_INVALID);
    printf("c_abi_experimental_smoke ok (fmhip %d)\n", fmhip_version());
    return 0;
}
