/*
 * msm.c -- BucketMSM (src/msm.rs) restated in plain C, algorithm-faithful:
 * k = b/c windows, window 0 = most significant c bits, 2^c - 1 buckets with a
 * running-sum reduction, Horner combine with c doublings per window.
 * TEST INFRASTRUCTURE ONLY (see bp_oracle.h).
 */
#include "bp_oracle.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* msm.rs:119-139 (+ u8_to_bool_array :64-75, bools_to_u64 :51-62): bits
 * [start, start+c) of the canonical integer written MSB-first over 256 bits. */
uint64_t msm_get_c_bit_chunk(const fr_t *scalar, size_t chunk_index, size_t chunk_size) {
    uint8_t bytes[32];
    fr_to_bytes(bytes, scalar);          /* little-endian canonical (scalar.rs:292) */
    size_t start = chunk_index * chunk_size;
    uint64_t res = 0;
    for (size_t k = 0; k < chunk_size; k++) {
        size_t be_bit = start + k;       /* index in the big-endian bit string */
        size_t le_bit = 255 - be_bit;
        uint64_t bit = (bytes[le_bit >> 3] >> (le_bit & 7)) & 1;
        res = (res << 1) | bit;
    }
    return res;
}

/* msm.rs:23-49 */
void msm_c_bit_msm(g1_proj_t *r, const g1_proj_t *points, const uint64_t *digits, size_t n, size_t c) {
    size_t num_buckets = ((size_t)1 << c) - 1;
    g1_proj_t *buckets = malloc(num_buckets * sizeof *buckets);
    g1_proj_t acc, res, t;
    for (size_t i = 0; i < num_buckets; i++) g1_identity(&buckets[i]);
    for (size_t i = 0; i < n; i++) {
        if (digits[i] != 0) {
            g1_add(&t, &buckets[digits[i] - 1], &points[i]);
            buckets[digits[i] - 1] = t;
        }
    }
    g1_identity(&acc);
    g1_identity(&res);
    for (size_t i = num_buckets; i-- > 0;) {
        g1_add(&t, &acc, &buckets[i]);
        acc = t;
        g1_add(&t, &res, &acc);
        res = t;
    }
    free(buckets);
    *r = res;
}

static void window_sum(g1_proj_t *t, const g1_proj_t *points, const fr_t *scalars, size_t n, size_t i, size_t c) {
    uint64_t *digits = malloc((n ? n : 1) * sizeof *digits);
    for (size_t j = 0; j < n; j++) digits[j] = msm_get_c_bit_chunk(&scalars[j], i, c);
    msm_c_bit_msm(t, points, digits, n, c);
    free(digits);
}

static void horner(g1_proj_t *r, const g1_proj_t *t_points, size_t k, size_t c) {
    g1_proj_t res = t_points[0], t;      /* msm.rs:107-115 */
    for (size_t j = 1; j < k; j++) {
        for (size_t d = 0; d < c; d++) g1_double(&res, &res);
        g1_add(&t, &res, &t_points[j]);
        res = t;
    }
    *r = res;
}

/* msm.rs:76-118.  zip() in c_bit_msm truncates to the shorter of points / scalars (msm.rs:29). */
void msm_bucket_msm(g1_proj_t *r, const g1_proj_t *points, size_t n_points,
                    const fr_t *scalars, size_t n_scalars, size_t b, size_t c) {
    size_t k = b / c, n = n_points < n_scalars ? n_points : n_scalars;
    g1_proj_t *t_points = malloc(k * sizeof *t_points);
    for (size_t i = 0; i < k; i++) window_sum(&t_points[i], points, scalars, n, i, c);
    horner(r, t_points, k, c);
    free(t_points);
}

void msm_bucket_msm_mt(g1_proj_t *r, const g1_proj_t *points, size_t n_points,
                       const fr_t *scalars, size_t n_scalars, size_t b, size_t c, int threads) {
    size_t k = b / c, n = n_points < n_scalars ? n_points : n_scalars;
    g1_proj_t *t_points = malloc(k * sizeof *t_points);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (long i = 0; i < (long)k; i++) window_sum(&t_points[i], points, scalars, n, (size_t)i, c);
    horner(r, t_points, k, c);
    free(t_points);
}
