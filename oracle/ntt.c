/*
 * ntt.c -- the reference's Fr "NTT" (src/utils.rs:39-129), restated twice:
 *   ntt_381 / i_ntt_381 : faithful O(n^2) DFT (one constant-time pow per matrix entry)
 *   ntt_fast            : O(n log n) radix-2 with bit-identical natural-order output
 *                         (all values are unique reduced residues, so any exact algorithm agrees)
 * TEST INFRASTRUCTURE ONLY (see bp_oracle.h).
 */
#include "bp_oracle.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* utils.rs:39-43 */
void ntt_root_of_unity(fr_t *r, uint64_t group_order) {
    uint64_t e[4] = {((uint64_t)1 << 32) / group_order, 0, 0, 0};
    fr_pow(r, &FR_ROOT_OF_UNITY, e);
}
/* utils.rs:45-52 */
void ntt_roots_of_unity(fr_t *out, uint64_t group_order) {
    fr_t g;
    ntt_root_of_unity(&g, group_order);
    fr_from_u64(&out[0], 1);
    for (uint64_t i = 1; i < group_order; i++) fr_mul(&out[i], &out[i - 1], &g);
}
/* utils.rs:54-61 */
size_t ntt_find_next_power_of_two(size_t n, size_t m) {
    size_t power = 1, target = n + m + 1;
    while (power < target) power <<= 1;
    return power;
}
/* utils.rs:82-84 */
static int is_power_of_two(uint64_t n) { return n != 0 && (n & (n - 1)) == 0; }

static int dft_faithful(fr_t *out, const fr_t *in, size_t n, const fr_t *gen, int scale) {
    if (!is_power_of_two(n)) return -1;          /* utils.rs:65,108 assert */
    fr_t ninv;
    if (scale) {
        fr_from_u64(&ninv, n);
        fr_invert(&ninv, &ninv);                  /* utils.rs:126 */
    }
    for (uint64_t x = 0; x < n; x++) {
        fr_t sum, w, t;
        fr_zero(&sum);
        for (uint64_t y = 0; y < n; y++) {
            uint64_t e[4] = {x * y * (((uint64_t)1 << 32) / n), 0, 0, 0};   /* utils.rs:76,119-124 */
            fr_pow(&w, gen, e);
            fr_mul(&t, &in[y], &w);
            fr_add(&sum, &sum, &t);
        }
        if (scale) fr_mul(&sum, &sum, &ninv);
        out[x] = sum;
    }
    return 0;
}
/* utils.rs:63-81 */
int ntt_381(fr_t *out, const fr_t *in, size_t n) { return dft_faithful(out, in, n, &FR_ROOT_OF_UNITY, 0); }
/* utils.rs:106-129 */
int i_ntt_381(fr_t *out, const fr_t *in, size_t n) { return dft_faithful(out, in, n, &FR_ROOT_OF_UNITY_INV, 1); }

static size_t bitrev(size_t x, int bits) {
    size_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

int ntt_fast_mt(fr_t *a, size_t n, int inverse, int threads) {
    if (!is_power_of_two(n)) return -1;
    int logn = 0;
    while (((size_t)1 << logn) < n) logn++;
    if (n == 1) return 0;
    /* omega_n = ROOT^(2^32/n)  (utils.rs:39-43, 76, 119) */
    fr_t w_n;
    uint64_t e[4] = {((uint64_t)1 << 32) / n, 0, 0, 0};
    fr_pow(&w_n, inverse ? &FR_ROOT_OF_UNITY_INV : &FR_ROOT_OF_UNITY, e);
    fr_t *tw = malloc((n / 2) * sizeof *tw);
    fr_from_u64(&tw[0], 1);
    for (size_t i = 1; i < n / 2; i++) fr_mul(&tw[i], &tw[i - 1], &w_n);
    for (size_t i = 0; i < n; i++) {
        size_t j = bitrev(i, logn);
        if (i < j) { fr_t t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (int s = 1; s <= logn; s++) {
        size_t m = (size_t)1 << s, half = m >> 1, step = n / m;
#pragma omp parallel for schedule(static) num_threads(threads) if (threads > 1 && n >= 4096)
        for (long b = 0; b < (long)(n / 2); b++) {
            size_t k = (size_t)b / half * m, j = (size_t)b % half;
            fr_t t, u = a[k + j];
            fr_mul(&t, &a[k + j + half], &tw[j * step]);
            fr_add(&a[k + j], &u, &t);
            fr_sub(&a[k + j + half], &u, &t);
        }
    }
    if (inverse) {
        fr_t ninv;
        fr_from_u64(&ninv, n);
        fr_invert(&ninv, &ninv);
        for (size_t i = 0; i < n; i++) fr_mul(&a[i], &a[i], &ninv);
    }
    free(tw);
    return 0;
}
int ntt_fast(fr_t *a, size_t n, int inverse) { return ntt_fast_mt(a, n, inverse, 1); }
