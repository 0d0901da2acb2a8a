/*
 * helpers.c -- synthetic-input generators and timing helpers for tests and the
 * bench's cpu_baseline leg.  Not part of the reference; TEST INFRASTRUCTURE ONLY.
 */
#include "bp_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <time.h>

static uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
/* 64 bytes from a fixed-seed counter PRNG per element, reduced as
 * Scalar::from_bytes_wide (scalar.rs:308-339).  Element i uses stream state seed + 8*i. */
void oracle_splitmix_scalars(fr_t *out, size_t n, uint64_t seed) {
    for (size_t i = 0; i < n; i++) {
        uint64_t s = seed + 0x9e3779b97f4a7c15ull * 8 * i, limbs[8];
        for (int k = 0; k < 8; k++) limbs[k] = splitmix64(&s);
        fr_from_u512(&out[i], limbs);
    }
}
/* P_i = (a + i*d) * G, affine */
void oracle_points_progression(g1_affine_t *out, size_t n, const fr_t *a, const fr_t *d) {
    if (n == 0) return;
    g1_proj_t g, cur, step, t, *proj = malloc(n * sizeof *proj);
    g1_affine_t step_aff;
    g1_generator(&g);
    g1_mul(&cur, &g, a);
    g1_mul(&step, &g, d);
    g1_to_affine(&step_aff, &step);
    for (size_t i = 0; i < n; i++) {
        proj[i] = cur;
        g1_add_mixed(&t, &cur, &step_aff);
        cur = t;
    }
    g1_batch_normalize(proj, out, n);
    free(proj);
}
void oracle_points_to_bytes96(uint8_t *out, const g1_affine_t *p, size_t n) {
    for (size_t i = 0; i < n; i++) g1_to_uncompressed(out + 96 * i, &p[i]);
}
void oracle_proj_from_bytes96(g1_proj_t *out, const uint8_t *in, size_t n) {
    for (size_t i = 0; i < n; i++) {
        g1_affine_t a;
        g1_from_uncompressed_unchecked(&a, in + 96 * i);
        g1_from_affine(&out[i], &a);
    }
}
double oracle_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
/* r = sum_i s_i * (a + i*d)  -- the closed-form exponent of an MSM over the progression points */
void oracle_dot_progression(fr_t *r, const fr_t *s, size_t n, const fr_t *a, const fr_t *d) {
    fr_t acc, cur = *a, t;
    fr_zero(&acc);
    for (size_t i = 0; i < n; i++) {
        fr_mul(&t, &s[i], &cur);
        fr_add(&acc, &acc, &t);
        fr_add(&cur, &cur, d);
    }
    *r = acc;
}
