/*
 * poly.c -- Polynomial operators (src/polynomial.rs:14-380) restated in plain C,
 * including the reference's length rules and its quirks (Sub<Scalar> in the
 * Lagrange basis adds; Div compacts zero quotient coefficients away).
 * TEST INFRASTRUCTURE ONLY (see bp_oracle.h).   basis: 0 = Lagrange, 1 = Monomial.
 */
#include "bp_oracle.h"
#include <stdlib.h>
#include <string.h>

/* polynomial.rs:34-45 -- sum of coeff * x.pow(i), one pow per term */
void poly_coeffs_evaluate(fr_t *r, const fr_t *coeffs, size_t n, const fr_t *x) {
    fr_t res, p, t;
    fr_zero(&res);
    for (size_t i = 0; i < n; i++) {
        uint64_t e[4] = {i, 0, 0, 0};
        fr_pow(&p, x, e);
        fr_mul(&t, &coeffs[i], &p);
        fr_add(&res, &res, &t);
    }
    *r = res;
}
/* same value by Horner's rule (exact field arithmetic => identical bits) */
void poly_coeffs_evaluate_fast(fr_t *r, const fr_t *coeffs, size_t n, const fr_t *x) {
    fr_t res;
    fr_zero(&res);
    for (size_t i = n; i-- > 0;) {
        fr_mul(&res, &res, x);
        fr_add(&res, &res, &coeffs[i]);
    }
    *r = res;
}
/* polynomial.rs:22-33 */
void poly_shift_left(fr_t *out, const fr_t *in, size_t len, size_t n) {
    n %= len;
    for (size_t i = 0; i < len; i++) out[(i + len - n) % len] = in[i];
}
/* polynomial.rs:57-74 */
void poly_add_scalar(fr_t *out, const fr_t *a, size_t n, const fr_t *s, int basis) {
    memcpy(out, a, n * sizeof *a);
    if (basis == 1) {
        fr_add(&out[0], &out[0], s);
    } else {
        for (size_t i = 0; i < n; i++) fr_add(&out[i], &out[i], s);
    }
}
/* polynomial.rs:119-132 -- NOTE the Lagrange branch *adds* (:126-128), as in the reference */
void poly_sub_scalar(fr_t *out, const fr_t *a, size_t n, const fr_t *s, int basis) {
    memcpy(out, a, n * sizeof *a);
    if (basis == 1) {
        fr_sub(&out[0], &out[0], s);
    } else {
        for (size_t i = 0; i < n; i++) fr_add(&out[i], &out[i], s);
    }
}
/* polynomial.rs:176-187 */
void poly_mul_scalar(fr_t *out, const fr_t *a, size_t n, const fr_t *s) {
    for (size_t i = 0; i < n; i++) fr_mul(&out[i], &a[i], s);
}
/* polynomial.rs:76-117; Lagrange asserts equal lengths -> (size_t)-1 */
size_t poly_add(fr_t *out, const fr_t *a, size_t na, const fr_t *b, size_t nb, int basis) {
    if (basis == 0 && na != nb) return (size_t)-1;
    size_t n = na > nb ? na : nb;
    for (size_t i = 0; i < n; i++) {
        fr_zero(&out[i]);
        if (i < na) fr_add(&out[i], &out[i], &a[i]);
        if (i < nb) fr_add(&out[i], &out[i], &b[i]);
    }
    return n;
}
/* polynomial.rs:134-174 */
size_t poly_sub(fr_t *out, const fr_t *a, size_t na, const fr_t *b, size_t nb, int basis) {
    if (basis == 0 && na != nb) return (size_t)-1;
    size_t n = na > nb ? na : nb;
    for (size_t i = 0; i < n; i++) {
        fr_zero(&out[i]);
        if (i < na) fr_add(&out[i], &out[i], &a[i]);
        if (i < nb) fr_sub(&out[i], &out[i], &b[i]);
    }
    return n;
}
/* polynomial.rs:240-273 -- evaluate both at the N roots with coeffs_evaluate, multiply, i_ntt_381, truncate */
size_t poly_mul(fr_t *out, const fr_t *a, size_t na, const fr_t *b, size_t nb) {
    size_t n = na - 1, m = nb - 1, N = ntt_find_next_power_of_two(n, m);
    fr_t *roots = malloc(N * sizeof *roots), *va = malloc(N * sizeof *va), *vb = malloc(N * sizeof *vb);
    ntt_roots_of_unity(roots, N);
    for (size_t i = 0; i < N; i++) {
        poly_coeffs_evaluate(&va[i], a, na, &roots[i]);
        poly_coeffs_evaluate(&vb[i], b, nb, &roots[i]);
        fr_mul(&va[i], &va[i], &vb[i]);
    }
    i_ntt_381(vb, va, N);
    memcpy(out, vb, (n + m + 1) * sizeof *out);
    free(roots); free(va); free(vb);
    return n + m + 1;
}
size_t poly_mul_fast(fr_t *out, const fr_t *a, size_t na, const fr_t *b, size_t nb) {
    size_t n = na - 1, m = nb - 1, N = ntt_find_next_power_of_two(n, m);
    fr_t *va = calloc(N, sizeof *va), *vb = calloc(N, sizeof *vb);
    memcpy(va, a, na * sizeof *a);
    memcpy(vb, b, nb * sizeof *b);
    ntt_fast(va, N, 0);
    ntt_fast(vb, N, 0);
    for (size_t i = 0; i < N; i++) fr_mul(&va[i], &va[i], &vb[i]);
    ntt_fast(va, N, 1);
    memcpy(out, va, (n + m + 1) * sizeof *out);
    free(va); free(vb);
    return n + m + 1;
}
/* polynomial.rs:314-380, as written: one quotient coefficient per loop turn is inserted at
 * the front (:376) while *all* newly zero leading remainder terms are popped (:371-373). */
size_t poly_div(fr_t *out, const fr_t *a, size_t na, const fr_t *b, size_t nb) {
    while (na > 0 && fr_is_zero(&a[na - 1])) na--;       /* :325-331 */
    while (nb > 0 && fr_is_zero(&b[nb - 1])) nb--;       /* :333-339 */
    if (nb == 0) return (size_t)-1;                      /* unwrap() on None panics (:347/:348) */
    fr_t *r = malloc((na ? na : 1) * sizeof *r), *q = malloc((na ? na : 1) * sizeof *q);
    size_t rl = na, ql = 0;
    memcpy(r, a, na * sizeof *a);
    fr_t lead_inv, coeff, t;
    fr_invert(&lead_inv, &b[nb - 1]);
    while (rl >= nb && rl > 0 && !fr_is_zero(&r[rl - 1])) {
        fr_mul(&coeff, &r[rl - 1], &lead_inv);
        size_t diff = rl - nb;
        for (size_t i = 0; i < nb; i++) {                /* r -= coeff * x^diff * b */
            if (fr_is_zero(&b[i])) continue;             /* a zero term subtracts nothing: same values, lets x^n - 1 at n = 2^16 finish */
            fr_mul(&t, &b[i], &coeff);
            fr_sub(&r[diff + i], &r[diff + i], &t);
        }
        while (rl > 0 && fr_is_zero(&r[rl - 1])) rl--;
        q[ql++] = coeff;                                 /* reversed; flipped below */
    }
    for (size_t i = 0; i < ql; i++) out[i] = q[ql - 1 - i];
    free(r); free(q);
    return ql;
}

/* src/utils.rs:161-169 -- Rlc for Scalar: self + other * beta + gamma */
static void rlc(fr_t *r, const fr_t *self, const fr_t *other, const fr_t *beta, const fr_t *gamma) {
    fr_t t;
    fr_mul(&t, other, beta);
    fr_add(&t, self, &t);
    fr_add(r, &t, gamma);
}
/* src/prover.rs:279-319 -- round 2's permutation grand product z(x) in the Lagrange basis, as written:
 * z_0 = 1, z_{i+1} = z_i * prod(numerators) * prod(inverted denominators); asserts z_n == 1 and pops it.
 * returns 0, -1 where a denominator is zero (invert().unwrap() panics), -2 where the final assert fails. */
int prover_round2_z(fr_t *z_out, const fr_t *a, const fr_t *b, const fr_t *c, const fr_t *s1, const fr_t *s2,
                    const fr_t *s3, size_t n, const fr_t *beta, const fr_t *gamma, const fr_t *k1, const fr_t *k2) {
    fr_t *roots = malloc((n ? n : 1) * sizeof *roots);
    fr_t cur, t, u, one;
    ntt_roots_of_unity(roots, n);
    fr_from_u64(&one, 1);
    cur = one;
    int rc = 0;
    for (size_t i = 0; i < n && rc == 0; i++) {
        z_out[i] = cur;
        rlc(&t, &a[i], &roots[i], beta, gamma);
        fr_mul(&cur, &cur, &t);
        fr_mul(&u, &roots[i], k1);
        rlc(&t, &b[i], &u, beta, gamma);
        fr_mul(&cur, &cur, &t);
        fr_mul(&u, &roots[i], k2);
        rlc(&t, &c[i], &u, beta, gamma);
        fr_mul(&cur, &cur, &t);
        const fr_t *w[3] = {&a[i], &b[i], &c[i]}, *s[3] = {&s1[i], &s2[i], &s3[i]};
        for (int j = 0; j < 3; j++) {
            rlc(&t, w[j], s[j], beta, gamma);
            if (!fr_invert(&t, &t)) { rc = -1; break; }
            fr_mul(&cur, &cur, &t);
        }
    }
    if (rc == 0 && !fr_eq(&cur, &one)) rc = -2;
    free(roots);
    return rc;
}
