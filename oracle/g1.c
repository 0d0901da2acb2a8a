/*
 * g1.c -- G1 group law and wire encodings (lib/bls12_381/src/g1.rs) restated in plain C.
 * TEST INFRASTRUCTURE ONLY (see bp_oracle.h).
 */
#include "bp_oracle.h"
#include <stdlib.h>
#include <string.h>

/* g1.rs:197-217 */
static const fp_t GEN_X = {{0x5cb38790fd530c16ull, 0x7817fc679976fff5ull, 0x154f95c7143ba1c1ull,
                            0xf0ae6acdf3d0e747ull, 0xedce6ecc21dbf440ull, 0x120177419e0bfb75ull}};
static const fp_t GEN_Y = {{0xbaac93d50ce72271ull, 0x8c22631a7918fd8eull, 0xdd595f13570725ceull,
                            0x51ac582950405194ull, 0x0e1c8c3fad0059c0ull, 0x0bbc3efc5008a26aull}};

/* g1.rs:186-192 */
void g1_affine_identity(g1_affine_t *r) {
    fp_zero(&r->x);
    fp_one(&r->y);
    r->infinity = 1;
}
void g1_affine_generator(g1_affine_t *r) {
    r->x = GEN_X;
    r->y = GEN_Y;
    r->infinity = 0;
}
/* g1.rs:605-611 */
void g1_identity(g1_proj_t *r) {
    fp_zero(&r->x);
    fp_one(&r->y);
    fp_zero(&r->z);
}
/* g1.rs:615-635 */
void g1_generator(g1_proj_t *r) {
    r->x = GEN_X;
    r->y = GEN_Y;
    fp_one(&r->z);
}
/* g1.rs:843-845 */
int g1_is_identity(const g1_proj_t *p) { return fp_is_zero(&p->z); }

/* g1.rs:849-855: Y^2 Z = X^3 + b Z^3, or identity */
int g1_is_on_curve(const g1_proj_t *p) {
    fp_t l, r, t;
    fp_square(&l, &p->y);
    fp_mul(&l, &l, &p->z);
    fp_square(&r, &p->x);
    fp_mul(&r, &r, &p->x);
    fp_square(&t, &p->z);
    fp_mul(&t, &t, &p->z);
    fp_mul(&t, &t, &FP_B);
    fp_add(&r, &r, &t);
    return fp_eq(&l, &r) || g1_is_identity(p);
}
/* g1.rs:479-496 -- equality of projective classes */
int g1_eq(const g1_proj_t *a, const g1_proj_t *b) {
    fp_t x1, x2, y1, y2;
    fp_mul(&x1, &a->x, &b->z);
    fp_mul(&x2, &b->x, &a->z);
    fp_mul(&y1, &a->y, &b->z);
    fp_mul(&y2, &b->y, &a->z);
    int az = fp_is_zero(&a->z), bz = fp_is_zero(&b->z);
    return (az && bz) || (!az && !bz && fp_eq(&x1, &x2) && fp_eq(&y1, &y2));
}
/* g1.rs:520-528 */
void g1_neg(g1_proj_t *r, const g1_proj_t *p) {
    r->x = p->x;
    fp_neg(&r->y, &p->y);
    r->z = p->z;
}
/* g1.rs:597-601 */
static void mul_by_3b(fp_t *r, const fp_t *a) {
    fp_t t, u;
    fp_add(&t, a, a);
    fp_add(&t, &t, &t);
    fp_add(&u, &t, &t);
    fp_add(r, &u, &t);
}
/* g1.rs:638-667 -- Renes-Costello-Batina 2015/1060 Algorithm 9 */
void g1_double(g1_proj_t *r, const g1_proj_t *p) {
    fp_t t0, t1, t2, x3, y3, z3;
    fp_square(&t0, &p->y);
    fp_add(&z3, &t0, &t0);
    fp_add(&z3, &z3, &z3);
    fp_add(&z3, &z3, &z3);
    fp_mul(&t1, &p->y, &p->z);
    fp_square(&t2, &p->z);
    mul_by_3b(&t2, &t2);
    fp_mul(&x3, &t2, &z3);
    fp_add(&y3, &t0, &t2);
    fp_mul(&z3, &t1, &z3);
    fp_add(&t1, &t2, &t2);
    fp_add(&t2, &t1, &t2);
    fp_sub(&t0, &t0, &t2);
    fp_mul(&y3, &t0, &y3);
    fp_add(&y3, &x3, &y3);
    fp_mul(&t1, &p->x, &p->y);
    fp_mul(&x3, &t0, &t1);
    fp_add(&x3, &x3, &x3);
    if (g1_is_identity(p)) {
        g1_identity(r);
    } else {
        r->x = x3;
        r->y = y3;
        r->z = z3;
    }
}
/* g1.rs:670-712 -- Algorithm 7 (complete) */
void g1_add(g1_proj_t *r, const g1_proj_t *a, const g1_proj_t *b) {
    fp_t t0, t1, t2, t3, t4, x3, y3, z3;
    fp_mul(&t0, &a->x, &b->x);
    fp_mul(&t1, &a->y, &b->y);
    fp_mul(&t2, &a->z, &b->z);
    fp_add(&t3, &a->x, &a->y);
    fp_add(&t4, &b->x, &b->y);
    fp_mul(&t3, &t3, &t4);
    fp_add(&t4, &t0, &t1);
    fp_sub(&t3, &t3, &t4);
    fp_add(&t4, &a->y, &a->z);
    fp_add(&x3, &b->y, &b->z);
    fp_mul(&t4, &t4, &x3);
    fp_add(&x3, &t1, &t2);
    fp_sub(&t4, &t4, &x3);
    fp_add(&x3, &a->x, &a->z);
    fp_add(&y3, &b->x, &b->z);
    fp_mul(&x3, &x3, &y3);
    fp_add(&y3, &t0, &t2);
    fp_sub(&y3, &x3, &y3);
    fp_add(&x3, &t0, &t0);
    fp_add(&t0, &x3, &t0);
    mul_by_3b(&t2, &t2);
    fp_add(&z3, &t1, &t2);
    fp_sub(&t1, &t1, &t2);
    mul_by_3b(&y3, &y3);
    fp_mul(&x3, &t4, &y3);
    fp_mul(&t2, &t3, &t1);
    fp_sub(&x3, &t2, &x3);
    fp_mul(&y3, &y3, &t0);
    fp_mul(&t1, &t1, &z3);
    fp_add(&y3, &t1, &y3);
    fp_mul(&t0, &t0, &t3);
    fp_mul(&z3, &z3, &t4);
    fp_add(&z3, &z3, &t0);
    r->x = x3;
    r->y = y3;
    r->z = z3;
}
/* g1.rs:715-752 -- Algorithm 8 (mixed) */
void g1_add_mixed(g1_proj_t *r, const g1_proj_t *a, const g1_affine_t *b) {
    fp_t t0, t1, t2, t3, t4, x3, y3, z3;
    if (b->infinity) {
        *r = *a;
        return;
    }
    fp_mul(&t0, &a->x, &b->x);
    fp_mul(&t1, &a->y, &b->y);
    fp_add(&t3, &b->x, &b->y);
    fp_add(&t4, &a->x, &a->y);
    fp_mul(&t3, &t3, &t4);
    fp_add(&t4, &t0, &t1);
    fp_sub(&t3, &t3, &t4);
    fp_mul(&t4, &b->y, &a->z);
    fp_add(&t4, &t4, &a->y);
    fp_mul(&y3, &b->x, &a->z);
    fp_add(&y3, &y3, &a->x);
    fp_add(&x3, &t0, &t0);
    fp_add(&t0, &x3, &t0);
    mul_by_3b(&t2, &a->z);
    fp_add(&z3, &t1, &t2);
    fp_sub(&t1, &t1, &t2);
    mul_by_3b(&y3, &y3);
    fp_mul(&x3, &t4, &y3);
    fp_mul(&t2, &t3, &t1);
    fp_sub(&x3, &t2, &x3);
    fp_mul(&y3, &y3, &t0);
    fp_mul(&t1, &t1, &z3);
    fp_add(&y3, &t1, &y3);
    fp_mul(&t0, &t0, &t3);
    fp_mul(&z3, &z3, &t4);
    fp_add(&z3, &z3, &t0);
    r->x = x3;
    r->y = y3;
    r->z = z3;
}
/* g1.rs:754-774 -- MSB-first double-and-add over the canonical bytes, top bit skipped */
void g1_mul(g1_proj_t *r, const g1_proj_t *p, const fr_t *by) {
    uint8_t bytes[32];
    g1_proj_t acc, t;
    fr_to_bytes(bytes, by);
    g1_identity(&acc);
    for (int bit = 254; bit >= 0; bit--) {
        g1_double(&acc, &acc);
        if ((bytes[bit >> 3] >> (bit & 7)) & 1) {
            g1_add(&t, &acc, p);
            acc = t;
        }
    }
    *r = acc;
}
/* g1.rs:49-63 */
void g1_to_affine(g1_affine_t *r, const g1_proj_t *p) {
    fp_t zinv;
    if (!fp_invert(&zinv, &p->z)) {
        g1_affine_identity(r);
        return;
    }
    fp_mul(&r->x, &p->x, &zinv);
    fp_mul(&r->y, &p->y, &zinv);
    r->infinity = 0;
}
/* g1.rs:34-47 */
void g1_from_affine(g1_proj_t *r, const g1_affine_t *p) {
    r->x = p->x;
    r->y = p->y;
    if (p->infinity) fp_zero(&r->z); else fp_one(&r->z);
}
/* g1.rs:806-839 -- one inversion for the whole batch */
void g1_batch_normalize(const g1_proj_t *p, g1_affine_t *q, size_t n) {
    fp_t acc = FP_R, tmp;
    for (size_t i = 0; i < n; i++) {
        q[i].x = acc;
        if (!g1_is_identity(&p[i])) fp_mul(&acc, &acc, &p[i].z);
    }
    fp_invert(&acc, &acc);
    for (size_t i = n; i-- > 0;) {
        int skip = g1_is_identity(&p[i]);
        fp_mul(&tmp, &q[i].x, &acc);
        if (!skip) fp_mul(&acc, &acc, &p[i].z);
        if (skip) {
            g1_affine_identity(&q[i]);
        } else {
            fp_mul(&q[i].x, &p[i].x, &tmp);
            fp_mul(&q[i].y, &p[i].y, &tmp);
            q[i].infinity = 0;
        }
    }
}
/* g1.rs:246-260 */
void g1_to_uncompressed(uint8_t out[96], const g1_affine_t *p) {
    fp_t z;
    fp_zero(&z);
    fp_to_bytes(out, p->infinity ? &z : &p->x);
    fp_to_bytes(out + 48, p->infinity ? &z : &p->y);
    if (p->infinity) out[0] |= 1u << 6;
}
/* g1.rs:273-322 */
int g1_from_uncompressed_unchecked(g1_affine_t *r, const uint8_t in[96]) {
    int compression = (in[0] >> 7) & 1, infinity = (in[0] >> 6) & 1, sort = (in[0] >> 5) & 1;
    uint8_t tmp[48];
    fp_t x, y;
    memcpy(tmp, in, 48);
    tmp[0] &= 0x1f;
    if (!fp_from_bytes(&x, tmp)) return 0;
    if (!fp_from_bytes(&y, in + 48)) return 0;
    if (infinity) {
        g1_affine_identity(r);
    } else {
        r->x = x;
        r->y = y;
        r->infinity = 0;
    }
    return (!infinity || (fp_is_zero(&x) && fp_is_zero(&y))) && !compression && !sort;
}
/* g1.rs:221-242 */
void g1_to_compressed(uint8_t out[48], const g1_affine_t *p) {
    fp_t z;
    fp_zero(&z);
    fp_to_bytes(out, p->infinity ? &z : &p->x);
    out[0] |= 1u << 7;
    if (p->infinity) out[0] |= 1u << 6;
    if (!p->infinity && fp_lexicographically_largest(&p->y)) out[0] |= 1u << 5;
}
/* g1.rs:337-390 -- y = sqrt(x^3 + 4), sign picked by the sort flag */
int g1_from_compressed_unchecked(g1_affine_t *r, const uint8_t in[48]) {
    int compression = (in[0] >> 7) & 1, infinity = (in[0] >> 6) & 1, sort = (in[0] >> 5) & 1;
    uint8_t tmp[48];
    fp_t x, y, rhs, ny;
    memcpy(tmp, in, 48);
    tmp[0] &= 0x1f;
    if (!fp_from_bytes(&x, tmp)) return 0;
    if (infinity) {
        g1_affine_identity(r);
        return compression && !sort && fp_is_zero(&x);
    }
    fp_square(&rhs, &x);
    fp_mul(&rhs, &rhs, &x);
    fp_add(&rhs, &rhs, &FP_B);
    if (!fp_sqrt(&y, &rhs)) return 0;
    fp_neg(&ny, &y);
    if (fp_lexicographically_largest(&y) != sort) y = ny;
    r->x = x;
    r->y = y;
    r->infinity = 0;
    return compression;
}
