/*
 * field.c -- Fr (scalar.rs) and Fp (fp.rs) restated in plain C.
 * TEST INFRASTRUCTURE ONLY (see bp_oracle.h).  Citations: /root/reference/lib/bls12_381/src/.
 *
 * Both fields are word-serial Montgomery arithmetic on 64-bit limbs with a
 * final conditional subtraction, so every value is the unique fully reduced
 * representative -- the property that makes "bit-exact" well defined.
 */
#include "bp_oracle.h"
#include <string.h>

typedef unsigned __int128 u128;

/* util.rs:3-20 -- adc / sbb / mac */
static inline uint64_t adc(uint64_t a, uint64_t b, uint64_t *carry) {
    u128 t = (u128)a + b + *carry;
    *carry = (uint64_t)(t >> 64);
    return (uint64_t)t;
}
static inline uint64_t sbb(uint64_t a, uint64_t b, uint64_t *borrow) {
    u128 t = (u128)a - ((u128)b + (*borrow >> 63));
    *borrow = (uint64_t)(t >> 64);
    return (uint64_t)t;
}
static inline uint64_t mac(uint64_t a, uint64_t b, uint64_t c, uint64_t *carry) {
    u128 t = (u128)a + (u128)b * c + *carry;
    *carry = (uint64_t)(t >> 64);
    return (uint64_t)t;
}

/* ------------------------------------------------------------------ */
/* generic n-limb helpers, n = 4 (Fr) or 6 (Fp)                         */
/* ------------------------------------------------------------------ */

/* r = a - m if a >= m else a   (scalar.rs:590-604 used as the final step, fp.rs:361-379) */
static void cond_sub(uint64_t *r, const uint64_t *a, const uint64_t *m, int n) {
    uint64_t t[6], borrow = 0;
    for (int i = 0; i < n; i++) t[i] = sbb(a[i], m[i], &borrow);
    uint64_t mask = borrow;  /* all ones when a < m */
    for (int i = 0; i < n; i++) r[i] = (a[i] & mask) | (t[i] & ~mask);
}

/* scalar.rs:608-617 / fp.rs:382-393: add then subtract the modulus once */
static void mod_add(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *m, int n) {
    uint64_t t[6], carry = 0;
    for (int i = 0; i < n; i++) t[i] = adc(a[i], b[i], &carry);
    /* both moduli leave a spare top bit, so carry == 0 here */
    cond_sub(r, t, m, n);
}

/* scalar.rs:590-604: subtract, add the modulus back when it borrowed */
static void mod_sub(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *m, int n) {
    uint64_t t[6], borrow = 0, carry = 0;
    for (int i = 0; i < n; i++) t[i] = sbb(a[i], b[i], &borrow);
    for (int i = 0; i < n; i++) r[i] = adc(t[i], m[i] & borrow, &carry);
}

/* scalar.rs:621-635 / fp.rs:396-418: m - a, masked to zero for a == 0 */
static void mod_neg(uint64_t *r, const uint64_t *a, const uint64_t *m, int n) {
    uint64_t borrow = 0, any = 0;
    for (int i = 0; i < n; i++) any |= a[i];
    uint64_t mask = any ? ~(uint64_t)0 : 0;
    for (int i = 0; i < n; i++) r[i] = sbb(m[i], a[i], &borrow) & mask;
}

/* scalar.rs:514-558 / fp.rs:487-562 -- HAC 14.32 on a 2n-limb value */
static void mont_reduce(uint64_t *r, uint64_t *t /* 2n limbs, clobbered */, const uint64_t *m,
                        uint64_t inv, int n) {
    uint64_t carry2 = 0;
    for (int i = 0; i < n; i++) {
        uint64_t k = t[i] * inv, carry = 0;
        (void)mac(t[i], k, m[0], &carry);
        for (int j = 1; j < n; j++) t[i + j] = mac(t[i + j], k, m[j], &carry);
        /* fold the row carry and the previous row's overflow into limb i+n */
        u128 s = (u128)t[i + n] + carry2 + carry;
        t[i + n] = (uint64_t)s;
        carry2 = (uint64_t)(s >> 64);
    }
    cond_sub(r, t + n, m, n);
}

/* scalar.rs:562-586 / fp.rs:565-609 -- schoolbook product then reduction */
static void mont_mul(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *m,
                     uint64_t inv, int n) {
    uint64_t t[12];
    memset(t, 0, sizeof t);
    for (int i = 0; i < n; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < n; j++) t[i + j] = mac(t[i + j], a[i], b[j], &carry);
        t[i + n] = carry;
    }
    mont_reduce(r, t, m, inv, n);
}

/* ------------------------------------------------------------------ */
/* Fr                                                                   */
/* ------------------------------------------------------------------ */
/* scalar.rs:83-88 */
const fr_t FR_MODULUS = {{0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull}};
/* scalar.rs:164 */
static const uint64_t FR_INV = 0xfffffffeffffffffull;
/* scalar.rs:167-188 */
const fr_t FR_R  = {{0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull, 0x1824b159acc5056full}};
const fr_t FR_R2 = {{0xc999e990f3f29c6dull, 0x2b6cedcb87925c23ull, 0x05d314967254398full, 0x0748d9d99f59ff11ull}};
const fr_t FR_R3 = {{0xc62c1807439b73afull, 0x1b3e0d188cf06990ull, 0x73d13c71c7b5f418ull, 0x6e2a5bb9c8db33e9ull}};
/* scalar.rs:208-221 */
const fr_t FR_ROOT_OF_UNITY     = {{0xb9b58d8c5f0e466aull, 0x5b1b4c801819d7ecull, 0x0af53ae352a31e64ull, 0x5bf3adda19e9b27bull}};
const fr_t FR_ROOT_OF_UNITY_INV = {{0x4256481adcf3219aull, 0x45f37b7f96b6cad3ull, 0xf9c3f1d75f7a3b27ull, 0x2d2fc049658afd43ull}};

void fr_zero(fr_t *r) { memset(r, 0, sizeof *r); }
void fr_one(fr_t *r) { *r = FR_R; }
int fr_eq(const fr_t *a, const fr_t *b) { return memcmp(a, b, sizeof *a) == 0; }
int fr_is_zero(const fr_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
void fr_add(fr_t *r, const fr_t *a, const fr_t *b) { mod_add(r->l, a->l, b->l, FR_MODULUS.l, 4); }
void fr_sub(fr_t *r, const fr_t *a, const fr_t *b) { mod_sub(r->l, a->l, b->l, FR_MODULUS.l, 4); }
void fr_neg(fr_t *r, const fr_t *a) { mod_neg(r->l, a->l, FR_MODULUS.l, 4); }
void fr_mul(fr_t *r, const fr_t *a, const fr_t *b) { mont_mul(r->l, a->l, b->l, FR_MODULUS.l, FR_INV, 4); }
void fr_square(fr_t *r, const fr_t *a) { fr_mul(r, a, a); }      /* scalar.rs:349-377 (same value) */
void fr_double(fr_t *r, const fr_t *a) { fr_add(r, a, a); }

/* scalar.rs:381-392 -- MSB-first square-and-multiply; the reference's version is
 * constant time (always multiplies, then selects) which yields the same value */
void fr_pow(fr_t *r, const fr_t *a, const uint64_t by[4]) {
    fr_t res = FR_R, base = *a;
    for (int i = 3; i >= 0; i--)
        for (int j = 63; j >= 0; j--) {
            fr_square(&res, &res);
            if ((by[i] >> j) & 1) fr_mul(&res, &res, &base);
        }
    *r = res;
}
/* scalar.rs:400-412 */
void fr_pow_vartime(fr_t *r, const fr_t *a, const uint64_t by[4]) { fr_pow(r, a, by); }

/* scalar.rs:416-511 -- the reference uses an addition chain for a^(q-2); the value is the same */
int fr_invert(fr_t *r, const fr_t *a) {
    static const uint64_t q_minus_2[4] = {0xfffffffeffffffffull, 0x53bda402fffe5bfeull,
                                          0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
    fr_pow(r, a, q_minus_2);
    return !fr_is_zero(a);
}
/* scalar.rs:343-345 */
void fr_from_raw(fr_t *r, const uint64_t v[4]) {
    fr_t t = {{v[0], v[1], v[2], v[3]}};
    fr_mul(r, &t, &FR_R2);
}
/* scalar.rs:48-52 */
void fr_from_u64(fr_t *r, uint64_t v) {
    uint64_t t[4] = {v, 0, 0, 0};
    fr_from_raw(r, t);
}
/* scalar.rs:264-288 -- little-endian, rejects values >= q */
int fr_from_bytes(fr_t *r, const uint8_t b[32]) {
    fr_t t;
    for (int i = 0; i < 4; i++) {
        uint64_t w = 0;
        for (int k = 7; k >= 0; k--) w = (w << 8) | b[8 * i + k];
        t.l[i] = w;
    }
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) (void)sbb(t.l[i], FR_MODULUS.l[i], &borrow);
    fr_mul(r, &t, &FR_R2);
    return (int)(borrow & 1);
}
/* scalar.rs:292-304 */
void fr_to_bytes(uint8_t b[32], const fr_t *a) {
    uint64_t t[8] = {a->l[0], a->l[1], a->l[2], a->l[3], 0, 0, 0, 0};
    fr_t c;
    mont_reduce(c.l, t, FR_MODULUS.l, FR_INV, 4);
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 8; k++) b[8 * i + k] = (uint8_t)(c.l[i] >> (8 * k));
}
/* scalar.rs:323-339 */
void fr_from_u512(fr_t *r, const uint64_t limbs[8]) {
    fr_t d0 = {{limbs[0], limbs[1], limbs[2], limbs[3]}};
    fr_t d1 = {{limbs[4], limbs[5], limbs[6], limbs[7]}};
    fr_mul(&d0, &d0, &FR_R2);
    fr_mul(&d1, &d1, &FR_R3);
    fr_add(r, &d0, &d1);
}
/* scalar.rs:308-321 */
void fr_from_bytes_wide(fr_t *r, const uint8_t b[64]) {
    uint64_t limbs[8];
    for (int i = 0; i < 8; i++) {
        uint64_t w = 0;
        for (int k = 7; k >= 0; k--) w = (w << 8) | b[8 * i + k];
        limbs[i] = w;
    }
    fr_from_u512(r, limbs);
}

/* ------------------------------------------------------------------ */
/* Fp                                                                   */
/* ------------------------------------------------------------------ */
/* fp.rs:70-77 */
static const uint64_t FP_MODULUS[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                                       0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
/* fp.rs:80 */
static const uint64_t FP_INV = 0x89f3fffcfffcfffdull;
/* fp.rs:83-110 */
const fp_t FP_R  = {{0x760900000002fffdull, 0xebf4000bc40c0002ull, 0x5f48985753c758baull,
                     0x77ce585370525745ull, 0x5c071a97a256ec6dull, 0x15f65ec3fa80e493ull}};
const fp_t FP_R2 = {{0xf4df1f341c341746ull, 0x0a76e6a609d104f1ull, 0x8de5476c4c95b6d5ull,
                     0x67eb88a9939d83c0ull, 0x9a793e85b519952dull, 0x11988fe592cae3aaull}};
const fp_t FP_R3 = {{0xed48ac6bd94ca1e0ull, 0x315f831e03a7adf8ull, 0x9a53352a615e29ddull,
                     0x34c04e5e921e1761ull, 0x2512d43565724728ull, 0x0aa6346091755d4dull}};
/* g1.rs:176-183 -- curve constant b = 4 */
const fp_t FP_B  = {{0xaa270000000cfff3ull, 0x53cc0032fc34000aull, 0x478fe97a6b0a807full,
                     0xb1d37ebee6ba24d7ull, 0x8ec9733bbf78ab2full, 0x09d645513d83de7eull}};

void fp_zero(fp_t *r) { memset(r, 0, sizeof *r); }
void fp_one(fp_t *r) { *r = FP_R; }
int fp_eq(const fp_t *a, const fp_t *b) { return memcmp(a, b, sizeof *a) == 0; }
int fp_is_zero(const fp_t *a) {
    return (a->l[0] | a->l[1] | a->l[2] | a->l[3] | a->l[4] | a->l[5]) == 0;
}
void fp_add(fp_t *r, const fp_t *a, const fp_t *b) { mod_add(r->l, a->l, b->l, FP_MODULUS, 6); }
void fp_neg(fp_t *r, const fp_t *a) { mod_neg(r->l, a->l, FP_MODULUS, 6); }
/* fp.rs:421-423: a - b = a + (-b) */
void fp_sub(fp_t *r, const fp_t *a, const fp_t *b) {
    fp_t nb;
    fp_neg(&nb, b);
    fp_add(r, a, &nb);
}
void fp_mul(fp_t *r, const fp_t *a, const fp_t *b) { mont_mul(r->l, a->l, b->l, FP_MODULUS, FP_INV, 6); }
void fp_square(fp_t *r, const fp_t *a) { fp_mul(r, a, a); }      /* fp.rs:613-660 (same value) */

/* fp.rs:300-318 pow_vartime: MSB-first */
void fp_pow_vartime(fp_t *r, const fp_t *a, const uint64_t by[6]) {
    fp_t res = FP_R, base = *a;
    for (int i = 5; i >= 0; i--)
        for (int j = 63; j >= 0; j--) {
            fp_square(&res, &res);
            if ((by[i] >> j) & 1) fp_mul(&res, &res, &base);
        }
    *r = res;
}
/* fp.rs:346-358 -- a^(p-2) */
int fp_invert(fp_t *r, const fp_t *a) {
    static const uint64_t e[6] = {0xb9feffffffffaaa9ull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                                  0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
    fp_pow_vartime(r, a, e);
    return !fp_is_zero(a);
}
/* fp.rs:324-340 -- a^((p+1)/4), valid because p = 3 mod 4; returns 1 iff a is a square */
int fp_sqrt(fp_t *r, const fp_t *a) {
    static const uint64_t e[6] = {0xee7fbfffffffeaabull, 0x07aaffffac54ffffull, 0xd9cc34a83dac3d89ull,
                                  0xd91dd2e13ce144afull, 0x92c6e9ed90d2eb35ull, 0x0680447a8e5ff9a6ull};
    fp_t s, s2;
    fp_pow_vartime(&s, a, e);
    fp_square(&s2, &s);
    *r = s;
    return fp_eq(&s2, a);
}
/* fp.rs:179-208 -- 48 bytes big-endian, rejects >= p */
int fp_from_bytes(fp_t *r, const uint8_t b[48]) {
    fp_t t;
    for (int i = 0; i < 6; i++) {
        uint64_t w = 0;
        for (int k = 0; k < 8; k++) w = (w << 8) | b[8 * (5 - i) + k];
        t.l[i] = w;
    }
    uint64_t borrow = 0;
    for (int i = 0; i < 6; i++) (void)sbb(t.l[i], FP_MODULUS[i], &borrow);
    fp_mul(r, &t, &FP_R2);
    return (int)(borrow & 1);
}
static void fp_canonical(uint64_t c[6], const fp_t *a) {
    uint64_t t[12];
    memset(t, 0, sizeof t);
    memcpy(t, a->l, sizeof a->l);
    mont_reduce(c, t, FP_MODULUS, FP_INV, 6);
}
/* fp.rs:212-227 */
void fp_to_bytes(uint8_t b[48], const fp_t *a) {
    uint64_t c[6];
    fp_canonical(c, a);
    for (int i = 0; i < 6; i++)
        for (int k = 0; k < 8; k++) b[8 * (5 - i) + k] = (uint8_t)(c[i] >> (8 * (7 - k)));
}
/* fp.rs:273-298 -- canonical value > (p-1)/2 */
int fp_lexicographically_largest(const fp_t *a) {
    static const uint64_t half_plus_1[6] = {0xdcff7fffffffd556ull, 0x0f55ffff58a9ffffull, 0xb39869507b587b12ull,
                                            0xb23ba5c279c2895full, 0x258dd3db21a5d66bull, 0x0d0088f51cbff34dull};
    uint64_t c[6], borrow = 0;
    fp_canonical(c, a);
    for (int i = 0; i < 6; i++) (void)sbb(c[i], half_plus_1[i], &borrow);
    return !(borrow & 1);
}
