/*
 * bp_oracle.h -- CPU restatement (plain C) of the reference's MSM + Fr-DFT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity checker and the `cpu_baseline`
 * of bench.py.  Nothing under oracle/ is linked, imported or executed by the
 * product path (baby_plonk_rust_amd/): only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it.
 *
 * Pinning: every function cites the reference file:line it restates
 * (paths relative to /root/reference).  The restatement is pinned against the
 * reference's own known-answer tests and fixtures (tests/test_oracle_*.py):
 *   - Fp literal KATs        lib/bls12_381/src/fp.rs:700-979
 *   - Fr KATs                lib/bls12_381/src/scalar.rs:794-1258
 *   - G1 KATs                lib/bls12_381/src/g1.rs:1263-1297,1372-1417
 *   - 1000-point wire vectors lib/bls12_381/src/tests/ g1_{un,}compressed .dat (copied as data to tests/golden/)
 *   - MSM / SRS identities   src/setup.rs:46-116
 * The reference itself is Rust and no Rust toolchain exists in this image, so
 * there is no oracle/_ref build (unbuildable here).
 */
#ifndef BP_ORACLE_H
#define BP_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* scalar.rs:16-22  -- 4 x u64 little-endian limbs, always Montgomery form, R = 2^256 */
typedef struct { uint64_t l[4]; } fr_t;
/* fp.rs:11-15      -- 6 x u64 little-endian limbs, Montgomery form, R = 2^384 */
typedef struct { uint64_t l[6]; } fp_t;
/* g1.rs:28-32 */
typedef struct { fp_t x, y; uint64_t infinity; } g1_affine_t;
/* g1.rs:442-446    -- homogeneous projective */
typedef struct { fp_t x, y, z; } g1_proj_t;

/* ---- Fr (scalar.rs) ---- */
extern const fr_t FR_MODULUS, FR_R, FR_R2, FR_R3, FR_ROOT_OF_UNITY, FR_ROOT_OF_UNITY_INV;
void fr_zero(fr_t *r);
void fr_one(fr_t *r);
int  fr_eq(const fr_t *a, const fr_t *b);
int  fr_is_zero(const fr_t *a);
void fr_add(fr_t *r, const fr_t *a, const fr_t *b);
void fr_sub(fr_t *r, const fr_t *a, const fr_t *b);
void fr_neg(fr_t *r, const fr_t *a);
void fr_mul(fr_t *r, const fr_t *a, const fr_t *b);
void fr_square(fr_t *r, const fr_t *a);
void fr_double(fr_t *r, const fr_t *a);
void fr_pow(fr_t *r, const fr_t *a, const uint64_t by[4]);
void fr_pow_vartime(fr_t *r, const fr_t *a, const uint64_t by[4]);
int  fr_invert(fr_t *r, const fr_t *a);                  /* 1 = ok, 0 = zero input */
void fr_from_u64(fr_t *r, uint64_t v);
void fr_from_raw(fr_t *r, const uint64_t v[4]);
int  fr_from_bytes(fr_t *r, const uint8_t b[32]);        /* 1 = canonical */
void fr_to_bytes(uint8_t b[32], const fr_t *a);
void fr_from_bytes_wide(fr_t *r, const uint8_t b[64]);
void fr_from_u512(fr_t *r, const uint64_t limbs[8]);

/* ---- Fp (fp.rs) ---- */
extern const fp_t FP_R, FP_R2, FP_R3, FP_B;
void fp_zero(fp_t *r);
void fp_one(fp_t *r);
int  fp_eq(const fp_t *a, const fp_t *b);
int  fp_is_zero(const fp_t *a);
void fp_add(fp_t *r, const fp_t *a, const fp_t *b);
void fp_sub(fp_t *r, const fp_t *a, const fp_t *b);
void fp_neg(fp_t *r, const fp_t *a);
void fp_mul(fp_t *r, const fp_t *a, const fp_t *b);
void fp_square(fp_t *r, const fp_t *a);
void fp_pow_vartime(fp_t *r, const fp_t *a, const uint64_t by[6]);
int  fp_invert(fp_t *r, const fp_t *a);
int  fp_sqrt(fp_t *r, const fp_t *a);
int  fp_from_bytes(fp_t *r, const uint8_t b[48]);        /* big-endian, 1 = canonical */
void fp_to_bytes(uint8_t b[48], const fp_t *a);
int  fp_lexicographically_largest(const fp_t *a);

/* ---- G1 (g1.rs) ---- */
void g1_affine_identity(g1_affine_t *r);
void g1_affine_generator(g1_affine_t *r);
void g1_identity(g1_proj_t *r);
void g1_generator(g1_proj_t *r);
int  g1_is_identity(const g1_proj_t *p);
int  g1_is_on_curve(const g1_proj_t *p);
int  g1_eq(const g1_proj_t *a, const g1_proj_t *b);
void g1_neg(g1_proj_t *r, const g1_proj_t *p);
void g1_double(g1_proj_t *r, const g1_proj_t *p);
void g1_add(g1_proj_t *r, const g1_proj_t *a, const g1_proj_t *b);
void g1_add_mixed(g1_proj_t *r, const g1_proj_t *a, const g1_affine_t *b);
void g1_mul(g1_proj_t *r, const g1_proj_t *p, const fr_t *by);
void g1_to_affine(g1_affine_t *r, const g1_proj_t *p);
void g1_from_affine(g1_proj_t *r, const g1_affine_t *p);
void g1_batch_normalize(const g1_proj_t *p, g1_affine_t *q, size_t n);
void g1_to_uncompressed(uint8_t out[96], const g1_affine_t *p);
int  g1_from_uncompressed_unchecked(g1_affine_t *r, const uint8_t in[96]);
void g1_to_compressed(uint8_t out[48], const g1_affine_t *p);
int  g1_from_compressed_unchecked(g1_affine_t *r, const uint8_t in[48]);

/* ---- MSM (src/msm.rs) ---- */
uint64_t msm_get_c_bit_chunk(const fr_t *scalar, size_t chunk_index, size_t chunk_size);
void msm_c_bit_msm(g1_proj_t *r, const g1_proj_t *points, const uint64_t *digits, size_t n, size_t c);
void msm_bucket_msm(g1_proj_t *r, const g1_proj_t *points, size_t n_points,
                    const fr_t *scalars, size_t n_scalars, size_t b, size_t c);
/* same algorithm, the k independent windows spread over `threads` OpenMP threads */
void msm_bucket_msm_mt(g1_proj_t *r, const g1_proj_t *points, size_t n_points,
                       const fr_t *scalars, size_t n_scalars, size_t b, size_t c, int threads);

/* ---- DFT (src/utils.rs) ---- */
void ntt_root_of_unity(fr_t *r, uint64_t group_order);
void ntt_roots_of_unity(fr_t *out, uint64_t group_order);
size_t ntt_find_next_power_of_two(size_t n, size_t m);
int  ntt_381(fr_t *out, const fr_t *in, size_t n);       /* faithful O(n^2); 0 ok, -1 not pow2 */
int  i_ntt_381(fr_t *out, const fr_t *in, size_t n);     /* faithful O(n^2) */
int  ntt_fast(fr_t *data, size_t n, int inverse);        /* O(n log n), identical output, in place */
int  ntt_fast_mt(fr_t *data, size_t n, int inverse, int threads);

/* ---- Polynomial (src/polynomial.rs); basis: 0 = Lagrange, 1 = Monomial ---- */
void poly_coeffs_evaluate(fr_t *r, const fr_t *coeffs, size_t n, const fr_t *x);
void poly_coeffs_evaluate_fast(fr_t *r, const fr_t *coeffs, size_t n, const fr_t *x);
void poly_shift_left(fr_t *out, const fr_t *in, size_t len, size_t n);
void poly_add_scalar(fr_t *out, const fr_t *a, size_t n, const fr_t *s, int basis);
void poly_sub_scalar(fr_t *out, const fr_t *a, size_t n, const fr_t *s, int basis);
void poly_mul_scalar(fr_t *out, const fr_t *a, size_t n, const fr_t *s);
size_t poly_add(fr_t *out, const fr_t *a, size_t na, const fr_t *b, size_t nb, int basis);
size_t poly_sub(fr_t *out, const fr_t *a, size_t na, const fr_t *b, size_t nb, int basis);
/* Monomial product: faithful (coeffs_evaluate per root + i_ntt_381) and fast (same output) */
size_t poly_mul(fr_t *out, const fr_t *a, size_t na, const fr_t *b, size_t nb);
size_t poly_mul_fast(fr_t *out, const fr_t *a, size_t na, const fr_t *b, size_t nb);
/* Division as written in the reference (quotient only, incl. its zero-coefficient quirk).
 * returns quotient length, or (size_t)-1 where the reference panics. out needs na slots. */
size_t poly_div(fr_t *out, const fr_t *a, size_t na, const fr_t *b, size_t nb);
/* src/prover.rs:279-319 round-2 grand product (Lagrange values of z, n entries) */
int prover_round2_z(fr_t *z_out, const fr_t *a, const fr_t *b, const fr_t *c, const fr_t *s1, const fr_t *s2,
                    const fr_t *s3, size_t n, const fr_t *beta, const fr_t *gamma, const fr_t *k1, const fr_t *k2);

/* ---- helpers for tests / bench (not in the reference) ---- */
void oracle_splitmix_scalars(fr_t *out, size_t n, uint64_t seed);     /* from_bytes_wide of a counter PRNG */
void oracle_points_progression(g1_affine_t *out, size_t n, const fr_t *a, const fr_t *d);
void oracle_points_to_bytes96(uint8_t *out, const g1_affine_t *p, size_t n);
void oracle_proj_from_bytes96(g1_proj_t *out, const uint8_t *in, size_t n);
void oracle_dot_progression(fr_t *r, const fr_t *s, size_t n, const fr_t *a, const fr_t *d);
double oracle_now(void);

#ifdef __cplusplus
}
#endif
#endif
