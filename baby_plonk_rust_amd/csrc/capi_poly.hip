// capi_poly.hip -- C ABI, part 4: Polynomial and its operators (polynomial.rs:14-380), host and device-resident forms, the
// grand product and Setup::commit of a polynomial.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "ctx.hpp"

#include "capi_common.hpp"

using namespace bp;
// ---------------------------------------------------------------------------------------------- Polynomial
int bp_poly_evaluate(bp_ctx* ctx, const void* coeffs, size_t n, int basis, const void* x32, int scalar_fmt, void* out32) {
  if (!ctx || !x32 || !out32 || !fmt_ok(scalar_fmt) || !basis_ok(basis) || (n && !coeffs)) return BP_ERR_INVALID_ARG;
  if (basis != BP_BASIS_MONOMIAL) return fail(ctx, BP_ERR_BASIS, "coeffs_evaluate needs the Monomial basis", hipSuccess, __FILE__, __LINE__);
  fr_t x, r;
  if (!fr_bytes_to_mont(x, (const uint8_t*)x32, scalar_fmt)) return fail(ctx, BP_ERR_BAD_SCALAR, "scalar >= q", hipSuccess, __FILE__, __LINE__);
  DeviceGuard guard(ctx->device);
  fr_t* d;
  BP_TRY(upload_fr(ctx, "io.poly_a", coeffs, n, n, scalar_fmt, &d));
  BP_TRY(poly_eval_run(ctx, d, n, x, &r));
  fr_mont_to_bytes((uint8_t*)out32, r, scalar_fmt);
  return BP_OK;
}

static int poly_addsub(bp_ctx* ctx, const void* a, size_t na, const void* b, size_t nb, int basis, int fmt, void* out, size_t* n_out,
                       int op) {
  if (!ctx || !n_out || !fmt_ok(fmt) || !basis_ok(basis) || (na && !a) || (nb && !b)) return BP_ERR_INVALID_ARG;
  if (basis == BP_BASIS_LAGRANGE && na != nb)
    return fail(ctx, BP_ERR_LENGTH, "Polynomials must have the same length", hipSuccess, __FILE__, __LINE__);
  const size_t n = std::max(na, nb);
  *n_out = n;
  if (n == 0) return BP_OK;
  if (!out) return BP_ERR_INVALID_ARG;
  DeviceGuard guard(ctx->device);
  fr_t *da, *db, *dout;
  // add/sub commute with the Montgomery map, so canonical inputs need no conversion at all
  BP_TRY(upload_fr(ctx, "io.poly_a", a, na, na, BP_FR_MONT, &da));
  BP_TRY(upload_fr(ctx, "io.poly_b", b, nb, nb, BP_FR_MONT, &db));
  BP_TRY(ws_get(ctx, "io.poly_out", n * sizeof(fr_t), (void**)&dout));
  BP_TRY(fr_binary_run(ctx, da, na, db, nb, dout, n, op));
  return download_fr(ctx, dout, out, n, BP_FR_MONT);
}
int bp_poly_add(bp_ctx* ctx, const void* a, size_t na, const void* b, size_t nb, int basis, int scalar_fmt, void* out, size_t* n_out) {
  return poly_addsub(ctx, a, na, b, nb, basis, scalar_fmt, out, n_out, 0);
}
int bp_poly_sub(bp_ctx* ctx, const void* a, size_t na, const void* b, size_t nb, int basis, int scalar_fmt, void* out, size_t* n_out) {
  return poly_addsub(ctx, a, na, b, nb, basis, scalar_fmt, out, n_out, 1);
}

int bp_poly_scalar_op(bp_ctx* ctx, const void* a, size_t n, int basis, const void* s32, int op, int scalar_fmt, void* out) {
  if (!ctx || !s32 || !fmt_ok(scalar_fmt) || !basis_ok(basis) || op < 0 || op > 2 || (n && (!a || !out))) return BP_ERR_INVALID_ARG;
  // Monomial Add/Sub<Scalar> index values[0] (polynomial.rs:62,123): an empty polynomial panics there
  if (n == 0) return op == 2 || basis == BP_BASIS_LAGRANGE ? BP_OK
                                                           : fail(ctx, BP_ERR_INVALID_ARG, "empty polynomial", hipSuccess, __FILE__, __LINE__);
  fr_t s;
  if (!fr_bytes_to_mont(s, (const uint8_t*)s32, scalar_fmt)) return fail(ctx, BP_ERR_BAD_SCALAR, "scalar >= q", hipSuccess, __FILE__, __LINE__);
  DeviceGuard guard(ctx->device);
  fr_t *da, *dout;
  BP_TRY(upload_fr(ctx, "io.poly_a", a, n, n, scalar_fmt, &da));
  BP_TRY(ws_get(ctx, "io.poly_out", n * sizeof(fr_t), (void**)&dout));
  if (op == 2) {
    BP_TRY(fr_scalar_run(ctx, da, s, dout, n, 2));
  } else if (basis == BP_BASIS_MONOMIAL) {
    BP_HIP(ctx, hipMemcpyAsync(dout, da, n * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
    BP_TRY(fr_scalar_run(ctx, da, s, dout, 1, op));          // values[0] += / -= rhs
  } else {
    BP_TRY(fr_scalar_run(ctx, da, s, dout, n, 0));           // Lagrange: += rhs for Add AND Sub (polynomial.rs:126-128)
  }
  return download_fr(ctx, dout, out, n, scalar_fmt);
}

int bp_poly_mul(bp_ctx* ctx, const void* a, size_t na, const void* b, size_t nb, int basis, int scalar_fmt, void* out, size_t* n_out) {
  if (!ctx || !n_out || !fmt_ok(scalar_fmt) || !basis_ok(basis) || !a || !b || !out) return BP_ERR_INVALID_ARG;
  if (basis != BP_BASIS_MONOMIAL) return fail(ctx, BP_ERR_BASIS, "Polynomial * Polynomial: Lagrange basis is todo!() in the reference", hipSuccess, __FILE__, __LINE__);
  if (na == 0 || nb == 0) return fail(ctx, BP_ERR_INVALID_ARG, "empty polynomial (len - 1 underflows, polynomial.rs:248-249)", hipSuccess, __FILE__, __LINE__);
  // find_next_power_of_two(n, m) with n = na-1, m = nb-1: smallest power of two >= n + m + 1 (utils.rs:54-61)
  const size_t target = na + nb - 1;
  uint32_t k = 0;
  while (((size_t)1 << k) < target) k++;
  if (k > 28) return fail(ctx, BP_ERR_TOO_LARGE, "product too long", hipSuccess, __FILE__, __LINE__);
  const size_t N = (size_t)1 << k;
  DeviceGuard guard(ctx->device);
  fr_t* d;
  BP_TRY(ws_get(ctx, "io.poly_mul", 2 * N * sizeof(fr_t), (void**)&d));
  BP_HIP(ctx, hipMemsetAsync(d, 0, 2 * N * sizeof(fr_t), ctx->stream));
  BP_HIP(ctx, hipMemcpyAsync(d, a, na * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
  BP_HIP(ctx, hipMemcpyAsync(d + N, b, nb * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
  if (scalar_fmt == BP_FR_BYTES_LE) BP_TRY(fr_convert_run(ctx, d, 2 * N, 0));
  BP_TRY(ntt_run(ctx, d, k, 0, 2, N));                        // evaluate both at the N roots (polynomial.rs:255-260)
  BP_TRY(fr_binary_run(ctx, d, N, d + N, N, d, N, 2));        // pointwise product (:262-266)
  BP_TRY(ntt_run(ctx, d, k, 1, 1, N));                        // i_ntt_381 (:270)
  *n_out = target;                                            // [0 ..= n+m] (:272)
  return download_fr(ctx, d, out, target, scalar_fmt);
}

int bp_poly_div(bp_ctx* ctx, const void* a, size_t na, const void* b, size_t nb, int basis, int scalar_fmt, void* out, size_t* n_out) {
  if (!ctx || !n_out || !fmt_ok(scalar_fmt) || !basis_ok(basis) || (na && !a) || (nb && !b)) return BP_ERR_INVALID_ARG;
  if (basis != BP_BASIS_MONOMIAL) return fail(ctx, BP_ERR_BASIS, "Div needs the Monomial basis (polynomial.rs:319)", hipSuccess, __FILE__, __LINE__);
  const fr_t* ha = (const fr_t*)a;
  const fr_t* hb = (const fr_t*)b;
  while (na > 0 && big_is_zero(ha[na - 1])) na--;             // polynomial.rs:325-339 (zero is all-zero in both formats)
  while (nb > 0 && big_is_zero(hb[nb - 1])) nb--;
  if (nb == 0) return fail(ctx, BP_ERR_DIV_ZERO, "division by the zero polynomial", hipSuccess, __FILE__, __LINE__);
  *n_out = 0;
  if (na < nb) return BP_OK;
  if (!out) return BP_ERR_INVALID_ARG;
  const size_t nq = na - nb + 1;
  DeviceGuard guard(ctx->device);
  fr_t *da, *db, *dq;
  BP_TRY(upload_fr(ctx, "io.poly_a", a, na, na, scalar_fmt, &da));
  BP_TRY(upload_fr(ctx, "io.poly_b", b, nb, nb, scalar_fmt, &db));
  BP_TRY(ws_get(ctx, "io.poly_out", nq * sizeof(fr_t), (void**)&dq));
  fr_t b0, b_lead;
  if (!fr_bytes_to_mont(b0, (const uint8_t*)&hb[0], scalar_fmt) || !fr_bytes_to_mont(b_lead, (const uint8_t*)&hb[nb - 1], scalar_fmt))
    return fail(ctx, BP_ERR_BAD_SCALAR, "scalar >= q", hipSuccess, __FILE__, __LINE__);
  bool binomial = nb >= 2;
  for (size_t i = 1; i + 1 < nb && binomial; i++) binomial = big_is_zero(hb[i]);
  BP_TRY(poly_div_run(ctx, da, na, db, nb, b0, b_lead, binomial, dq, nq));
  std::vector<fr_t> q(nq);
  BP_TRY(download_fr(ctx, dq, q.data(), nq, scalar_fmt));
  // The reference inserts one quotient coefficient per loop turn and pops every newly zero leading remainder
  // term (polynomial.rs:371-376): its result is the true quotient with the zero coefficients squeezed out.
  fr_t* o = (fr_t*)out;
  size_t m = 0;
  for (size_t i = 0; i < nq; i++)
    if (!big_is_zero(q[i])) o[m++] = q[i];
  *n_out = m;
  return BP_OK;
}

// ---------------------------------------------------------------------------------------------- device-resident
// The same operators on HBM-resident Montgomery data (the quotient / linearisation
// pipeline without PCIe round trips).  Length rules and quirks are those of the host-pointer entry points.
static int poly_addsub_device(bp_ctx* ctx, const void* a, size_t na, const void* b, size_t nb, int basis, void* out, size_t* n_out, int op) {
  if (!ctx || !n_out || !basis_ok(basis) || (na && !a) || (nb && !b)) return BP_ERR_INVALID_ARG;
  if (basis == BP_BASIS_LAGRANGE && na != nb) return fail(ctx, BP_ERR_LENGTH, "Polynomials must have the same length", hipSuccess, __FILE__, __LINE__);
  const size_t n = std::max(na, nb);
  *n_out = n;
  if (n == 0) return BP_OK;
  if (!out) return BP_ERR_INVALID_ARG;
  DeviceGuard guard(ctx->device);
  BP_TRY(fr_binary_run(ctx, (const fr_t*)a, na, (const fr_t*)b, nb, (fr_t*)out, n, op));
  BP_HIP(ctx, stream_wait(ctx->stream));
  return BP_OK;
}
int bp_poly_add_device(bp_ctx* ctx, const void* d_a, size_t na, const void* d_b, size_t nb, int basis, void* d_out, size_t* n_out) {
  return poly_addsub_device(ctx, d_a, na, d_b, nb, basis, d_out, n_out, 0);
}
int bp_poly_sub_device(bp_ctx* ctx, const void* d_a, size_t na, const void* d_b, size_t nb, int basis, void* d_out, size_t* n_out) {
  return poly_addsub_device(ctx, d_a, na, d_b, nb, basis, d_out, n_out, 1);
}
int bp_poly_scalar_op_device(bp_ctx* ctx, const void* d_a, size_t n, int basis, const void* s32_mont, int op, void* d_out) {
  if (!ctx || !s32_mont || !basis_ok(basis) || op < 0 || op > 2 || (n && (!d_a || !d_out))) return BP_ERR_INVALID_ARG;
  if (n == 0) return op == 2 || basis == BP_BASIS_LAGRANGE ? BP_OK : fail(ctx, BP_ERR_INVALID_ARG, "empty polynomial", hipSuccess, __FILE__, __LINE__);
  fr_t s;
  memcpy(&s, s32_mont, 32);
  DeviceGuard guard(ctx->device);
  const fr_t* a = (const fr_t*)d_a;
  fr_t* out = (fr_t*)d_out;
  if (op == 2) {
    BP_TRY(fr_scalar_run(ctx, a, s, out, n, 2));
  } else if (basis == BP_BASIS_MONOMIAL) {
    if (out != a) BP_HIP(ctx, hipMemcpyAsync(out, a, n * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
    BP_TRY(fr_scalar_run(ctx, a, s, out, 1, op));
  } else {
    BP_TRY(fr_scalar_run(ctx, a, s, out, n, 0));
  }
  BP_HIP(ctx, stream_wait(ctx->stream));
  return BP_OK;
}
int bp_poly_mul_device(bp_ctx* ctx, const void* d_a, size_t na, const void* d_b, size_t nb, int basis, void* d_out, size_t* n_out) {
  if (!ctx || !n_out || !basis_ok(basis) || !d_a || !d_b || !d_out) return BP_ERR_INVALID_ARG;
  if (basis != BP_BASIS_MONOMIAL) return fail(ctx, BP_ERR_BASIS, "Polynomial * Polynomial: Lagrange basis is todo!() in the reference", hipSuccess, __FILE__, __LINE__);
  if (na == 0 || nb == 0) return fail(ctx, BP_ERR_INVALID_ARG, "empty polynomial", hipSuccess, __FILE__, __LINE__);
  const size_t target = na + nb - 1;
  uint32_t k = 0;
  while (((size_t)1 << k) < target) k++;
  if (k > 28) return fail(ctx, BP_ERR_TOO_LARGE, "product too long", hipSuccess, __FILE__, __LINE__);
  const size_t N = (size_t)1 << k;
  DeviceGuard guard(ctx->device);
  fr_t* d;
  BP_TRY(ws_get(ctx, "io.poly_mul", 2 * N * sizeof(fr_t), (void**)&d));
  BP_HIP(ctx, hipMemsetAsync(d, 0, 2 * N * sizeof(fr_t), ctx->stream));
  BP_HIP(ctx, hipMemcpyAsync(d, d_a, na * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
  BP_HIP(ctx, hipMemcpyAsync(d + N, d_b, nb * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
  BP_TRY(ntt_run(ctx, d, k, 0, 2, N));
  BP_TRY(fr_binary_run(ctx, d, N, d + N, N, d, N, 2));
  BP_TRY(ntt_run(ctx, d, k, 1, 1, N));
  BP_HIP(ctx, hipMemcpyAsync(d_out, d, target * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
  BP_HIP(ctx, stream_wait(ctx->stream));
  *n_out = target;
  return BP_OK;
}
int bp_poly_div_device(bp_ctx* ctx, const void* d_a, size_t na, const void* d_b, size_t nb, int basis, void* d_out, size_t* n_out) {
  if (!ctx || !n_out || !basis_ok(basis) || (na && !d_a) || (nb && !d_b)) return BP_ERR_INVALID_ARG;
  if (basis != BP_BASIS_MONOMIAL) return fail(ctx, BP_ERR_BASIS, "Div needs the Monomial basis (polynomial.rs:319)", hipSuccess, __FILE__, __LINE__);
  DeviceGuard guard(ctx->device);
  size_t na_eff, nb_eff, dummy, mid_nonzero = 0;
  BP_TRY(fr_nonzero_stats_run(ctx, (const fr_t*)d_a, na, 0, 0, &na_eff, &dummy));           // trailing zeros trimmed (:325-339)
  BP_TRY(fr_nonzero_stats_run(ctx, (const fr_t*)d_b, nb, 0, 0, &nb_eff, &dummy));
  if (nb_eff == 0) return fail(ctx, BP_ERR_DIV_ZERO, "division by the zero polynomial", hipSuccess, __FILE__, __LINE__);
  *n_out = 0;
  if (na_eff < nb_eff) return BP_OK;
  if (!d_out) return BP_ERR_INVALID_ARG;
  if (nb_eff > 2) BP_TRY(fr_nonzero_stats_run(ctx, (const fr_t*)d_b, nb_eff, 1, nb_eff - 1, &dummy, &mid_nonzero));
  const bool binomial = nb_eff >= 2 && mid_nonzero == 0;
  fr_t ends[2];
  BP_HIP(ctx, hipMemcpyAsync(&ends[0], d_b, sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
  BP_HIP(ctx, hipMemcpyAsync(&ends[1], (const fr_t*)d_b + (nb_eff - 1), sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
  BP_HIP(ctx, stream_wait(ctx->stream));
  const size_t nq = na_eff - nb_eff + 1;
  fr_t *work, *q;
  BP_TRY(ws_get(ctx, "io.poly_div_work", na_eff * sizeof(fr_t), (void**)&work));       // the general path clobbers its dividend
  BP_TRY(ws_get(ctx, "io.poly_out", nq * sizeof(fr_t), (void**)&q));
  BP_HIP(ctx, hipMemcpyAsync(work, d_a, na_eff * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
  BP_TRY(poly_div_run(ctx, work, na_eff, (const fr_t*)d_b, nb_eff, ends[0], ends[1], binomial, q, nq));
  // the reference squeezes zero quotient coefficients out (polynomial.rs:371-376); they are rare, so count first
  size_t q_eff, q_nonzero;
  BP_TRY(fr_nonzero_stats_run(ctx, q, nq, 0, nq, &q_eff, &q_nonzero));
  if (q_nonzero == nq) {
    BP_HIP(ctx, hipMemcpyAsync(d_out, q, nq * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
    BP_HIP(ctx, stream_wait(ctx->stream));
    *n_out = nq;
    return BP_OK;
  }
  size_t m = nq;
  BP_TRY(fr_compact_nonzero_run(ctx, q, &m));
  if (m) BP_HIP(ctx, hipMemcpyAsync(d_out, q, m * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
  BP_HIP(ctx, stream_wait(ctx->stream));
  *n_out = m;
  return BP_OK;
}
int bp_poly_evaluate_device(bp_ctx* ctx, const void* d_coeffs, size_t n, int basis, const void* x32_mont, void* out32_mont) {
  if (!ctx || !x32_mont || !out32_mont || !basis_ok(basis) || (n && !d_coeffs)) return BP_ERR_INVALID_ARG;
  if (basis != BP_BASIS_MONOMIAL) return fail(ctx, BP_ERR_BASIS, "coeffs_evaluate needs the Monomial basis", hipSuccess, __FILE__, __LINE__);
  fr_t x, r;
  memcpy(&x, x32_mont, 32);
  DeviceGuard guard(ctx->device);
  BP_TRY(poly_eval_run(ctx, (const fr_t*)d_coeffs, n, x, &r));
  memcpy(out32_mont, &r, 32);
  return BP_OK;
}
int bp_poly_scale_powers_device(bp_ctx* ctx, const void* d_a, size_t n, const void* w32_mont, void* d_out) {
  if (!ctx || !w32_mont || (n && (!d_a || !d_out))) return BP_ERR_INVALID_ARG;
  fr_t w;
  memcpy(&w, w32_mont, 32);
  DeviceGuard guard(ctx->device);
  BP_TRY(fr_scale_powers_run(ctx, (const fr_t*)d_a, n, w, (fr_t*)d_out));
  BP_HIP(ctx, stream_wait(ctx->stream));
  return BP_OK;
}
int bp_roots_of_unity_device(bp_ctx* ctx, uint64_t group_order, void* d_out) {
  if (!ctx || !d_out) return BP_ERR_INVALID_ARG;
  fr_t w;
  if (!host_root_of_unity(w, group_order)) return fail(ctx, BP_ERR_INVALID_ARG, "group_order == 0", hipSuccess, __FILE__, __LINE__);
  if (group_order > ((uint64_t)1 << 28)) return fail(ctx, BP_ERR_TOO_LARGE, "group_order > 2^28", hipSuccess, __FILE__, __LINE__);
  DeviceGuard guard(ctx->device);
  BP_TRY(roots_run(ctx, w, group_order, (fr_t*)d_out));
  BP_HIP(ctx, stream_wait(ctx->stream));
  return BP_OK;
}
int bp_grand_product_device(bp_ctx* ctx, const void* a, const void* b, const void* c, const void* s1, const void* s2, const void* s3, size_t n,
                            const void* beta32, const void* gamma32, const void* k1_32, const void* k2_32, void* d_z) {
  if (!ctx || !beta32 || !gamma32 || !k1_32 || !k2_32 || (n && (!a || !b || !c || !s1 || !s2 || !s3 || !d_z))) return BP_ERR_INVALID_ARG;
  if (n == 0) return BP_OK;
  if (n > ((size_t)1 << 25)) return fail(ctx, BP_ERR_TOO_LARGE, "grand product longer than 2^25", hipSuccess, __FILE__, __LINE__);
  fr_t beta, gamma, k1, k2, root;
  memcpy(&beta, beta32, 32); memcpy(&gamma, gamma32, 32); memcpy(&k1, k1_32, 32); memcpy(&k2, k2_32, 32);
  host_root_of_unity(root, n);
  DeviceGuard guard(ctx->device);
  BP_TRY(grand_product_run(ctx, (const fr_t*)a, (const fr_t*)b, (const fr_t*)c, (const fr_t*)s1, (const fr_t*)s2, (const fr_t*)s3, n, beta, gamma,
                           k1, k2, root, (fr_t*)d_z));
  BP_HIP(ctx, stream_wait(ctx->stream));
  return BP_OK;
}
int bp_commit_device(bp_ctx* ctx, uint64_t srs_handle, const void* d_coeffs, size_t n, int basis, uint8_t out96[96]) {
  if (!ctx || !basis_ok(basis) || !out96) return BP_ERR_INVALID_ARG;
  if (basis != BP_BASIS_MONOMIAL) return fail(ctx, BP_ERR_BASIS, "commit needs the Monomial basis (setup.rs:34)", hipSuccess, __FILE__, __LINE__);
  uint8_t part[144];
  BP_TRY(bp_msm_g1_partial(ctx, srs_handle, 0, d_coeffs, n, BP_FR_MONT, 1, part));
  return bp_g1_partial_to_bytes96(part, out96);
}

// Several commitments against one SRS in one call (the three of prover.rs:249-251, of :483-485, the two of :640-641): their
// pipelines are in flight together (commit_many: per-device lanes, or queued shards on a group context), so one commitment's
// latency-bound tail runs under another's bulk kernel.
int bp_commit_many_device(bp_ctx* ctx, uint64_t srs_handle, const void* const* d_coeffs, const size_t* n, size_t count, int basis,
                          uint8_t* out96) {
  if (!ctx || !basis_ok(basis) || (count && (!d_coeffs || !n || !out96))) return BP_ERR_INVALID_ARG;
  if (basis != BP_BASIS_MONOMIAL) return fail(ctx, BP_ERR_BASIS, "commit needs the Monomial basis (setup.rs:34)", hipSuccess, __FILE__, __LINE__);
  if (count > 64) return fail(ctx, BP_ERR_TOO_LARGE, "more than 64 commitments in one call", hipSuccess, __FILE__, __LINE__);
  for (size_t i = 0; i < count; i++)
    if (n[i] && !d_coeffs[i]) return BP_ERR_INVALID_ARG;
  std::vector<g1_proj> cm(count);
  BP_TRY(commit_many(ctx, srs_handle, reinterpret_cast<const fr_t* const*>(d_coeffs), n, (int)count, cm.data()));
  for (size_t i = 0; i < count; i++) host_encode96(out96 + 96 * i, cm[i]);
  return BP_OK;
}

int bp_grand_product(bp_ctx* ctx, const void* a, const void* b, const void* c, const void* s1, const void* s2, const void* s3, size_t n,
                     const void* beta32, const void* gamma32, const void* k1_32, const void* k2_32, int scalar_fmt, void* z_out) {
  if (!ctx || !fmt_ok(scalar_fmt) || !beta32 || !gamma32 || !k1_32 || !k2_32 || (n && (!a || !b || !c || !s1 || !s2 || !s3 || !z_out)))
    return BP_ERR_INVALID_ARG;
  if (n == 0) return BP_OK;
  if (n > ((size_t)1 << 25)) return fail(ctx, BP_ERR_TOO_LARGE, "grand product longer than 2^25", hipSuccess, __FILE__, __LINE__);
  fr_t beta, gamma, k1, k2, root;
  if (!fr_bytes_to_mont(beta, (const uint8_t*)beta32, scalar_fmt) || !fr_bytes_to_mont(gamma, (const uint8_t*)gamma32, scalar_fmt) ||
      !fr_bytes_to_mont(k1, (const uint8_t*)k1_32, scalar_fmt) || !fr_bytes_to_mont(k2, (const uint8_t*)k2_32, scalar_fmt))
    return fail(ctx, BP_ERR_BAD_SCALAR, "scalar >= q", hipSuccess, __FILE__, __LINE__);
  host_root_of_unity(root, n);                                   // roots_of_unity(group_order), utils.rs:45-52
  DeviceGuard guard(ctx->device);
  fr_t* cols;
  BP_TRY(ws_get(ctx, "io.gp_cols", 7 * n * sizeof(fr_t), (void**)&cols));
  const void* src[6] = {a, b, c, s1, s2, s3};
  for (int j = 0; j < 6; j++) BP_HIP(ctx, hipMemcpyAsync(cols + (size_t)j * n, src[j], n * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
  if (scalar_fmt == BP_FR_BYTES_LE) BP_TRY(fr_convert_run(ctx, cols, 6 * n, 0));
  fr_t* z = cols + 6 * n;
  BP_TRY(grand_product_run(ctx, cols, cols + n, cols + 2 * n, cols + 3 * n, cols + 4 * n, cols + 5 * n, n, beta, gamma, k1, k2, root, z));
  return download_fr(ctx, z, z_out, n, scalar_fmt);
}

int bp_commit(bp_ctx* ctx, uint64_t srs_handle, const void* coeffs, size_t n, int basis, int scalar_fmt, uint8_t out96[96]) {
  if (!ctx || !basis_ok(basis)) return BP_ERR_INVALID_ARG;
  if (basis != BP_BASIS_MONOMIAL) return fail(ctx, BP_ERR_BASIS, "commit needs the Monomial basis (setup.rs:34)", hipSuccess, __FILE__, __LINE__);
  return bp_msm_g1(ctx, srs_handle, coeffs, n, scalar_fmt, out96);
}

