// prover_kernels.hpp -- element-wise kernels of the native prover (prover.hip): the polynomial identities of
// src/prover.rs rounds 1-5 evaluated without the reference's chain of O(n)-allocation Polynomial temporaries.
// All values are Montgomery Fr (scalar.rs:22), all kernels HBM-streaming.
#pragma once
#include "fields.hpp"
#include "fr_io.hpp"

namespace bp {

// out[i] = in[i] * tbl[i] for i < len, 0 for len <= i < n_out      (coset shift x -> g x, then zero padding)
__global__ void __launch_bounds__(256) fr_mul_table_pad(const fr_t* __restrict__ in, size_t len, const fr_t* __restrict__ tbl,
                                                         fr_t* __restrict__ out, size_t n_out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  fr_t r = Fr::zero();
  if (i < len) Fr::mul(r, load_fr(&in[i]), load_fr(&tbl[i]));
  store_fr(&out[i], r);
}

// out = (c0 + c1 x + c2 x^2)(x^n - 1) + base,  base has n coefficients, out has n + k  (k = 2: prover.rs:241-247, k = 3: :359-362)
__global__ void __launch_bounds__(256) fr_blind(const fr_t* __restrict__ base, size_t n, fr_t c0, fr_t c1, fr_t c2, uint32_t k,
                                                 fr_t* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n + k) return;
  fr_t r = i < n ? load_fr(&base[i]) : Fr::zero();
  const size_t j = i < n ? i : i - n;
  if (j < k) {
    fr_t c = j == 0 ? c0 : (j == 1 ? c1 : c2);
    if (i < n) Fr::sub(r, r, c); else Fr::add(r, r, c);
  }
  store_fr(&out[i], r);
}

// p[0] += v
__global__ void fr_poke_add(fr_t* p, fr_t v) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    fr_t r;
    Fr::add(r, load_fr(p), v);
    store_fr(p, r);
  }
}

// out[i] = sum_k c_k * p_k[i] (terms with i >= len_k drop out) + (i == 0 ? constant : 0),  i < n_out
constexpr int LINCOMB_MAX = 16;
struct LinComb {
  const fr_t* p[LINCOMB_MAX];
  size_t len[LINCOMB_MAX];
  fr_t c[LINCOMB_MAX];
  fr_t constant;
  int terms;
};
__global__ void __launch_bounds__(256) fr_lincomb(LinComb lc, fr_t* __restrict__ out, size_t n_out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  fr_t acc = i == 0 ? lc.constant : Fr::zero();
  for (int k = 0; k < lc.terms; k++) {
    if (i < lc.len[k]) {
      fr_t t;
      Fr::mul(t, load_fr(&lc.p[k][i]), lc.c[k]);
      Fr::add(acc, acc, t);
    }
  }
  store_fr(&out[i], acc);
}

// Round 3 (prover.rs:370-450) on the coset g <w_4n>: the quotient's evaluations
//   t = [ a ql + b qr + a b qm + c qo + PI + qc
//         + alpha ( (a + beta X + gamma)(b + beta k1 X + gamma)(c + beta k2 X + gamma) z
//                 - (a + beta s1 + gamma)(b + beta s2 + gamma)(c + beta s3 + gamma) z(w X) )
//         + alpha^2 (z - 1) L1 ] / (X^n - 1)
// wit = a | b | c | z | PI evaluations (5 x N), pre = ql qr qm qo qc s1 s2 s3 L1 evaluations (9 x N), N = 4n;
// z(w X) is z four places further round the coset (w = w_4n^4); X^n - 1 takes four values (zh_inv[i & 3]).
struct QuotientArgs {
  fr_t alpha, alpha2, beta, gamma, beta_k1, beta_k2, one;
  fr_t zh_inv[4];
};
// zshift: how many places further z(w X) sits -- 4 on the whole quotient coset g <w_4n> (w = w_4n^4), 1 on one of its four
// cosets s_j <w_n> (the coset split of a group context, where N = n and the four zh_inv are that coset's one value)
__global__ void __launch_bounds__(256) quotient_coset(const fr_t* __restrict__ wit, const fr_t* __restrict__ pre,
                                                       const fr_t* __restrict__ xs, size_t N, QuotientArgs q,
                                                       fr_t* __restrict__ out, uint32_t zshift) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const fr_t a = load_fr(&wit[i]), b = load_fr(&wit[N + i]), c = load_fr(&wit[2 * N + i]), z = load_fr(&wit[3 * N + i]);
  const fr_t zw = load_fr(&wit[3 * N + ((i + zshift) & (N - 1))]);
  fr_t acc, t, u, v;
  // gate constraints
  Fr::mul(acc, a, load_fr(&pre[i]));                                   // a ql
  Fr::mul(t, b, load_fr(&pre[N + i]));                                 // b qr
  Fr::add(acc, acc, t);
  Fr::mul(t, a, b);
  Fr::mul(t, t, load_fr(&pre[2 * N + i]));                             // a b qm
  Fr::add(acc, acc, t);
  Fr::mul(t, c, load_fr(&pre[3 * N + i]));                             // c qo
  Fr::add(acc, acc, t);
  Fr::add(acc, acc, load_fr(&wit[4 * N + i]));                         // PI
  Fr::add(acc, acc, load_fr(&pre[4 * N + i]));                         // qc
  // permutation argument
  const fr_t x = load_fr(&xs[i]);
  fr_t ag, bg, cg;                                                     // a + gamma etc.
  Fr::add(ag, a, q.gamma);
  Fr::add(bg, b, q.gamma);
  Fr::add(cg, c, q.gamma);
  Fr::mul(t, x, q.beta);     Fr::add(t, t, ag);
  Fr::mul(u, x, q.beta_k1);  Fr::add(u, u, bg);
  Fr::mul(v, x, q.beta_k2);  Fr::add(v, v, cg);
  fr_t lhs;
  Fr::mul(lhs, t, u);
  Fr::mul(lhs, lhs, v);
  Fr::mul(lhs, lhs, z);
  Fr::mul(t, load_fr(&pre[5 * N + i]), q.beta);  Fr::add(t, t, ag);
  Fr::mul(u, load_fr(&pre[6 * N + i]), q.beta);  Fr::add(u, u, bg);
  Fr::mul(v, load_fr(&pre[7 * N + i]), q.beta);  Fr::add(v, v, cg);
  fr_t rhs;
  Fr::mul(rhs, t, u);
  Fr::mul(rhs, rhs, v);
  Fr::mul(rhs, rhs, zw);
  Fr::sub(lhs, lhs, rhs);
  Fr::mul(lhs, lhs, q.alpha);
  Fr::add(acc, acc, lhs);
  // first row of the grand product
  Fr::sub(t, z, q.one);
  Fr::mul(t, t, load_fr(&pre[8 * N + i]));
  Fr::mul(t, t, q.alpha2);
  Fr::add(acc, acc, t);
  Fr::mul(acc, acc, q.zh_inv[i & 3]);
  store_fr(&out[i], acc);
}

// ---- round 3 by coset (group contexts) ---------------------------------------------------------------------------------
// out[i] = in[stride * i + offset]: one of the four cosets out of a table kept in the order of the whole quotient coset
__global__ void __launch_bounds__(256) fr_gather_stride(const fr_t* __restrict__ in, size_t stride, size_t offset, size_t n, fr_t* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) store_fr(&out[i], load_fr(&in[i * stride + offset]));
}
// Coefficients c[0 .. len), len < 2 n, prepared for the size-n transform that evaluates the polynomial on s <w_n>:
// out[i] = (c[i] + c[i + n] s^n) s^i  (x^(i+n) = s^n x^i on that coset), i < n; spow[i] = s^i
__global__ void __launch_bounds__(256) fr_fold_scale(const fr_t* __restrict__ c, size_t len, size_t n, fr_t s_n, const fr_t* __restrict__ spow,
                                                      fr_t* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fr_t v = i < len ? load_fr(&c[i]) : Fr::zero();
  if (i + n < len) {
    fr_t t;
    Fr::mul(t, load_fr(&c[i + n]), s_n);
    Fr::add(v, v, t);
  }
  Fr::mul(v, v, load_fr(&spow[i]));
  store_fr(&out[i], v);
}
// t = sum_m x^(m n) t_m from its residues w_j = t mod (x^n - sigma_j), sigma_j = G i4^j (G = g^n, i4 a primitive fourth root of
// unity): w_j = sum_m G^m i4^(j m) t_m, so G^m t_m = (1/4) sum_j i4^(-j m) w_j -- a radix-4 butterfly per coefficient index.
// v[j * n + i] = coefficient i of w_j; scale[m] = G^-m / 4; iinv = i4^-1; out[m * n + i] = coefficient i of t_m.
struct RecombineArgs { fr_t scale[4], iinv; };
__global__ void __launch_bounds__(256) coset_recombine(const fr_t* __restrict__ v, size_t n, RecombineArgs a, fr_t* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const fr_t w0 = load_fr(&v[i]), w1 = load_fr(&v[n + i]), w2 = load_fr(&v[2 * n + i]), w3 = load_fr(&v[3 * n + i]);
  fr_t e, o, d, f, r;
  Fr::add(e, w0, w2);
  Fr::add(o, w1, w3);
  Fr::sub(d, w0, w2);
  Fr::sub(f, w1, w3);
  Fr::mul(f, f, a.iinv);                     // i4^-1 (w1 - w3)
  Fr::add(r, e, o);  Fr::mul(r, r, a.scale[0]);  store_fr(&out[i], r);                 // m = 0: w0 + w1 + w2 + w3
  Fr::add(r, d, f);  Fr::mul(r, r, a.scale[1]);  store_fr(&out[n + i], r);             // m = 1: w0 + i' w1 - w2 - i' w3
  Fr::sub(r, e, o);  Fr::mul(r, r, a.scale[2]);  store_fr(&out[2 * n + i], r);         // m = 2: w0 - w1 + w2 - w3
  Fr::sub(r, d, f);  Fr::mul(r, r, a.scale[3]);  store_fr(&out[3 * n + i], r);         // m = 3: w0 - i' w1 - w2 + i' w3
}

}  // namespace bp
