// prover.hip -- Prover::prove (src/prover.rs:64-175; rounds 1-5 :177-647) on the GPU behind the C ABI, with the eleven
// blinding scalars as an input (the reference draws them from thread_rng, :108-110, so its proofs are not reproducible)
// and the Fiat-Shamir challenges from the host transcript (transcript.hpp, src/transcript.rs:4-86).
//
// The reference builds every round out of value-semantics `Polynomial` operators (each product = three O(n^2) DFTs,
// each division a long division).  Every committed or evaluated polynomial is uniquely determined by the witness, the
// circuit, the blinders and the challenges, so this file computes the same polynomials by the cheapest exact route:
//   round 1/2  b(x) (x^n - 1) + iNTT(values) written directly (fr_blind); grand product by scans (poly_kernels.hpp);
//   round 3    the quotient t = (gate + alpha perm + alpha^2 first_row) / (x^n - 1) point-wise on the coset g <w_4n>
//              (5 coset NTTs of the witness polynomials + 1 inverse; the circuit's 9 coset columns are cached);
//   round 5    the opening numerator as one fused linear combination, division by x - zeta as a scan (poly.hip).
// The reference's Div squeezes zero quotient coefficients out (polynomial.rs:371-376); the same compaction is applied
// wherever a quotient has one (probability ~ deg / q with random blinders).
// Parity: tests/test_native_prover.py -- proof bytes equal the reference-shaped restatement (tests/prover_rounds.py) run
// on the CPU oracle, and the committed golden proof of the toy circuit (tests/verify_proof_test.rs).
#include <string.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "ctx.hpp"
#include "prover_kernels.hpp"
#include "transcript.hpp"

namespace bp {

// G1Affine::to_compressed (g1.rs:221-244): big-endian x, bit 7 compressed, bit 6 infinity, bit 5 y lexicographically largest
void host_compress48(uint8_t out[48], const g1_proj& p) {
  memset(out, 0, 48);
  if (g1_is_identity(p)) {
    out[0] = 0xc0;
    return;
  }
  g1_affine a = g1_to_affine(p);
  fp_t x, y, ny;
  Fp::from_mont(x, a.x);
  Fp::from_mont(y, a.y);
  Fp::neg(ny, a.y);
  Fp::from_mont(ny, ny);
  for (int i = 0; i < 12; i++) {
    uint8_t* q = out + 4 * (11 - i);
    q[0] = (uint8_t)(x.l[i] >> 24); q[1] = (uint8_t)(x.l[i] >> 16); q[2] = (uint8_t)(x.l[i] >> 8); q[3] = (uint8_t)x.l[i];
  }
  bool larger = false;                                       // y > -y  (fp.rs:273-298)
  for (int i = 11; i >= 0; i--) {
    if (y.l[i] != ny.l[i]) {
      larger = y.l[i] > ny.l[i];
      break;
    }
  }
  out[0] |= 0x80 | (larger ? 0x20 : 0);
}

// k points at once: their affine forms share ONE field inversion (Montgomery's trick, as G1Projective::batch_normalize does,
// g1.rs:806-839) -- a Fermat inversion is ~570 field multiplications on the host, ~26 us; the identity (z = 0) is skipped
static void host_compress48_many(uint8_t* out, const g1_proj* p, int k) {
  fp_t prefix[16], acc = Fp::one();
  bool inf[16];
  for (int j = 0; j < k; j++) {
    inf[j] = g1_is_identity(p[j]);
    prefix[j] = acc;
    if (!inf[j]) Fp::mul(acc, acc, p[j].z);
  }
  fp_t inv;
  fp_invert(inv, acc);
  for (int j = k; j-- > 0;) {
    if (inf[j]) {
      host_compress48(out + 48 * j, p[j]);
      continue;
    }
    fp_t zinv;
    Fp::mul(zinv, inv, prefix[j]);
    Fp::mul(inv, inv, p[j].z);
    g1_affine a;                               // encoded directly from the affine pair (host_compress48 would invert z again)
    Fp::mul(a.x, p[j].x, zinv);
    Fp::mul(a.y, p[j].y, zinv);
    uint8_t* o = out + 48 * j;
    memset(o, 0, 48);
    fp_t x, y, ny;
    Fp::from_mont(x, a.x);
    Fp::from_mont(y, a.y);
    Fp::neg(ny, a.y);
    Fp::from_mont(ny, ny);
    for (int i = 0; i < 12; i++) {
      uint8_t* q = o + 4 * (11 - i);
      q[0] = (uint8_t)(x.l[i] >> 24); q[1] = (uint8_t)(x.l[i] >> 16); q[2] = (uint8_t)(x.l[i] >> 8); q[3] = (uint8_t)x.l[i];
    }
    bool larger = false;                       // y > -y  (fp.rs:273-298)
    for (int i = 11; i >= 0; i--) {
      if (y.l[i] != ny.l[i]) {
        larger = y.l[i] > ny.l[i];
        break;
      }
    }
    o[0] |= 0x80 | (larger ? 0x20 : 0);
  }
}

namespace {

// ------------------------------------------------------------------------------------------------ host field helpers
fr_t fmul(const fr_t& a, const fr_t& b) { fr_t r; Fr::mul(r, a, b); return r; }
fr_t fadd(const fr_t& a, const fr_t& b) { fr_t r; Fr::add(r, a, b); return r; }
fr_t fsub(const fr_t& a, const fr_t& b) { fr_t r; Fr::sub(r, a, b); return r; }
fr_t fneg(const fr_t& a) { fr_t r; Fr::neg(r, a); return r; }
fr_t finv(const fr_t& a) { fr_t r; fr_invert(r, a); return r; }
fr_t fpow(const fr_t& a, uint64_t e) {
  uint32_t e32[2] = {(uint32_t)e, (uint32_t)(e >> 32)};
  fr_t r;
  Fr::pow(r, a, e32, 2);
  return r;
}
fr_t from_u64(uint64_t v) {
  fr_t c = Fr::zero(), r;
  c.l[0] = (uint32_t)v;
  c.l[1] = (uint32_t)(v >> 32);
  Fr::to_mont(r, c);
  return r;
}
bool from_le32(fr_t& out, const uint8_t* b32) {             // Scalar::from_bytes (scalar.rs:264-288)
  fr_t v, t;
  memcpy(&v, b32, 32);
  if (!big_sub(t, v, Fr::modulus())) return false;
  Fr::to_mont(out, v);
  return true;
}
void to_le32(uint8_t* b32, const fr_t& v) {                  // Scalar::to_bytes (scalar.rs:292-304)
  fr_t t;
  Fr::from_mont(t, v);
  memcpy(b32, &t, 32);
}
fr_t root_of_unity(uint64_t order) {                         // utils.rs:39-43
  return fpow(fr_root_of_unity(false), ((uint64_t)1 << 32) / order);
}

// src/transcript.rs:4-86 (alpha is drawn under the label "z_1", :24; challenges are rejection-sampled until canonical
// and non-zero and then re-absorbed, :70-82)
struct PlonkTranscript {
  MerlinTranscript t{"plonk"};                               // prover.rs:112
  void point(const char* label, const g1_proj& p) {
    uint8_t c[48];
    host_compress48(c, p);
    t.append_message(label, c, 48);
  }
  void point48(const char* label, const uint8_t c[48]) { t.append_message(label, c, 48); }       // already compressed
  void scalar(const char* label, const fr_t& v) {
    uint8_t b[32];
    to_le32(b, v);
    t.append_message(label, b, 32);
  }
  fr_t challenge(const char* label) {
    for (;;) {
      uint8_t b[32];
      t.challenge_bytes(label, b, 32);
      fr_t v;
      if (from_le32(v, b) && !big_is_zero(v)) {
        t.append_message(label, b, 32);
        return v;
      }
    }
  }
};

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int commit(bp_ctx* ctx, uint64_t srs, const fr_t* d_coeffs, size_t n, g1_proj* out) {       // setup.rs:32-37
  uint8_t part[144];
  BP_TRY(bp_msm_g1_partial(ctx, srs, 0, d_coeffs, n, BP_FR_MONT, 1, part));
  memcpy(out, part, 144);
  return BP_OK;
}

// the reference's Div drops zero quotient coefficients (polynomial.rs:371-376): compact d_q[0..n) in place
int squeeze_zeros(bp_ctx* ctx, fr_t* d_q, size_t* n) {
  size_t eff, nonzero;
  BP_TRY(fr_nonzero_stats_run(ctx, d_q, *n, 0, *n, &eff, &nonzero));
  if (nonzero == *n) return BP_OK;
  return fr_compact_nonzero_run(ctx, d_q, n);      // scan + scatter on the device: no proof path blocks on a host copy
}

// (numerator of na coefficients) / (x - point) -> d_q, *nq coefficients, by the reference's Div semantics
int divide_by_linear(bp_ctx* ctx, fr_t* d_num, size_t na, const fr_t& point, fr_t* d_q, size_t* nq) {
  size_t na_eff, dummy;
  BP_TRY(fr_nonzero_stats_run(ctx, d_num, na, 0, 0, &na_eff, &dummy));               // trailing zeros trimmed (polynomial.rs:325-339)
  *nq = 0;
  if (na_eff < 2) return BP_OK;
  fr_t* d_b;
  BP_TRY(ws_get(ctx, "prove.divisor", 2 * sizeof(fr_t), (void**)&d_b));
  const fr_t hb[2] = {fneg(point), Fr::one()};
  BP_HIP(ctx, hipMemcpyAsync(d_b, hb, sizeof hb, hipMemcpyHostToDevice, ctx->stream));
  *nq = na_eff - 1;
  BP_TRY(poly_div_run(ctx, d_num, na_eff, d_b, 2, hb[0], hb[1], true, d_q, *nq));
  return squeeze_zeros(ctx, d_q, nq);
}

constexpr uint64_t COSET_GEN = 7;        // generator of Fr^* (scalar.rs GENERATOR): g^n w^j != 1 for every n-th-root coset used here
constexpr int N_PRE = 9;                 // ql qr qm qo qc s1 s2 s3 (+ L1 on the coset)

}  // namespace

// ------------------------------------------------------------------------------------------------ circuit
static int circuit_fill(bp_ctx* ctx, const fr_t* d_lag, CircuitEntry& e);

// takes ownership of d_lag only on success
int circuit_build(bp_ctx* ctx, uint32_t log_n, fr_t* d_lag, CircuitEntry* out) {
  CircuitEntry e;
  e.log_n = log_n;
  int rc = circuit_fill(ctx, d_lag, e);
  if (rc == BP_OK && stream_wait(ctx->stream) != hipSuccess) rc = fail(ctx, BP_ERR_HIP, "circuit_build", hipGetLastError(), __FILE__, __LINE__);
  if (rc != BP_OK) {
    circuit_release(e);               // e.lag is still null: the caller keeps d_lag
    return rc;
  }
  e.lag = d_lag;
  *out = e;
  return BP_OK;
}

static int circuit_fill(bp_ctx* ctx, const fr_t* d_lag, CircuitEntry& e) {
  const uint32_t log_n = e.log_n;
  const size_t n = (size_t)1 << log_n, N = 4 * n;
  BP_HIP(ctx, hipMalloc((void**)&e.coef, 8 * n * sizeof(fr_t)));
  BP_HIP(ctx, hipMalloc((void**)&e.coset, (size_t)N_PRE * N * sizeof(fr_t)));
  BP_HIP(ctx, hipMalloc((void**)&e.coset_x, N * sizeof(fr_t)));
  BP_HIP(ctx, hipMalloc((void**)&e.g_pow, (n + 8) * sizeof(fr_t)));
  BP_HIP(ctx, hipMalloc((void**)&e.roots, n * sizeof(fr_t)));
  BP_HIP(ctx, hipMalloc((void**)&e.ginv_pow, N * sizeof(fr_t)));
  const fr_t g = from_u64(COSET_GEN), w4n = root_of_unity(N);
  BP_TRY(roots_run(ctx, g, n + 8, e.g_pow));
  BP_TRY(roots_run(ctx, root_of_unity(n), n, e.roots));
  BP_TRY(roots_run(ctx, finv(g), N, e.ginv_pow));
  BP_TRY(roots_run(ctx, w4n, N, e.coset_x));                                             // w_4n^i ...
  BP_TRY(fr_scalar_run(ctx, e.coset_x, g, e.coset_x, N, 2));                            // ... times g
  // coefficient forms (prover.rs:379-386 recomputes these i_ntt's in every proof)
  BP_HIP(ctx, hipMemcpyAsync(e.coef, d_lag, 8 * n * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
  BP_TRY(ntt_run(ctx, e.coef, log_n, 1, 8, n));
  // their evaluations on the quotient coset, and L1's (l1_coeff = i_ntt(e_0) = [1/n; n], prover.rs:430)
  const unsigned blocks = (unsigned)((N + 255) / 256);
  for (int k = 0; k < 8; k++)
    hipLaunchKernelGGL(fr_mul_table_pad, dim3(blocks), dim3(256), 0, ctx->stream, e.coef + (size_t)k * n, n, e.g_pow, e.coset + (size_t)k * N, N);
  fr_t* l1;
  BP_TRY(ws_get(ctx, "prove.l1", n * sizeof(fr_t), (void**)&l1));
  BP_HIP(ctx, hipMemsetAsync(l1, 0, n * sizeof(fr_t), ctx->stream));
  BP_TRY(fr_scalar_run(ctx, l1, finv(from_u64(n)), l1, n, 0));                          // 0 + 1/n
  hipLaunchKernelGGL(fr_mul_table_pad, dim3(blocks), dim3(256), 0, ctx->stream, l1, n, e.g_pow, e.coset + (size_t)8 * N, N);
  BP_HIP(ctx, hipGetLastError());
  BP_TRY(ntt_run(ctx, e.coset, log_n + 2, 0, N_PRE, N));
  // 1 / (X^n - 1) on the coset: X^n = g^n i^j for X = g w_4n^j, i = w_4n^n the 4th root of unity
  const fr_t gn = fpow(g, n), i4 = fpow(w4n, n);
  fr_t p = Fr::one();
  for (int j = 0; j < 4; j++) {
    e.zh_inv[j] = finv(fsub(fmul(gn, p), Fr::one()));
    p = fmul(p, i4);
  }
  return BP_OK;
}
void circuit_release(CircuitEntry& e) {
  fr_t** owned[7] = {&e.lag, &e.coef, &e.coset, &e.coset_x, &e.g_pow, &e.ginv_pow, &e.roots};
  for (fr_t** p : owned) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  for (CosetShare& sh : e.split) {
    DeviceGuard guard(sh.member ? sh.member->device : 0);
    if (sh.member) (void)hipStreamSynchronize(sh.member->stream);
    if (sh.work) (void)hipStreamSynchronize(sh.work->stream);        // the work context itself is the member's side context: destroyed with the member
    fr_t** mine[4] = {&sh.pre, &sh.xs, &sh.spow, &sh.sinv};
    for (fr_t** p : mine) {
      if (*p) (void)hipFree(*p);
      *p = nullptr;
    }
    if (sh.done) (void)hipEventDestroy(sh.done);
    sh.done = nullptr;
  }
  e.split.clear();
}

// ------------------------------------------------------------------------------------------------ round 3 by coset (group contexts)
// src (on device src_dev) -> dst (on m's device), n elements, on m's stream
// (owner: the group member the stream's context belongs to, when m is that member's side context)
static hipError_t copy_to_member(bp_ctx* m, fr_t* dst, const fr_t* src, int src_dev, size_t n, const bp_ctx* owner = nullptr) {
  if (!peer_path(owner ? owner : m, src_dev)) return hipMemcpyAsync(dst, src, n * sizeof(fr_t), hipMemcpyDeviceToDevice, m->stream);
  return hipMemcpyPeerAsync(dst, m->device, src, src_dev, n * sizeof(fr_t), m->stream);
}

// The members' coset shares of a circuit (leader of a group; called once per circuit).  Coset j of the four is entries 4 i + j of
// the leader's tables over g <w_4n>, so a share is gathered there (strided) and copied to its member; the s_j^i / s_j^-i tables are
// computed on the member.  The first 2 members (groups of 2 or 3) take two cosets each, the first 4 (groups of 4 and more) one.
int circuit_split_build(bp_ctx* ctx, CircuitEntry& e) {
  // experiment (BP_PROVE_COSET_ONE=1): ONE device takes all four cosets itself -- four size-n coset transforms per
  // polynomial instead of one zero-padded size-4n transform; measured in profiles/r06_round3_coset_one_gpu_ab.txt
  const bool one_gpu = ctx->members.size() < 2 && knob_u32("BP_PROVE_COSET_ONE", 0, 0, 1) == 1;
  const size_t R = one_gpu ? 1 : ctx->members.size();
  if (R < 2 && !one_gpu) return BP_OK;
  const uint32_t k = e.log_n;
  const size_t n = (size_t)1 << k, N = 4 * n;
  const uint32_t used = one_gpu ? 1 : (R >= 4 ? 4 : 2), per = 4 / used;
  const fr_t g = from_u64(COSET_GEN), w4n = root_of_unity(N);
  fr_t* tmp;
  BP_TRY(ws_get(ctx, "prove.split_tmp", 10 * n * sizeof(fr_t), (void**)&tmp));
  const unsigned blocks = (unsigned)((n + 255) / 256);
  for (uint32_t r = 0; r < used; r++) {
    CosetShare sh;
    bp_ctx* m = one_gpu ? ctx : ctx->members[r];
    sh.member = m;
    sh.first = r * per;
    sh.count = per;
    {
      int rc = side_ctx_get(m, &sh.work);      // a stream, workspace and transform tables of its own on the member's device: the share's work
      if (rc != BP_OK) {                       // runs beside the member's MSM shards instead of queueing behind them
        ctx->last_error = m->last_error;
        return rc;
      }
    }
    {
      DeviceGuard guard(m->device);
      hipError_t he = hipMalloc((void**)&sh.pre, (size_t)per * 9 * n * sizeof(fr_t));
      if (he == hipSuccess) he = hipMalloc((void**)&sh.xs, (size_t)per * n * sizeof(fr_t));
      if (he == hipSuccess) he = hipMalloc((void**)&sh.spow, (size_t)per * n * sizeof(fr_t));
      if (he == hipSuccess) he = hipMalloc((void**)&sh.sinv, (size_t)per * n * sizeof(fr_t));
      if (he == hipSuccess) he = hipEventCreateWithFlags(&sh.done, hipEventDisableTiming);
      e.split.push_back(sh);                 // owned by the entry from here on (circuit_release frees what was allocated)
      if (he != hipSuccess) return fail(ctx, BP_ERR_HIP, "coset share", he, __FILE__, __LINE__);
    }
    for (uint32_t c = 0; c < per; c++) {
      const uint32_t j = sh.first + c;
      {                                       // gather coset j on the leader, wait, copy to the member
        DeviceGuard guard(ctx->device);
        for (int col = 0; col < N_PRE; col++)
          hipLaunchKernelGGL(fr_gather_stride, dim3(blocks), dim3(256), 0, ctx->stream, e.coset + (size_t)col * N, (size_t)4, (size_t)j, n, tmp + (size_t)col * n);
        hipLaunchKernelGGL(fr_gather_stride, dim3(blocks), dim3(256), 0, ctx->stream, e.coset_x, (size_t)4, (size_t)j, n, tmp + (size_t)N_PRE * n);
        BP_HIP(ctx, hipGetLastError());
        BP_HIP(ctx, stream_wait(ctx->stream));
      }
      DeviceGuard guard(m->device);
      BP_HIP(ctx, copy_to_member(m, sh.pre + (size_t)c * 9 * n, tmp, ctx->device, (size_t)N_PRE * n));
      BP_HIP(ctx, copy_to_member(m, sh.xs + (size_t)c * n, tmp + (size_t)N_PRE * n, ctx->device, n));
      const fr_t sj = fmul(g, fpow(w4n, j));
      int rc = roots_run(m, sj, n, sh.spow + (size_t)c * n);
      if (rc == BP_OK) rc = roots_run(m, finv(sj), n, sh.sinv + (size_t)c * n);
      if (rc != BP_OK) {
        ctx->last_error = m->last_error;
        return rc;
      }
      BP_HIP(ctx, stream_wait(m->stream));    // tmp is reused for the next coset
    }
  }
  return BP_OK;
}

// Round 3 by coset, in two parts, both on the shares' work contexts (CosetShare::work).
//  early: a, b, c and PI depend on no challenge (prover.rs:386-450 evaluates them after alpha only because the reference is
//         sequential): the leader enqueues their copies, folds and size-n transforms right after round 1's blinding, so they run on
//         the members beside the commitments of rounds 1 and 2 -- the members' GPUs only hold an MSM shard each then.
//  late:  once z's coefficients exist, its copy / fold / transform, the quotient on each coset, the inverse transform, the
//         unscaling by s_j^-i and the copy of the n coefficients back to the leader, which recombines the four residues.
// slots: a, b, c, z, PI (the order quotient_coset reads); `which` selects the slots a call handles.
static int coset_inputs(bp_ctx* ctx, const CircuitEntry& cir, const CosetShare& sh, const fr_t* const coefs5[5], const size_t lens5[5], unsigned which, fr_t** ev_out,
                        bool zero_pi = false) {
  const uint32_t k = cir.log_n;
  const size_t n = (size_t)1 << k, N = 4 * n, cap = n + 8;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  const fr_t g = from_u64(COSET_GEN), w4n = root_of_unity(N), gn = fpow(g, n), i4 = fpow(w4n, n);
  bp_ctx* w = sh.work;
  fr_t *cf, *ev;
  int rc = ws_get(w, "prove.split_coef", 5 * cap * sizeof(fr_t), (void**)&cf);
  if (rc == BP_OK) rc = ws_get(w, "prove.split_ev", (size_t)sh.count * 5 * n * sizeof(fr_t), (void**)&ev);
  if (rc != BP_OK) {
    ctx->last_error = w->last_error;
    return rc;
  }
  *ev_out = ev;
  BP_HIP(ctx, hipStreamWaitEvent(w->stream, ctx->ev[4], 0));          // the coefficient vectors are ready behind the leader's event
  for (int p = 0; p < 5; p++)
    if (which >> p & 1) BP_HIP(ctx, copy_to_member(w, cf + (size_t)p * cap, coefs5[p], ctx->device, lens5[p], sh.member));
  for (uint32_t c = 0; c < sh.count; c++) {
    const uint32_t j = sh.first + c;
    fr_t sn = gn;                                                     // s_j^n = g^n i4^j
    for (uint32_t q = 0; q < j; q++) sn = fmul(sn, i4);
    fr_t* e = ev + (size_t)c * 5 * n;
    if (zero_pi) BP_HIP(ctx, hipMemsetAsync(e + 4 * n, 0, n * sizeof(fr_t), w->stream));      // PI(X) = 0 on every coset
    for (int p = 0; p < 5; p++)
      if (which >> p & 1)
        hipLaunchKernelGGL(fr_fold_scale, dim3(blocks), dim3(256), 0, w->stream, cf + (size_t)p * cap, lens5[p], n, sn, sh.spow + (size_t)c * n, e + (size_t)p * n);
    BP_HIP(ctx, hipGetLastError());
    // contiguous runs of selected slots are transformed together
    for (int p = 0; p < 5 && rc == BP_OK;) {
      if (!(which >> p & 1)) {
        p++;
        continue;
      }
      int q = p;
      while (q < 5 && (which >> q & 1)) q++;
      rc = ntt_run(w, e + (size_t)p * n, k, 0, (size_t)(q - p), n);
      p = q;
    }
    if (rc != BP_OK) {
      ctx->last_error = w->last_error;
      return rc;
    }
  }
  return BP_OK;
}

static int round3_coset_early(bp_ctx* ctx, const CircuitEntry& cir, const fr_t* const coefs5[5], const size_t lens5[5], bool pi_zero) {
  BP_HIP(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
  for (const CosetShare& sh : cir.split) {
    DeviceGuard guard(sh.work->device);
    fr_t* ev;
    BP_TRY(coset_inputs(ctx, cir, sh, coefs5, lens5, pi_zero ? 0x07u : 0x17u, &ev, pi_zero));    // a, b, c, PI
  }
  return BP_OK;
}

// t (4n coefficients on the leader) = quotient of round 3.  early_done: a, b, c, PI are already on the cosets (round3_coset_early).
static int round3_by_coset(bp_ctx* ctx, const CircuitEntry& cir, const fr_t* const coefs5[5], const size_t lens5[5], const QuotientArgs& qa0, bool early_done,
                           bool pi_zero, fr_t* t) {
  const uint32_t k = cir.log_n;
  const size_t n = (size_t)1 << k, N = 4 * n;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  const fr_t g = from_u64(COSET_GEN), w4n = root_of_unity(N), gn = fpow(g, n), i4 = fpow(w4n, n);
  fr_t* v;                                                            // the four residues, one after the other
  BP_TRY(ws_get(ctx, "prove.split_v", 4 * n * sizeof(fr_t), (void**)&v));
  BP_HIP(ctx, hipEventRecord(ctx->ev[4], ctx->stream));              // z's coefficients (and the others, if not sent early) are ready behind this event
  for (const CosetShare& sh : cir.split) {
    bp_ctx* w = sh.work;
    DeviceGuard guard(w->device);
    fr_t *ev, *tq;
    BP_TRY(coset_inputs(ctx, cir, sh, coefs5, lens5, early_done ? 0x08u : (pi_zero ? 0x0Fu : 0x1Fu), &ev, pi_zero && !early_done));
    int rc = ws_get(w, "prove.split_tq", n * sizeof(fr_t), (void**)&tq);
    if (rc != BP_OK) {
      ctx->last_error = w->last_error;
      return rc;
    }
    for (uint32_t c = 0; c < sh.count; c++) {
      const uint32_t j = sh.first + c;
      QuotientArgs qa = qa0;
      for (int q = 0; q < 4; q++) qa.zh_inv[q] = cir.zh_inv[j];       // X^n - 1 = s_j^n - 1 on the whole coset
      hipLaunchKernelGGL(quotient_coset, dim3(blocks), dim3(256), 0, w->stream, ev + (size_t)c * 5 * n, sh.pre + (size_t)c * 9 * n, sh.xs + (size_t)c * n, n, qa, tq, 1u);
      BP_HIP(ctx, hipGetLastError());
      rc = ntt_run(w, tq, k, 1, 1, n);
      if (rc == BP_OK) rc = fr_binary_run(w, tq, n, sh.sinv + (size_t)c * n, n, tq, n, 2);      // coefficients of t mod (x^n - s_j^n)
      if (rc != BP_OK) {
        ctx->last_error = w->last_error;
        return rc;
      }
      // back to the leader: the copy is issued on the work stream (it follows the kernels that produced tq)
      hipError_t he = !peer_path(sh.member ? sh.member : w, ctx->device)
                          ? hipMemcpyAsync(v + (size_t)j * n, tq, n * sizeof(fr_t), hipMemcpyDeviceToDevice, w->stream)
                          : hipMemcpyPeerAsync(v + (size_t)j * n, ctx->device, tq, w->device, n * sizeof(fr_t), w->stream);
      if (he != hipSuccess) return fail(ctx, BP_ERR_HIP, "coset quotient back to the leader", he, __FILE__, __LINE__);
    }
    BP_HIP(ctx, hipEventRecord(sh.done, w->stream));
  }
  DeviceGuard guard(ctx->device);
  for (const CosetShare& sh : cir.split) BP_HIP(ctx, hipStreamWaitEvent(ctx->stream, sh.done, 0));
  RecombineArgs ra;
  const fr_t quarter = finv(from_u64(4)), gn_inv = finv(gn);
  ra.scale[0] = quarter;
  for (int mm = 1; mm < 4; mm++) ra.scale[mm] = fmul(ra.scale[mm - 1], gn_inv);
  ra.iinv = finv(i4);
  hipLaunchKernelGGL(coset_recombine, dim3(blocks), dim3(256), 0, ctx->stream, v, n, ra, t);
  BP_HIP(ctx, hipGetLastError());
  return BP_OK;
}

// ------------------------------------------------------------------------------------------------ prove
// d_wit: a | b | c | PI Lagrange columns (4 x n, Montgomery, device).  blinders: b1..b11 (prover.rs:110).
int prove_run(bp_ctx* ctx, uint64_t srs, const CircuitEntry& cir, const fr_t* d_wit, const fr_t blind[11], uint8_t proof[624], bool pi_zero,
              const ProveStaged* staged) {
  const uint32_t k = cir.log_n;
  const size_t n = (size_t)1 << k, N = 4 * n;
  const fr_t omega = root_of_unity(n), k1 = from_u64(2), k2 = from_u64(3), one = Fr::one();      // prover.rs:99-100
  hipStream_t st = ctx->stream;
  if (ctx->side) BP_HIP(ctx, stream_wait(ctx->side->stream));     // side work of a proof that was abandoned half way must not outlive its buffers
  for (const CosetShare& sh : cir.split) {
    DeviceGuard guard(sh.work->device);
    BP_HIP(ctx, stream_wait(sh.work->stream));
  }
  const double t_start = now_ms();
  PlonkTranscript tr;
  g1_proj cm[9];

  // ---- round 1 (prover.rs:177-277): a, b, c (and PI) to coefficient form, blinded by (b2 + b1 x)(x^n - 1) etc.
  fr_t *coefs, *abc, *zc;
  BP_TRY(ws_get(ctx, "prove.coefs", 4 * n * sizeof(fr_t), (void**)&coefs));          // iNTT of a | b | c | PI
  BP_TRY(ws_get(ctx, "prove.abc", 3 * (n + 8) * sizeof(fr_t), (void**)&abc));        // a_coeff | b_coeff | c_coeff, n + 2 each
  BP_TRY(ws_get(ctx, "prove.z", 2 * (n + 8) * sizeof(fr_t), (void**)&zc));           // z (Lagrange -> coefficients) | z_coeff, n + 3
  fr_t* poly_abc[3] = {abc, abc + (n + 8), abc + 2 * (n + 8)};
  const unsigned blocks_n = (unsigned)((n + 8 + 255) / 256);
  bool r1_committed = false;
  if (staged) {
    // Host witness (ctx.hpp, ProveStaged): per column  upload -> (bytes -> Montgomery) -> inverse transform -> blinding  on the side
    // context's stream, an event, and the column's commitment on its lane behind the event.  The host blocks in the pageable copy of
    // column j + 1 while the GPU commits to column j: 2.2 ms of uploads at 2^20 gates leave 0.8 on the proof's path.
    bp_ctx* sd;
    BP_TRY(side_ctx_get(ctx, &sd));
    hipStream_t ss = sd->stream;
    for (auto& e : ctx->seam_ev)
      if (!e) BP_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    fr_t* wit = const_cast<fr_t*>(d_wit);                           // the caller's "prove.witness" workspace: filled here
    BP_HIP(ctx, hipEventRecord(ctx->seam_ev[3], st));               // behind whatever the context's stream still holds
    BP_HIP(ctx, hipStreamWaitEvent(ss, ctx->seam_ev[3], 0));
    MsmPending pend[3];
    int launched = 0, rc = BP_OK;
    for (int j = 0; j < 4 && rc == BP_OK; j++) {
      fr_t *w_j = wit + (size_t)j * n, *c_j = coefs + (size_t)j * n;
      hipError_t he;
      if (staged->cols[j]) {
        he = hipMemcpyAsync(w_j, staged->cols[j], n * sizeof(fr_t), hipMemcpyHostToDevice, ss);
        if (he == hipSuccess && staged->fmt == BP_FR_BYTES_LE && fr_convert_run(sd, w_j, n, 0) != BP_OK) he = hipErrorUnknown;
        if (he == hipSuccess) he = hipMemcpyAsync(c_j, w_j, n * sizeof(fr_t), hipMemcpyDeviceToDevice, ss);
        if (he == hipSuccess && ntt_run(sd, c_j, k, 1, 1, n) != BP_OK) he = hipErrorUnknown;
      } else {                                                      // only PI may be absent: a zero column and zero coefficients
        he = hipMemsetAsync(w_j, 0, n * sizeof(fr_t), ss);
        if (he == hipSuccess) he = hipMemsetAsync(c_j, 0, n * sizeof(fr_t), ss);
      }
      if (he == hipSuccess && j < 3) {
        hipLaunchKernelGGL(fr_blind, dim3(blocks_n), dim3(256), 0, ss, c_j, n, blind[2 * j + 1], blind[2 * j], Fr::zero(), 2u, poly_abc[j]);
        he = hipGetLastError();
      }
      if (he == hipSuccess) he = hipEventRecord(ctx->seam_ev[j], ss);
      if (he != hipSuccess) {
        rc = fail(ctx, BP_ERR_HIP, "round 1: staging of a witness column", he, __FILE__, __LINE__);
        break;
      }
      if (j < 3) {
        rc = commit_lane_launch(ctx, j, srs, poly_abc[j], n + 2, ctx->seam_ev[j], &pend[j]);
        if (rc == BP_OK) launched = j + 1;
      }
    }
    if (rc == BP_OK) {                                              // rounds 2.. read the columns and coefficients on the context's stream
      const hipError_t he = hipStreamWaitEvent(st, ctx->seam_ev[3], 0);
      if (he != hipSuccess) rc = fail(ctx, BP_ERR_HIP, "round 1: order", he, __FILE__, __LINE__);
    }
    for (int j = 0; j < launched; j++) {                            // every launched lane is waited for, also after a failure
      const int r2 = commit_lane_finish(ctx, j, pend[j], &cm[j]);
      if (rc == BP_OK) rc = r2;
    }
    if (rc != BP_OK) {
      (void)stream_wait(ss);
      return rc;
    }
    r1_committed = true;
  } else {
    BP_HIP(ctx, hipMemcpyAsync(coefs, d_wit, 4 * n * sizeof(fr_t), hipMemcpyDeviceToDevice, st));
    BP_TRY(ntt_run(ctx, coefs, k, 1, pi_zero ? 3 : 4, n));          // no public inputs: PI's column is zero and so are its coefficients
    for (int j = 0; j < 3; j++)
      hipLaunchKernelGGL(fr_blind, dim3(blocks_n), dim3(256), 0, st, coefs + (size_t)j * n, n, blind[2 * j + 1], blind[2 * j], Fr::zero(), 2u,
                         poly_abc[j]);
    BP_HIP(ctx, hipGetLastError());
  }
  // Round 3's evaluations of a, b, c and PI on the quotient coset (prover.rs:386-450) depend on no challenge: they are enqueued on the
  // side context's stream once round 1's commitments are in (beside them they only took the GPU from three concurrent pipelines:
  // +2.6 ms on round 1 for -1.9 on round 3) and run beside round 2 -- one commitment, whose sort and tree leave most of the chip idle,
  // and the host's transcript steps.  Round 3 waits for the second event and transforms z alone.  That pays while the commitment's
  // accumulation leaves the chip partly idle: below 2^20 gates (-0.05 .. -0.15 ms per proof at 2^16 / 2^18).  From 2^20 msm_accumulate
  // fills every SIMD for 2 ms and whatever runs beside it only delays its workgroups (+0.4 ms per proof), so all five transforms stay
  // in round 3 there.  BP_PROVE_SIDE=0 / 1 forces either.
  fr_t* ev;
  BP_TRY(ws_get(ctx, "prove.coset_wit", 5 * N * sizeof(fr_t), (void**)&ev));         // a | b | c | z | PI evaluations
  const unsigned blocks_N = (unsigned)((N + 255) / 256);
  bool split_on = !cir.split.empty();                                                // group context: round 3 by coset over the members
  {
    const char* v = knob("BP_PROVE_COSET_SPLIT");
    if (v && *v == '0') split_on = false;
  }
  bool side_on = !split_on && k < 20;
  {
    const char* v = knob("BP_PROVE_SIDE");
    if (v && *v == '0') side_on = false;
    if (v && *v == '1') side_on = !split_on;
  }
  bool early_on = split_on;                // a, b, c, PI to the members' cosets now, beside the commitments (BP_PROVE_COSET_EARLY=0: in round 3)
  {
    const char* v = knob("BP_PROVE_COSET_EARLY");
    if (v && *v == '0') early_on = false;
  }
  const fr_t* coefs5[5] = {poly_abc[0], poly_abc[1], poly_abc[2], zc + (n + 8), coefs + 3 * n};
  const size_t lens5[5] = {n + 2, n + 2, n + 2, n + 3, n};
  if (early_on) BP_TRY(round3_coset_early(ctx, cir, coefs5, lens5, pi_zero));
  if (!r1_committed) {                    // three independent commitments in flight together (commit_many)
    const fr_t* polys[3] = {poly_abc[0], poly_abc[1], poly_abc[2]};
    const size_t lens[3] = {n + 2, n + 2, n + 2};
    BP_TRY(commit_many(ctx, srs, polys, lens, 3, &cm[0]));
  }
  bp_ctx* side = nullptr;
  if (side_on) {
    BP_TRY(side_ctx_get(ctx, &side));
    BP_HIP(ctx, hipEventRecord(ctx->side_ev[0], st));
    hipStream_t ss = side->stream;
    BP_HIP(ctx, hipStreamWaitEvent(ss, ctx->side_ev[0], 0));
    for (int j = 0; j < 3; j++) hipLaunchKernelGGL(fr_mul_table_pad, dim3(blocks_N), dim3(256), 0, ss, poly_abc[j], n + 2, cir.g_pow, ev + (size_t)j * N, N);
    if (pi_zero) BP_HIP(ctx, hipMemsetAsync(ev + 4 * N, 0, N * sizeof(fr_t), ss));
    else hipLaunchKernelGGL(fr_mul_table_pad, dim3(blocks_N), dim3(256), 0, ss, coefs + 3 * n, n, cir.g_pow, ev + 4 * N, N);
    BP_HIP(ctx, hipGetLastError());
    int rc = ntt_run(side, ev, k + 2, 0, 3, N);
    if (rc == BP_OK && !pi_zero) rc = ntt_run(side, ev + 4 * N, k + 2, 0, 1, N);
    if (rc != BP_OK) {
      ctx->last_error = side->last_error;
      return rc;
    }
    BP_HIP(ctx, hipEventRecord(ctx->side_ev[1], ss));
  }
  host_compress48_many(proof, &cm[0], 3);        // compressed once, into the proof; the transcript absorbs the same 48 bytes (transcript.rs:66-69)
  tr.point48("a_1", proof); tr.point48("b_1", proof + 48); tr.point48("c_1", proof + 96);
  const fr_t beta = tr.challenge("beta"), gamma = tr.challenge("gamma");
  const double t_r1 = now_ms();

  // ---- round 2 (prover.rs:279-368): permutation grand product, blinded by (b9 + b8 x + b7 x^2)(x^n - 1)
  fr_t *z_lag = zc, *z_coeff = zc + (n + 8);
  BP_TRY(grand_product_run(ctx, d_wit, d_wit + n, d_wit + 2 * n, cir.lag + 5 * n, cir.lag + 6 * n, cir.lag + 7 * n, n, beta, gamma, k1, k2,
                           omega, z_lag, cir.roots));
  BP_TRY(ntt_run(ctx, z_lag, k, 1, 1, n));
  hipLaunchKernelGGL(fr_blind, dim3(blocks_n), dim3(256), 0, st, z_lag, n, blind[8], blind[7], blind[6], 3u, z_coeff);
  BP_HIP(ctx, hipGetLastError());
  BP_TRY(commit(ctx, srs, z_coeff, n + 3, &cm[3]));
  host_compress48_many(proof + 144, &cm[3], 1);
  tr.point48("z_1", proof + 144);
  const fr_t alpha = tr.challenge("z_1");                                           // transcript.rs:24
  const double t_r2 = now_ms();

  // ---- round 3 (prover.rs:370-500): quotient on the coset g <w_4n>
  fr_t* t;
  BP_TRY(ws_get(ctx, "prove.t", (N + 16) * sizeof(fr_t), (void**)&t));
  QuotientArgs qa;
  qa.alpha = alpha; qa.alpha2 = fmul(alpha, alpha); qa.beta = beta; qa.gamma = gamma;
  qa.beta_k1 = fmul(beta, k1); qa.beta_k2 = fmul(beta, k2); qa.one = one;
  for (int j = 0; j < 4; j++) qa.zh_inv[j] = cir.zh_inv[j];
  if (split_on) {
    BP_TRY(round3_by_coset(ctx, cir, coefs5, lens5, qa, early_on, pi_zero, t));                // t = the 4n quotient coefficients
  } else {
  if (side_on) {
    hipLaunchKernelGGL(fr_mul_table_pad, dim3(blocks_N), dim3(256), 0, st, z_coeff, n + 3, cir.g_pow, ev + 3 * N, N);
    BP_HIP(ctx, hipGetLastError());
    BP_TRY(ntt_run(ctx, ev + 3 * N, k + 2, 0, 1, N));
    BP_HIP(ctx, hipStreamWaitEvent(st, ctx->side_ev[1], 0));                         // a, b, c, PI: transformed beside rounds 1-2
  } else {
    for (int j = 0; j < 3; j++) hipLaunchKernelGGL(fr_mul_table_pad, dim3(blocks_N), dim3(256), 0, st, poly_abc[j], n + 2, cir.g_pow, ev + (size_t)j * N, N);
    hipLaunchKernelGGL(fr_mul_table_pad, dim3(blocks_N), dim3(256), 0, st, z_coeff, n + 3, cir.g_pow, ev + 3 * N, N);
    if (pi_zero) BP_HIP(ctx, hipMemsetAsync(ev + 4 * N, 0, N * sizeof(fr_t), st));
    else hipLaunchKernelGGL(fr_mul_table_pad, dim3(blocks_N), dim3(256), 0, st, coefs + 3 * n, n, cir.g_pow, ev + 4 * N, N);
    BP_HIP(ctx, hipGetLastError());
    BP_TRY(ntt_run(ctx, ev, k + 2, 0, pi_zero ? 4 : 5, N));
  }
  hipLaunchKernelGGL(quotient_coset, dim3(blocks_N), dim3(256), 0, st, ev, cir.coset, cir.coset_x, N, qa, t, 4u);
  BP_HIP(ctx, hipGetLastError());
  BP_TRY(ntt_run(ctx, t, k + 2, 1, 1, N));
  BP_TRY(fr_binary_run(ctx, t, N, cir.ginv_pow, N, t, N, 2));
  }
  size_t t_len, dummy;
  BP_TRY(fr_nonzero_stats_run(ctx, t, N, 0, 0, &t_len, &dummy));
  // deg(numerator) <= 4n + 5, so an exact quotient has at most 3n + 6 coefficients; anything longer means the division by
  // x^n - 1 left a remainder, i.e. the witness does not satisfy the circuit (the reference would panic at prover.rs:615)
  if (t_len > 3 * n + 6) return fail(ctx, BP_ERR_ASSERT, "round 3: constraints not divisible by x^n - 1 (witness does not satisfy the circuit)", hipSuccess, __FILE__, __LINE__);
  BP_TRY(squeeze_zeros(ctx, t, &t_len));
  if (t_len <= 2 * n) return fail(ctx, BP_ERR_ASSERT, "round 3: quotient shorter than 2n + 1 (t_hi would be empty; values[0] panics)", hipSuccess, __FILE__, __LINE__);
  // split_t_to_3pieces (:649-659) and the blinding of :475-481
  fr_t *t_lo, *t_mid, *t_hi;
  BP_TRY(ws_get(ctx, "prove.t_parts", 3 * (n + 8) * sizeof(fr_t), (void**)&t_lo));
  t_mid = t_lo + (n + 8);
  t_hi = t_mid + (n + 8);
  const size_t hi_len = t_len - 2 * n;
  BP_HIP(ctx, hipMemcpyAsync(t_lo, t, n * sizeof(fr_t), hipMemcpyDeviceToDevice, st));
  BP_HIP(ctx, hipMemcpyAsync(t_mid, t + n, n * sizeof(fr_t), hipMemcpyDeviceToDevice, st));
  BP_HIP(ctx, hipMemcpyAsync(t_hi, t + 2 * n, hi_len * sizeof(fr_t), hipMemcpyDeviceToDevice, st));
  BP_HIP(ctx, hipMemcpyAsync(t_lo + n, &blind[9], sizeof(fr_t), hipMemcpyHostToDevice, st));        // + b10 x^n
  BP_HIP(ctx, hipMemcpyAsync(t_mid + n, &blind[10], sizeof(fr_t), hipMemcpyHostToDevice, st));      // + b11 x^n ...
  hipLaunchKernelGGL(fr_poke_add, dim3(1), dim3(1), 0, st, t_mid, fneg(blind[9]));                  // ... - b10
  hipLaunchKernelGGL(fr_poke_add, dim3(1), dim3(1), 0, st, t_hi, fneg(blind[10]));                  // - b11
  BP_HIP(ctx, hipGetLastError());
  {
    const fr_t* polys[3] = {t_lo, t_mid, t_hi};
    const size_t lens[3] = {n + 1, n + 1, hi_len};
    BP_TRY(commit_many(ctx, srs, polys, lens, 3, &cm[4]));
  }
  host_compress48_many(proof + 192, &cm[4], 3);
  tr.point48("t_lo_1", proof + 192); tr.point48("t_mid_1", proof + 240); tr.point48("t_hi_1", proof + 288);
  const fr_t zeta = tr.challenge("zeta");
  const double t_r3 = now_ms();

  // ---- round 4 (prover.rs:502-541): evaluations at zeta (z at zeta w)
  const fr_t *s1c = cir.coef + 5 * n, *s2c = cir.coef + 6 * n, *s3c = cir.coef + 7 * n;
  fr_t a_bar, b_bar, c_bar, s1_bar, s2_bar, zw_bar, pi_zeta;
  {                                        // the six evaluations (and PI(zeta), which round 5 wants) in one enqueue and one wait
    const fr_t* polys[7] = {poly_abc[0], poly_abc[1], poly_abc[2], s1c, s2c, z_coeff, coefs + 3 * n};
    const size_t lens[7] = {n + 2, n + 2, n + 2, n, n, n + 3, n};
    const fr_t zw = fmul(zeta, omega);                                               // z_omega(zeta) = z(zeta w), :661-674
    const fr_t at[7] = {zeta, zeta, zeta, zeta, zeta, zw, zeta};
    fr_t vals[7];
    BP_TRY(poly_eval_many_run(ctx, 7, polys, lens, at, vals));
    a_bar = vals[0]; b_bar = vals[1]; c_bar = vals[2]; s1_bar = vals[3]; s2_bar = vals[4]; zw_bar = vals[5]; pi_zeta = vals[6];
  }
  tr.scalar("a_eval", a_bar); tr.scalar("b_eval", b_bar); tr.scalar("c_eval", c_bar);
  tr.scalar("s1_eval", s1_bar); tr.scalar("s2_eval", s2_bar); tr.scalar("z_shifted_eval", zw_bar);
  const fr_t nu = tr.challenge("nu");
  const double t_r4 = now_ms();

  // ---- round 5 (prover.rs:543-647): linearisation r and the two opening quotients
  const fr_t zeta_n = fpow(zeta, n), zh_zeta = fsub(zeta_n, one);
  // L1(zeta) = (1/n) sum_i zeta^i  (l1_coeff.coeffs_evaluate, :590)
  const fr_t l1_zeta = big_eq(zeta, one) ? one : fmul(zh_zeta, finv(fmul(from_u64(n), fsub(zeta, one))));
  const fr_t bz = fmul(beta, zeta);
  const fr_t f_a = fadd(fadd(a_bar, bz), gamma), f_b = fadd(fadd(b_bar, fmul(bz, k1)), gamma), f_c = fadd(fadd(c_bar, fmul(bz, k2)), gamma);
  const fr_t g_a = fadd(fadd(a_bar, fmul(s1_bar, beta)), gamma), g_b = fadd(fadd(b_bar, fmul(s2_bar, beta)), gamma);
  const fr_t alpha2 = fmul(alpha, alpha);
  const fr_t perm_z = fmul(alpha, fmul(fmul(f_a, f_b), f_c));                        // coefficient of z in alpha r2
  const fr_t perm_s = fmul(alpha, fmul(fmul(g_a, g_b), zw_bar));                     // alpha (..)(..) z_omega_bar
  fr_t nup[6];
  nup[0] = one;
  for (int j = 1; j < 6; j++) nup[j] = fmul(nup[j - 1], nu);
  LinComb lc;
  memset(&lc, 0, sizeof lc);
  int m = 0;
  auto term = [&](const fr_t* p, size_t len, const fr_t& c) { lc.p[m] = p; lc.len[m] = len; lc.c[m] = c; m++; };
  term(cir.coef + 2 * n, n, fmul(a_bar, b_bar));                                     // r1: qm a b + ql a + qr b + qo c + PI(zeta) + qc
  term(cir.coef + 0 * n, n, a_bar);
  term(cir.coef + 1 * n, n, b_bar);
  term(cir.coef + 3 * n, n, c_bar);
  term(cir.coef + 4 * n, n, one);
  term(z_coeff, n + 3, fadd(perm_z, fmul(alpha2, l1_zeta)));                         // alpha r2 (z part) + alpha^2 r3 (z part)
  term(s3c, n, fneg(fmul(perm_s, beta)));                                            // alpha r2: -(s3 beta + c + gamma) (..)(..) z_w
  term(t_lo, n + 1, fneg(zh_zeta));                                                  // -r4
  term(t_mid, n + 1, fneg(fmul(zeta_n, zh_zeta)));
  term(t_hi, hi_len, fneg(fmul(fmul(zeta_n, zeta_n), zh_zeta)));
  fr_t r_const = fsub(fsub(pi_zeta, fmul(perm_s, fadd(c_bar, gamma))), fmul(alpha2, l1_zeta));
  const int r_terms = m;
  term(poly_abc[0], n + 2, nup[1]);                                                  // + nu (a - a_bar) + ... (:623-634)
  term(poly_abc[1], n + 2, nup[2]);
  term(poly_abc[2], n + 2, nup[3]);
  term(s1c, n, nup[4]);
  term(s2c, n, nup[5]);
  fr_t open_const = fadd(fadd(fmul(nup[1], a_bar), fmul(nup[2], b_bar)), fadd(fmul(nup[3], c_bar), fadd(fmul(nup[4], s1_bar), fmul(nup[5], s2_bar))));
  lc.terms = m;
  lc.constant = fsub(r_const, open_const);
  size_t num_len = std::max({n + 3, n + 2, hi_len, n + 1});
  fr_t *num, *w_zeta, *w_zeta_omega;
  BP_TRY(ws_get(ctx, "prove.num", (n + 16) * sizeof(fr_t), (void**)&num));
  BP_TRY(ws_get(ctx, "prove.w", 2 * (n + 16) * sizeof(fr_t), (void**)&w_zeta));
  w_zeta_omega = w_zeta + (n + 16);
  if (num_len > n + 16) return fail(ctx, BP_ERR_ASSERT, "round 5: quotient piece longer than n + 16", hipSuccess, __FILE__, __LINE__);
  hipLaunchKernelGGL(fr_lincomb, dim3((unsigned)((num_len + 255) / 256)), dim3(256), 0, st, lc, num, num_len);
  BP_HIP(ctx, hipGetLastError());
  // r(zeta) == 0 (prover.rs:615): the opening terms vanish at zeta, so the numerator's value there is r's
  fr_t check;
  BP_TRY(poly_eval_run(ctx, num, num_len, zeta, &check));
  (void)r_terms;
  if (!big_is_zero(check)) return fail(ctx, BP_ERR_ASSERT, "round 5: r(zeta) != 0 (prover.rs:615)", hipSuccess, __FILE__, __LINE__);
  size_t wz_len, wzo_len;
  BP_TRY(divide_by_linear(ctx, num, num_len, zeta, w_zeta, &wz_len));
  // W_zeta_omega = (z - z_omega_bar) / (x - zeta w)  (:636-638)
  BP_HIP(ctx, hipMemcpyAsync(num, z_coeff, (n + 3) * sizeof(fr_t), hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(fr_poke_add, dim3(1), dim3(1), 0, st, num, fneg(zw_bar));
  BP_HIP(ctx, hipGetLastError());
  BP_TRY(divide_by_linear(ctx, num, n + 3, fmul(zeta, omega), w_zeta_omega, &wzo_len));
  {
    const fr_t* polys[2] = {w_zeta, w_zeta_omega};
    const size_t lens[2] = {wz_len, wzo_len};
    BP_TRY(commit_many(ctx, srs, polys, lens, 2, &cm[7]));
  }
  const double t_r5 = now_ms();

  // ---- Proof (verifier.rs:23-40 field order): 9 compressed points, then the 6 evaluations as 32-byte little-endian
  host_compress48_many(proof + 336, &cm[7], 2);                   // the first seven were compressed when the transcript absorbed them
  const fr_t evals[6] = {a_bar, b_bar, c_bar, s1_bar, s2_bar, zw_bar};
  for (int j = 0; j < 6; j++) to_le32(proof + 432 + 32 * j, evals[j]);
  ctx->prove_ms[0] = (float)(t_r1 - t_start); ctx->prove_ms[1] = (float)(t_r2 - t_r1); ctx->prove_ms[2] = (float)(t_r3 - t_r2);
  ctx->prove_ms[3] = (float)(t_r4 - t_r3); ctx->prove_ms[4] = (float)(t_r5 - t_r4); ctx->prove_ms[5] = (float)(now_ms() - t_start);
  return BP_OK;
}

void transcript_test_vector(uint8_t out32[32]) {
  MerlinTranscript t("test protocol");
  t.append_message("some label", (const uint8_t*)"some data", 9);
  t.challenge_bytes("challenge", out32, 32);
}

}  // namespace bp
