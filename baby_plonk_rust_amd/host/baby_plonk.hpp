// baby_plonk.hpp -- C++ host-side mirror of the reference's Rust interface for the hot path, on top of the
// C ABI (include/bp_msm_ntt.h).  The reference is compiled Rust and no Rust toolchain exists in this image,
// so this header plays the role the Rust shim of INTEGRATION.md plays in the reference tree: same names,
// same argument meaning, same failure behaviour (the reference panics; these throw bp::Panic).
//
//   reference                                               here
//   Scalar (lib/bls12_381/src/scalar.rs:22)                 baby_plonk::Scalar      (4 x u64 Montgomery limbs)
//   G1Projective / G1Affine wire form (g1.rs:246-260)       baby_plonk::G1          (96-byte uncompressed affine)
//   BucketMSM::bucket_msm (src/msm.rs:76-118)               BucketMSM::bucket_msm
//   ntt_381 / i_ntt_381 (src/utils.rs:63,106)               ntt_381 / i_ntt_381
//   root_of_unity / roots_of_unity (src/utils.rs:39-52)     root_of_unity / roots_of_unity
//   Polynomial + operators (src/polynomial.rs:14-380)       Polynomial
//   Setup::generate_srs / commit (src/setup.rs:12-37)       Setup
//   CommonPreprocessedInput (src/program.rs:34-50)          CommonPreprocessedInput
//   Prover::new / prove (src/prover.rs:64-175)              Prover::prove_with_blinding (blinders are an argument)
//   Proof (src/verifier.rs:23-40)                           Proof (624 bytes: 9 compressed points, 6 scalars)
#pragma once
#include <array>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/bp_msm_ntt.h"

namespace baby_plonk {

struct Panic : std::runtime_error {
  int code;
  Panic(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

class Context {
 public:
  explicit Context(int device = 0) {
    int rc = bp_init(&ctx_, device);
    if (rc != BP_OK) throw Panic(rc, "bp_init failed: no usable GPU (there is no CPU fallback)");
  }
  // one context over several GPUs (bp_init_multi): SRS and commitments sharded by point range inside the library
  explicit Context(const std::vector<int>& devices) {
    int rc = bp_init_multi(&ctx_, devices.data(), (int)devices.size());
    if (rc != BP_OK) throw Panic(rc, "bp_init_multi failed: no usable GPU (there is no CPU fallback)");
  }
  ~Context() { bp_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  bp_ctx* raw() const { return ctx_; }
  void check(int rc, const char* where) const {
    if (rc != BP_OK) throw Panic(rc, std::string(where) + ": " + bp_last_error(ctx_));
  }
  // the process-wide default: GPU 0, or the comma-separated device list of the environment variable BP_DEVICES
  // (e.g. BP_DEVICES=0,1,2,3,4,5,6,7 makes every Setup::commit of an unmodified caller span eight GPUs)
  static Context& global() {
    static Context c(env_devices());
    return c;
  }
  int shards() const { return bp_ctx_devices(ctx_, nullptr, 0); }
  // One process per GPU: rank 0 makes the 128-byte id, the host carries it to the other ranks, every rank joins.
  // The communicator (RCCL) lives inside the library's context; Setup::commit_over_ranks then is one call.
  static std::array<uint8_t, BP_COMM_ID_BYTES> unique_id() {
    std::array<uint8_t, BP_COMM_ID_BYTES> id{};
    int rc = bp_comm_unique_id(id.data());
    if (rc != BP_OK) throw Panic(rc, "bp_comm_unique_id failed");
    return id;
  }
  void join_ranks(const std::array<uint8_t, BP_COMM_ID_BYTES>& id, int rank, int world) { check(bp_comm_init_rank(ctx_, id.data(), rank, world), "join_ranks"); }
  void leave_ranks() { check(bp_comm_destroy(ctx_), "leave_ranks"); }
  // bound of every wait behind a collective and of join_ranks itself (ms; 0 = none): a missing or stuck rank becomes a Panic, not a stall
  void set_rank_timeout_ms(uint32_t ms) { check(bp_comm_set_timeout_ms(ctx_, ms), "set_rank_timeout_ms"); }
  uint64_t collectives() const {
    uint64_t n = 0;
    bp_comm_stats(ctx_, &n, nullptr);
    return n;
  }
  int world() const {
    int w = 0;
    bp_comm_info(ctx_, nullptr, &w);
    return w;
  }

 private:
  static std::vector<int> env_devices() {
    std::vector<int> d;
    if (const char* e = std::getenv("BP_DEVICES")) {
      for (const char* p = e; *p;) {
        char* end = nullptr;
        long v = std::strtol(p, &end, 10);
        if (end == p) break;
        d.push_back((int)v);
        p = *end == ',' ? end + 1 : end;
      }
    }
    if (d.empty()) d.push_back(0);
    return d;
  }
  bp_ctx* ctx_ = nullptr;
};

// scalar.rs:22 -- Montgomery limbs (what Scalar::to_array exposes, scalar.rs:35-40)
struct Scalar {
  std::array<uint64_t, 4> l{};
  bool operator==(const Scalar& o) const { return l == o.l; }
  bool operator!=(const Scalar& o) const { return !(l == o.l); }
  static Scalar zero() { return Scalar{}; }
  // Scalar::from_bytes (scalar.rs:264-288): 32-byte little-endian canonical -> Montgomery limbs; rejects >= q
  static Scalar from_bytes(const std::array<uint8_t, 32>& b) {
    Scalar out;
    int rc = bp_fr_convert(b.data(), 1, BP_FR_BYTES_LE, BP_FR_MONT, out.l.data());
    if (rc != BP_OK) throw Panic(rc, "Scalar::from_bytes: not canonical");
    return out;
  }
  // Scalar::to_bytes (scalar.rs:292-304)
  std::array<uint8_t, 32> to_bytes() const {
    std::array<uint8_t, 32> b{};
    bp_fr_convert(l.data(), 1, BP_FR_MONT, BP_FR_BYTES_LE, b.data());
    return b;
  }
  static Scalar from_u64(uint64_t v) {                         // impl From<u64> for Scalar (scalar.rs:48-52)
    std::array<uint8_t, 32> b{};
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i));
    return from_bytes(b);
  }
};

enum class Basis { Lagrange = BP_BASIS_LAGRANGE, Monomial = BP_BASIS_MONOMIAL };   // polynomial.rs:8-11

using G1 = std::array<uint8_t, 96>;   // G1Affine::to_uncompressed (g1.rs:246-260)

// src/utils.rs:39-43
inline Scalar root_of_unity(uint64_t group_order) {
  Scalar s;
  int rc = bp_root_of_unity(group_order, BP_FR_MONT, reinterpret_cast<uint8_t*>(s.l.data()));
  if (rc != BP_OK) throw Panic(rc, "root_of_unity: division by zero");
  return s;
}
// src/utils.rs:45-52
inline std::vector<Scalar> roots_of_unity(uint64_t group_order, Context& c = Context::global()) {
  std::vector<Scalar> out(group_order);
  c.check(bp_roots_of_unity(c.raw(), group_order, BP_FR_MONT, out.data()), "roots_of_unity");
  return out;
}
inline bool is_power_of_two(uint64_t n) { return n != 0 && (n & (n - 1)) == 0; }   // utils.rs:82-84
inline uint32_t log2_exact(uint64_t n) {
  uint32_t k = 0;
  while ((1ull << k) < n) k++;
  return k;
}
// src/utils.rs:63-81 -- asserts a power-of-two length
inline std::vector<Scalar> ntt_381(const std::vector<Scalar>& elements, Context& c = Context::global()) {
  if (!is_power_of_two(elements.size())) throw Panic(BP_ERR_NOT_POW2, "assertion failed: is_power_of_two(n)");
  std::vector<Scalar> out = elements;
  c.check(bp_ntt_fr(c.raw(), out.data(), log2_exact(out.size()), 0, BP_FR_MONT, 1, out.size()), "ntt_381");
  return out;
}
// src/utils.rs:106-129
inline std::vector<Scalar> i_ntt_381(const std::vector<Scalar>& elements, Context& c = Context::global()) {
  if (!is_power_of_two(elements.size())) throw Panic(BP_ERR_NOT_POW2, "assertion failed: is_power_of_two(n)");
  std::vector<Scalar> out = elements;
  c.check(bp_ntt_fr(c.raw(), out.data(), log2_exact(out.size()), 1, BP_FR_MONT, 1, out.size()), "i_ntt_381");
  return out;
}

// src/polynomial.rs:14-17 -- value semantics: every operator returns a fresh polynomial
class Polynomial {
 public:
  std::vector<Scalar> values;
  Basis basis;
  Polynomial(std::vector<Scalar> v, Basis b) : values(std::move(v)), basis(b) {}
  bool operator==(const Polynomial& o) const { return basis == o.basis && values == o.values; }

  // impl Rlc for Polynomial (utils.rs:170-175): self + other * beta + gamma
  Polynomial rlc(const Polynomial& other, const Scalar& beta, const Scalar& gamma) const { return *this + other * beta + gamma; }
  Scalar coeffs_evaluate(const Scalar& x, Context& c = Context::global()) const {      // polynomial.rs:34-45
    Scalar out;
    c.check(bp_poly_evaluate(c.raw(), values.data(), values.size(), (int)basis, x.l.data(), BP_FR_MONT, out.l.data()), "coeffs_evaluate");
    return out;
  }
  Polynomial ntt(Context& c = Context::global()) const {                               // polynomial.rs:47-51
    if (basis != Basis::Monomial) throw Panic(BP_ERR_BASIS, "assertion failed: basis == Monomial");
    return Polynomial(ntt_381(values, c), Basis::Lagrange);
  }
  Polynomial i_ntt(Context& c = Context::global()) const {                             // polynomial.rs:52-55
    if (basis != Basis::Lagrange) throw Panic(BP_ERR_BASIS, "assertion failed: basis == Lagrange");
    return Polynomial(i_ntt_381(values, c), Basis::Monomial);
  }
  friend Polynomial operator+(const Polynomial& a, const Polynomial& b) { return a.binop(b, bp_poly_add, "Polynomial + Polynomial"); }
  friend Polynomial operator-(const Polynomial& a, const Polynomial& b) { return a.binop(b, bp_poly_sub, "Polynomial - Polynomial"); }
  friend Polynomial operator*(const Polynomial& a, const Polynomial& b) { return a.binop(b, bp_poly_mul, "Polynomial * Polynomial"); }
  friend Polynomial operator/(const Polynomial& a, const Polynomial& b) { return a.binop(b, bp_poly_div, "Polynomial / Polynomial"); }
  friend Polynomial operator+(const Polynomial& a, const Scalar& s) { return a.scalar(s, 0, "Polynomial + Scalar"); }
  friend Polynomial operator-(const Polynomial& a, const Scalar& s) { return a.scalar(s, 1, "Polynomial - Scalar"); }
  friend Polynomial operator*(const Polynomial& a, const Scalar& s) { return a.scalar(s, 2, "Polynomial * Scalar"); }

 private:
  typedef int (*binfn)(bp_ctx*, const void*, size_t, const void*, size_t, int, int, void*, size_t*);
  Polynomial binop(const Polynomial& o, binfn fn, const char* what) const {
    if (basis != o.basis) throw Panic(BP_ERR_BASIS, "Basis must be the same");
    Context& c = Context::global();
    std::vector<Scalar> out(values.size() + o.values.size() + 1);
    size_t n = 0;
    c.check(fn(c.raw(), values.data(), values.size(), o.values.data(), o.values.size(), (int)basis, BP_FR_MONT, out.data(), &n), what);
    out.resize(n);
    return Polynomial(std::move(out), basis);
  }
  Polynomial scalar(const Scalar& s, int op, const char* what) const {
    Context& c = Context::global();
    std::vector<Scalar> out(values.size());
    c.check(bp_poly_scalar_op(c.raw(), values.data(), values.size(), (int)basis, s.l.data(), op, BP_FR_MONT, out.data()), what);
    return Polynomial(std::move(out), basis);
  }
};

// src/msm.rs:8,76-118
// The reference walks floor(b / c) windows of c bits from the most significant end of the scalar's 256-bit image (msm.rs:83,
// 119-139): b = 256 with c dividing 256 (its only call: setup.rs:36, b = 256, c = 4) uses the whole scalar, any other (b, c) drops
// the low 256 - c * floor(b / c) bits.  bp_msm_window_scalars reproduces that (the reference's panics become Panic).
inline int msm_scalars_as_walked(Context& ctx, const std::vector<Scalar>& scalars, size_t b, size_t c, std::vector<uint8_t>& eff) {
  if (c != 0 && c <= 63 && (b / c) * c == 256) return BP_FR_MONT;       // the whole scalar: hand the Montgomery limbs over as they are
  eff.assign(32 * scalars.size() + 32, 0);
  ctx.check(bp_msm_window_scalars(scalars.data(), scalars.size(), BP_FR_MONT, b, c, eff.data()), "bucket_msm: the reference panics for this (b, c)");
  return BP_FR_BYTES_LE;
}
struct BucketMSM {
  // points: 96-byte encodings
  static G1 bucket_msm(const std::vector<G1>& points, const std::vector<Scalar>& scalars, size_t b = 256, size_t c = 4,
                       Context& ctx = Context::global()) {
    std::vector<uint8_t> eff;
    const int fmt = msm_scalars_as_walked(ctx, scalars, b, c, eff);
    uint64_t h = 0;
    ctx.check(bp_srs_load(ctx.raw(), points.empty() ? nullptr : points[0].data(), points.size(), &h), "bucket_msm: points");
    G1 out{};
    int rc = bp_msm_g1(ctx.raw(), h, fmt == BP_FR_MONT ? (const void*)scalars.data() : (const void*)eff.data(), scalars.size(), fmt, out.data());
    bp_srs_free(ctx.raw(), h);
    ctx.check(rc, "bucket_msm");
    return out;
  }
};

// G1Projective as the reference keeps it in memory (g1.rs:442-446): x | y | z, 6 x u64 Montgomery limbs each
using G1ProjectiveImage = std::array<uint8_t, 144>;
// the literal seam bucket_msm(points: &[G1Projective], scalars: &[Scalar], b, c) (msm.rs:76-81), nothing cached: one call that
// uploads the operands in two pieces and multiplies the first while the second is on its way
inline G1 bucket_msm_projective(const std::vector<G1ProjectiveImage>& points, const std::vector<Scalar>& scalars, size_t b = 256,
                                size_t c = 4, Context& ctx = Context::global()) {
  std::vector<uint8_t> eff;
  const int fmt = msm_scalars_as_walked(ctx, scalars, b, c, eff);
  G1 out{};
  ctx.check(bp_msm_g1_projective144(ctx.raw(), points.empty() ? nullptr : points[0].data(), points.size(),
                                    fmt == BP_FR_MONT ? (const void*)scalars.data() : (const void*)eff.data(), scalars.size(), fmt, out.data()),
            "bucket_msm");
  return out;
}

// src/setup.rs:7-37 (G1 part)
class Setup {
 public:
  // A Setup serves every commitment of a prover: unless tables == false it also builds the SRS's fixed-base
  // window tables (bp_srs_precompute), which every later commit() then uses.
  static Setup generate_srs(size_t powers, const std::array<uint8_t, 32>& tau_le, Context& c = Context::global(),
                            bool tables = true) {                                                                     // setup.rs:12-31
    uint64_t h = 0;
    c.check(bp_srs_generate(c.raw(), powers, tau_le.data(), &h), "generate_srs");
    if (tables) (void)bp_srs_precompute(c.raw(), h, 0);      // an optimisation (windows x 128 B per point of HBM): commits work without it
    return Setup(h, c);
  }
  std::vector<G1> powers_of_x() const {
    size_t n = 0;
    ctx_->check(bp_srs_len(ctx_->raw(), handle_, &n), "srs_len");
    std::vector<G1> out(n);
    if (n) ctx_->check(bp_srs_export(ctx_->raw(), handle_, 0, n, out[0].data()), "srs_export");
    return out;
  }
  G1 commit(const Polynomial& p) const {                                                 // setup.rs:32-37
    G1 out{};
    ctx_->check(bp_commit(ctx_->raw(), handle_, p.values.data(), p.values.size(), (int)p.basis, BP_FR_MONT, out.data()), "commit");
    return out;
  }
  // this rank's point range of a larger SRS (one process per GPU): the points the rank keeps resident
  static Setup from_points(const std::vector<G1>& points, Context& c = Context::global(), bool tables = true) {
    uint64_t h = 0;
    c.check(bp_srs_load(c.raw(), points.empty() ? nullptr : points[0].data(), points.size(), &h), "from_points");
    if (tables) (void)bp_srs_precompute(c.raw(), h, 0);
    return Setup(h, c);
  }
  // Setup::commit over ALL ranks of the context's communicator: `slice` holds the coefficients of this rank's point range; every rank
  // receives the same commitment (bp_msm_g1_allgather: one ncclAllGather of the ranks' partial-sum records under the C ABI)
  G1 commit_over_ranks(const Polynomial& slice) const {
    if (slice.basis != Basis::Monomial) throw Panic(BP_ERR_BASIS, "commit: polynomial not in the Monomial basis (setup.rs:34)");
    G1 out{};
    ctx_->check(bp_msm_g1_allgather(ctx_->raw(), handle_, 0, slice.values.data(), slice.values.size(), BP_FR_MONT, 0, out.data()), "commit_over_ranks");
    return out;
  }

 uint64_t handle() const { return handle_; }
  Context& context() const { return *ctx_; }

 private:
  Setup(uint64_t h, Context& c) : handle_(h), ctx_(&c) {}
  uint64_t handle_;
  Context* ctx_;
};

// src/program.rs:34-50: selector and permutation polynomials in the Lagrange basis
struct CommonPreprocessedInput {
  uint64_t group_order;
  Polynomial ql, qr, qm, qo, qc, s1, s2, s3;
};

// src/verifier.rs:23-40, serialised in field order: a_1 b_1 c_1 z_1 t_lo_1 t_mid_1 t_hi_1 w_zeta_1 w_zeta_omega_1 as
// 48-byte compressed G1 (the encoding the transcript absorbs, transcript.rs:66-69), then a_bar b_bar c_bar s1_bar
// s2_bar z_omega_bar as Scalar::to_bytes()
struct Proof {
  std::array<uint8_t, 624> bytes{};
  std::array<uint8_t, 48> point(int i) const {
    std::array<uint8_t, 48> p{};
    std::memcpy(p.data(), bytes.data() + 48 * i, 48);
    return p;
  }
  std::array<uint8_t, 32> eval_bytes(int i) const {
    std::array<uint8_t, 32> e{};
    std::memcpy(e.data(), bytes.data() + 432 + 32 * i, 32);
    return e;
  }
};

// src/prover.rs:50-175.  The circuit front-end (Program / Assembly) stays on the reference's side; what arrives here
// is what it produces: the preprocessed columns once, and per proof the three wire columns and the public-input column.
class Prover {
 public:
  Prover(const Setup& setup, const CommonPreprocessedInput& pk) : setup_(&setup), n_(pk.group_order) {           // prover.rs:64-104
    const Polynomial* cols[8] = {&pk.ql, &pk.qr, &pk.qm, &pk.qo, &pk.qc, &pk.s1, &pk.s2, &pk.s3};
    const void* ptrs[8];
    uint32_t log_n = 0;
    while ((1ull << log_n) < n_) log_n++;
    if ((1ull << log_n) != n_) throw Panic(BP_ERR_NOT_POW2, "group_order must be a power of two");
    for (int k = 0; k < 8; k++) {
      if (cols[k]->basis != Basis::Lagrange || cols[k]->values.size() != n_) throw Panic(BP_ERR_LENGTH, "preprocessed column: Lagrange, group_order values");
      ptrs[k] = cols[k]->values.data();
    }
    setup.context().check(bp_circuit_load(setup.context().raw(), log_n, ptrs, BP_FR_MONT, 0, &circuit_), "Prover::new");
  }
  ~Prover() { bp_circuit_free(setup_->context().raw(), circuit_); }
  Prover(const Prover&) = delete;
  Prover& operator=(const Prover&) = delete;

  // prover.rs:106-175 with b1..b11 (:110) supplied by the caller; wire columns as built at :186-227, public-input column as at :114-127
  Proof prove_with_blinding(const std::vector<Scalar>& a, const std::vector<Scalar>& b, const std::vector<Scalar>& c,
                            const std::vector<Scalar>& public_input, const std::array<Scalar, 11>& blinders) const {
    if (a.size() != n_ || b.size() != n_ || c.size() != n_ || (!public_input.empty() && public_input.size() != n_))
      throw Panic(BP_ERR_LENGTH, "witness columns must have group_order entries");
    uint8_t bl[352];
    for (int j = 0; j < 11; j++) {
      auto bytes = blinders[j].to_bytes();
      std::memcpy(bl + 32 * j, bytes.data(), 32);
    }
    Proof proof;
    setup_->context().check(bp_prove(setup_->context().raw(), setup_->handle(), circuit_, a.data(), b.data(), c.data(),
                                     public_input.empty() ? nullptr : public_input.data(), BP_FR_MONT, 0, bl, proof.bytes.data()),
                            "Prover::prove");
    return proof;
  }

 private:
  const Setup* setup_;
  uint64_t n_;
  uint64_t circuit_ = 0;
};

}  // namespace baby_plonk
