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
#pragma once
#include <array>
#include <cstdint>
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
  ~Context() { bp_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  bp_ctx* raw() const { return ctx_; }
  void check(int rc, const char* where) const {
    if (rc != BP_OK) throw Panic(rc, std::string(where) + ": " + bp_last_error(ctx_));
  }
  static Context& global() {
    static Context c(0);
    return c;
  }

 private:
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
struct BucketMSM {
  // points: 96-byte encodings; (b, c) kept for signature parity, the group element does not depend on them
  static G1 bucket_msm(const std::vector<G1>& points, const std::vector<Scalar>& scalars, size_t /*b*/ = 256, size_t /*c*/ = 4,
                       Context& ctx = Context::global()) {
    uint64_t h = 0;
    ctx.check(bp_srs_load(ctx.raw(), points.empty() ? nullptr : points[0].data(), points.size(), &h), "bucket_msm: points");
    G1 out{};
    int rc = bp_msm_g1(ctx.raw(), h, scalars.data(), scalars.size(), BP_FR_MONT, out.data());
    bp_srs_free(ctx.raw(), h);
    ctx.check(rc, "bucket_msm");
    return out;
  }
};

// src/setup.rs:7-37 (G1 part)
class Setup {
 public:
  // A Setup serves every commitment of a prover: unless tables == false it also builds the SRS's fixed-base
  // window tables (bp_srs_precompute), which every later commit() then uses.
  static Setup generate_srs(size_t powers, const std::array<uint8_t, 32>& tau_le, Context& c = Context::global(),
                            bool tables = true) {                                                                     // setup.rs:12-31
    uint64_t h = 0;
    c.check(bp_srs_generate(c.raw(), powers, tau_le.data(), &h), "generate_srs");
    if (tables) c.check(bp_srs_precompute(c.raw(), h, 0), "generate_srs: tables");
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

 private:
  Setup(uint64_t h, Context& c) : handle_(h), ctx_(&c) {}
  uint64_t handle_;
  Context* ctx_;
};

}  // namespace baby_plonk
