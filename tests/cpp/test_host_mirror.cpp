// C++ parity test of the host-side mirror (baby_plonk_rust_amd/host/baby_plonk.hpp): restates the reference's own
// unit tests for the path -- src/setup.rs:46-116 (test_generate_srs, test_monomial_commit), src/polynomial.rs:386-521,
// src/utils.rs:239-242 -- through the mirror's Rust-shaped API.  Needs a GPU; built and run by tests/test_gpu_cpp_mirror.py.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../baby_plonk_rust_amd/host/baby_plonk.hpp"

using namespace baby_plonk;

#define CHECK(cond)                                                         \
  do {                                                                      \
    if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); std::exit(1); } \
  } while (0)

static Scalar S(uint64_t v) { return Scalar::from_u64(v); }
static Scalar neg(const Scalar& a) {                     // 0 - a through Polynomial - Polynomial
  Polynomial z({Scalar::zero()}, Basis::Monomial), p({a}, Basis::Monomial);
  return (z - p).values[0];
}
static Polynomial mono(std::vector<Scalar> v) { return Polynomial(std::move(v), Basis::Monomial); }
static std::array<uint8_t, 32> le(uint64_t v) {
  std::array<uint8_t, 32> b{};
  for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i));
  return b;
}
template <class F>
static bool panics(F f, int code) {
  try { f(); } catch (const Panic& p) { return p.code == code; }
  return false;
}

static Scalar mul(const Scalar& a, const Scalar& b) { return (mono({a}) * b).values[0]; }

// tests/verify_proof_test.rs:16-44 through the mirror: program ["e public", "c <== a * b + b", "e <== c * d"], group order 8,
// 14 powers of tau = 101, witness a=3 b=4 c=16 d=5 e=80.  The columns are what Program / Assembly produce (SURVEY.md
// appendix A): rows (e,-,-) (a,b,c) (c,d,e) + five empty rows; cells with equal names form one permutation cycle.
static Proof toy_proof(const std::array<Scalar, 11>& blinders) {
  const uint64_t n = 8;
  const char* wires[8][3] = {{"e", "", ""}, {"a", "b", "c"}, {"c", "d", "e"}, {"", "", ""}, {"", "", ""}, {"", "", ""}, {"", "", ""}, {"", "", ""}};
  auto value = [](const std::string& w) -> uint64_t { return w == "a" ? 3 : w == "b" ? 4 : w == "c" ? 16 : w == "d" ? 5 : w == "e" ? 80 : 0; };
  auto lag = [&](std::vector<Scalar> v) { v.resize(n, Scalar::zero()); return Polynomial(std::move(v), Basis::Lagrange); };
  CommonPreprocessedInput pk{n, lag({S(1)}), lag({S(0), neg(S(1))}), lag({S(0), neg(S(1)), neg(S(1))}), lag({S(0), S(1), S(1)}), lag({}),
                             lag({}), lag({}), lag({})};
  auto roots = roots_of_unity(n);
  std::vector<std::string> names;
  for (auto& row : wires) for (auto* w : row) if (std::find(names.begin(), names.end(), w) == names.end()) names.push_back(w);
  Polynomial* sigma[3] = {&pk.s1, &pk.s2, &pk.s3};
  for (auto& name : names) {                                    // program.rs:92-99
    std::vector<std::pair<int, int>> cells;                     // (column, row) in row-major order
    for (int row = 0; row < 8; row++) for (int col = 0; col < 3; col++) if (name == wires[row][col]) cells.push_back({col, row});
    for (size_t j = 0; j < cells.size(); j++) {
      auto next = cells[(j + 1) % cells.size()];
      sigma[next.first]->values[next.second] = mul(S(cells[j].first + 1), roots[cells[j].second]);     // s[next] = label(cell), program.rs:126-137, utils.rs:29-36
    }
  }
  std::vector<Scalar> col[3];
  for (int j = 0; j < 3; j++) for (int row = 0; row < 8; row++) col[j].push_back(S(value(wires[row][j])));
  std::vector<Scalar> pi(n, Scalar::zero());
  pi[0] = neg(S(80));                                            // prover.rs:114-127
  Setup setup = Setup::generate_srs(n + 6, le(101));
  Prover prover(setup, pk);
  Proof proof = prover.prove_with_blinding(col[0], col[1], col[2], pi, blinders);
  // a witness that breaks the gate is refused the way the reference panics (prover.rs:615)
  std::vector<Scalar> bad = col[0];
  bad[1] = S(4);
  bool refused = false;
  try { prover.prove_with_blinding(bad, col[1], col[2], pi, blinders); } catch (const Panic& p) { refused = p.code == BP_ERR_ASSERT; }
  CHECK(refused);
  return proof;
}

int main(int argc, char** argv) {
  if (argc == 2) {                       // 11 blinders as 11 x 64 hex digits (32-byte little-endian each): print the toy proof
    std::array<Scalar, 11> blinders;
    std::string hex = argv[1];
    CHECK(hex.size() == 11 * 64);
    for (int j = 0; j < 11; j++) {
      std::array<uint8_t, 32> b{};
      for (int i = 0; i < 32; i++) b[i] = (uint8_t)std::stoul(hex.substr(64 * j + 2 * i, 2), nullptr, 16);
      blinders[j] = Scalar::from_bytes(b);
    }
    Proof p = toy_proof(blinders);
    std::printf("proof ");
    for (uint8_t v : p.bytes) std::printf("%02x", v);
    std::printf("\n");
    return 0;
  }
  // ---- src/utils.rs:239-242 test_root_of_unity
  {
    Polynomial w({root_of_unity(4)}, Basis::Monomial);
    Polynomial w2 = w * w, w4 = w2 * w2;
    CHECK(w4.values[0] == S(1));
    CHECK(w2.values[0] != S(1));
    auto roots = roots_of_unity(8);
    CHECK(roots.size() == 8 && roots[0] == S(1) && roots[1] == root_of_unity(8));
  }
  // ---- src/polynomial.rs tests
  {
    CHECK((mono({S(1), S(2), S(3)}) + mono({S(4), S(5), S(6)})) == mono({S(5), S(7), S(9)}));
    CHECK((mono({S(1), S(2), S(3)}) + mono({S(4), S(5)})) == mono({S(5), S(7), S(3)}));
    CHECK((mono({S(4), S(5), S(6)}) - mono({S(1), S(2)})) == mono({S(3), S(3), S(6)}));
    CHECK((mono({S(1), S(2), S(3)}) * S(2)) == mono({S(2), S(4), S(6)}));
    CHECK((mono({S(1), S(1)}) * mono({S(1), S(1)})) == mono({S(1), S(2), S(1)}));                  // (1+x)^2, :437-451
    CHECK(mono({S(1), S(3), S(2)}).coeffs_evaluate(S(2)) == S(15));
    CHECK(mono({S(1), S(2)}).rlc(mono({S(3), S(4)}), S(3), S(4)) == mono({S(14), S(14)}));          // utils.rs:170-175: 1 + 3*3 + 4, 2 + 4*3
    // (3x^3 - x^2 - x - 1) / (x - 1) = 3x^2 + 2x + 1
    Polynomial num = mono({neg(S(1)), neg(S(1)), neg(S(1)), S(3)}), den = mono({neg(S(1)), S(1)});
    CHECK((num / den) == mono({S(1), S(2), S(3)}));
    CHECK(panics([&] { Polynomial({S(1), S(2)}, Basis::Lagrange) / Polynomial({S(1), S(1)}, Basis::Lagrange); }, BP_ERR_BASIS));   // :523-547
    CHECK(panics([&] { mono({S(1)}) / mono({Scalar::zero()}); }, BP_ERR_DIV_ZERO));
    CHECK(panics([&] { Polynomial({S(1), S(2), S(3)}, Basis::Lagrange) + Polynomial({S(1)}, Basis::Lagrange); }, BP_ERR_LENGTH));
    CHECK(panics([&] { ntt_381({S(1), S(2), S(3)}); }, BP_ERR_NOT_POW2));                         // utils.rs:65
    std::vector<Scalar> v = {S(3), S(3)};
    CHECK(i_ntt_381(ntt_381(v)) == v);                                                            // setup.rs:128-135
    CHECK(ntt_381(v) == (std::vector<Scalar>{S(6), S(0)}));
    CHECK(Scalar::from_bytes(S(77).to_bytes()) == S(77));
  }
  // ---- src/setup.rs:46-57 test_generate_srs and :60-116 test_monomial_commit
  {
    Setup s2 = Setup::generate_srs(8, le(2));
    auto pts = s2.powers_of_x();
    CHECK(pts.size() == 8);
    // powers_of_x[i] == G * tau^i: compare with a one-point MSM of G by tau^i
    for (uint64_t i = 0; i < 8; i++) CHECK(BucketMSM::bucket_msm({pts[0]}, {S(1ull << i)}) == pts[i]);
    Setup s10 = Setup::generate_srs(2, le(10));
    G1 commitment = s10.commit(mono({S(2), S(3)}));
    CHECK(commitment == BucketMSM::bucket_msm({pts[0]}, {S(32)}));                                 // 2 g1 + 3 * 10 g1
    CHECK(s2.commit(mono({S(0), S(1)})) == pts[1]);                                                // x -> tau G
    CHECK(s2.commit(mono({S(0), S(0), S(1)})) == pts[2]);
    // zip truncation (msm.rs:29): more scalars than points, fewer scalars than points
    CHECK(BucketMSM::bucket_msm({pts[0], pts[1]}, {S(5), S(7), S(9)}) == BucketMSM::bucket_msm({pts[0]}, {S(19)}));
    CHECK(BucketMSM::bucket_msm({pts[0], pts[1], pts[2]}, {S(5)}) == BucketMSM::bucket_msm({pts[0]}, {S(5)}));
    CHECK(panics([&] { s2.commit(Polynomial({S(1)}, Basis::Lagrange)); }, BP_ERR_BASIS));          // setup.rs:34
    // window parameters that make the reference drop low scalar bits (msm.rs:83,119-139): 13 = 0b1101 walked as 51 windows of 5 bits
    // loses its lowest bit (6), as 32 windows of 4 bits over b = 128 loses everything below bit 128 (0); panicking ones panic
    CHECK(BucketMSM::bucket_msm({pts[0]}, {S(13)}, 256, 5) == BucketMSM::bucket_msm({pts[0]}, {S(6)}));
    CHECK(BucketMSM::bucket_msm({pts[0]}, {S(13)}, 128, 4) == BucketMSM::bucket_msm({pts[0]}, {S(0)}));
    CHECK(BucketMSM::bucket_msm({pts[0]}, {S(13)}, 256, 8) == BucketMSM::bucket_msm({pts[0]}, {S(13)}));
    CHECK(panics([&] { BucketMSM::bucket_msm({pts[0]}, {S(13)}, 256, 0); }, BP_ERR_INVALID_ARG));
    CHECK(panics([&] { BucketMSM::bucket_msm({pts[0]}, {S(13)}, 300, 4); }, BP_ERR_INVALID_ARG));
  }
  // ---- one process per GPU through the library's own collective: a world of ONE rank is what a one-GPU box runs.
  // The rank holds the whole SRS as "its point range"; the commitment over the ranks must be the commitment.
  {
    Context rank_ctx(0);                                                                           // a plain context of its own: one communicator per context
    Setup s2 = Setup::generate_srs(8, le(2));
    auto pts = s2.powers_of_x();
    CHECK(rank_ctx.world() == 0);
    CHECK(panics([&] { Setup::from_points(pts, rank_ctx).commit_over_ranks(mono({S(1)})); }, BP_ERR_INVALID_ARG));   // no communicator yet
    rank_ctx.join_ranks(Context::unique_id(), 0, 1);
    CHECK(rank_ctx.world() == 1);
    Setup shard = Setup::from_points(pts, rank_ctx);
    Polynomial f = mono({S(3), S(1), S(4), S(1), S(5), S(9), S(2), S(6)});
    CHECK(shard.commit_over_ranks(f) == s2.commit(f));
    CHECK(shard.commit_over_ranks(mono({S(0), S(1)})) == pts[1]);
    CHECK(panics([&] { shard.commit_over_ranks(Polynomial({S(1)}, Basis::Lagrange)); }, BP_ERR_BASIS));
    rank_ctx.leave_ranks();
    CHECK(rank_ctx.world() == 0);
  }
  std::printf("host mirror ok\n");
  return 0;
}
