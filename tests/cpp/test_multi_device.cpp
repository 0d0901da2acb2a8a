// Single-process multi-GPU test of the C ABI (bp_init_multi): one context over a device list commits through the
// point-range-sharded SRS and must produce the bytes of the single-device context -- Setup::commit (src/setup.rs:32-37)
// of an unmodified, single-threaded caller spanning several GPUs.  argv[1] = comma-separated device list (default "0,0":
// two shards on one GPU, which is what a 1-GPU box can run).  Built and run by tests/test_gpu_multi_device.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../baby_plonk_rust_amd/host/baby_plonk.hpp"

using namespace baby_plonk;

#define CHECK(cond)                                                         \
  do {                                                                      \
    if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); std::exit(1); } \
  } while (0)

static std::array<uint8_t, 32> le(uint64_t v) {
  std::array<uint8_t, 32> b{};
  for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i));
  return b;
}
// deterministic pseudo-random canonical scalars (SplitMix64, top bits cleared so the value is < q)
static std::vector<Scalar> scalars(size_t n, uint64_t seed) {
  std::vector<Scalar> out(n);
  uint64_t x = seed;
  auto next = [&] {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  };
  for (size_t i = 0; i < n; i++) {
    std::array<uint8_t, 32> b{};
    for (int w = 0; w < 4; w++) {
      uint64_t v = next();
      if (w == 3) v &= 0x3fffffffffffffffull;
      std::memcpy(b.data() + 8 * w, &v, 8);
    }
    out[i] = Scalar::from_bytes(b);
  }
  return out;
}

int main(int argc, char** argv) {
  std::vector<int> devices;
  std::string list = argc > 1 ? argv[1] : "0,0";
  for (size_t p = 0; p < list.size();) {
    size_t q = list.find(',', p);
    if (q == std::string::npos) q = list.size();
    devices.push_back(std::atoi(list.substr(p, q - p).c_str()));
    p = q + 1;
  }
  Context one(devices[0]);
  Context many(devices);
  CHECK(many.shards() == (int)devices.size() && one.shards() == 1);

  for (size_t n : {1u, 5u, 1000u, 20000u}) {                 // fewer points than shards, ragged shards, a table-sized SRS
    const bool tables = n >= 1000;
    Setup s1 = Setup::generate_srs(n, le(0x1234567), one, tables);
    Setup sm = Setup::generate_srs(n, le(0x1234567), many, tables);
    CHECK(s1.powers_of_x() == sm.powers_of_x());              // the shards hold tau^i G for their own ranges
    for (size_t len : {n, n / 2 + 1, n + 3}) {                 // zip truncation on both sides (msm.rs:29)
      Polynomial p(scalars(len, 77 + n + len), Basis::Monomial);
      CHECK(s1.commit(p) == sm.commit(p));
    }
    // re-upload through the 96-byte seam and through the G1Projective-image seam: same commitment
    std::vector<G1> pts = s1.powers_of_x();
    std::vector<Scalar> sc = scalars(n, 5);
    G1 want = s1.commit(Polynomial(sc, Basis::Monomial));
    CHECK(BucketMSM::bucket_msm(pts, sc, 256, 4, many) == want);
    std::vector<G1ProjectiveImage> proj(n);
    for (size_t i = 0; i < n; i++) CHECK(bp_g1_bytes96_to_partial(pts[i].data(), proj[i].data()) == BP_OK);
    CHECK(bucket_msm_projective(proj, sc, 256, 4, many) == want);
    CHECK(bucket_msm_projective(proj, sc, 256, 4, one) == want);
  }
  // a scalar >= q in canonical-bytes input is refused by whichever shard sees it
  {
    Setup sm = Setup::generate_srs(64, le(3), many, false);
    std::vector<uint8_t> bad(64 * 32, 0);
    std::memset(bad.data() + 63 * 32, 0xff, 32);
    G1 out{};
    CHECK(bp_msm_g1(many.raw(), sm.handle(), bad.data(), 64, BP_FR_BYTES_LE, out.data()) == BP_ERR_BAD_SCALAR);
  }
  // batched host NTT: independent columns spread over the shards
  {
    const size_t N = 1 << 10, batch = 5;
    std::vector<Scalar> cols = scalars(N * batch, 9), a = cols, b = cols;
    one.check(bp_ntt_fr(one.raw(), a.data(), 10, 0, BP_FR_MONT, batch, N), "ntt one");
    many.check(bp_ntt_fr(many.raw(), b.data(), 10, 0, BP_FR_MONT, batch, N), "ntt many");
    CHECK(a == b);
    many.check(bp_ntt_fr(many.raw(), b.data(), 10, 1, BP_FR_MONT, batch, N), "intt many");
    CHECK(b == cols);
  }
  std::printf("multi device ok (%zu shards)\n", devices.size());
  return 0;
}
