// transcript.hpp -- host-side Fiat-Shamir transcript of the prover (src/transcript.rs:4-86).
//
// The reference uses merlin 3.0.0 (Cargo.lock:384-386), which is not in the reference tree; this is a restatement of
// merlin's published construction: STROBE-128 over Keccak-f[1600] (rate 166) with the framing
//   meta-AD(label) || meta-AD(len_le32, more) || AD(message)      /      ... || PRF(n)
// pinned by merlin's published conformance vector (tests/test_native_prover.py through bp_transcript_test_vector, and
// tests/test_merlin_transcript.py for the Python twin the parity tests use).  About 20 Keccak-f calls per proof.
#pragma once
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

namespace bp {

inline void keccak_f1600(uint64_t a[25]) {
  static const uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
                                  0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
                                  0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
                                  0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
                                  0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
                                  0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
  static const int ROT[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};
  auto rol = [](uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; };
  for (int round = 0; round < 24; round++) {
    uint64_t c[5], d[5], b[25];
    for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
    for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(a[x + 5 * y] ^ d[x], ROT[x][y]);      // rho + pi
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);   // chi
    a[0] ^= RC[round];
  }
}

class Strobe128 {
 public:
  explicit Strobe128(const std::string& protocol_label) {
    memset(st_, 0, sizeof st_);
    const uint8_t head[6] = {1, R + 2, 1, 0, 1, 96};
    memcpy(st_, head, 6);
    memcpy(st_ + 6, "STROBEv1.0.2", 12);
    permute();
    meta_ad((const uint8_t*)protocol_label.data(), protocol_label.size(), false);
  }
  void meta_ad(const uint8_t* data, size_t n, bool more) {
    begin_op(FLAG_M | FLAG_A, more);
    absorb(data, n);
  }
  void ad(const uint8_t* data, size_t n, bool more) {
    begin_op(FLAG_A, more);
    absorb(data, n);
  }
  void prf(uint8_t* out, size_t n, bool more) {
    begin_op(FLAG_I | FLAG_A | FLAG_C, more);
    for (size_t i = 0; i < n; i++) {
      out[i] = st_[pos_];
      st_[pos_] = 0;
      if (++pos_ == R) run_f();
    }
  }

 private:
  static constexpr int R = 166;
  static constexpr uint8_t FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32;
  uint8_t st_[200];
  int pos_ = 0, pos_begin_ = 0;
  uint8_t cur_flags_ = 0;

  void permute() {
    uint64_t lanes[25];
    for (int i = 0; i < 25; i++) {
      uint64_t v = 0;
      for (int j = 7; j >= 0; j--) v = (v << 8) | st_[8 * i + j];
      lanes[i] = v;
    }
    keccak_f1600(lanes);
    for (int i = 0; i < 25; i++)
      for (int j = 0; j < 8; j++) st_[8 * i + j] = (uint8_t)(lanes[i] >> (8 * j));
  }
  void run_f() {
    st_[pos_] ^= (uint8_t)pos_begin_;
    st_[pos_ + 1] ^= 0x04;
    st_[R + 1] ^= 0x80;
    permute();
    pos_ = pos_begin_ = 0;
  }
  void absorb(const uint8_t* data, size_t n) {
    for (size_t i = 0; i < n; i++) {
      st_[pos_] ^= data[i];
      if (++pos_ == R) run_f();
    }
  }
  void begin_op(uint8_t flags, bool more) {
    if (more) return;                               // continuation of the current operation (same flags by construction)
    const uint8_t old_begin = (uint8_t)pos_begin_;
    pos_begin_ = pos_ + 1;
    cur_flags_ = flags;
    const uint8_t frame[2] = {old_begin, flags};
    absorb(frame, 2);
    if ((flags & (FLAG_C | FLAG_K)) && pos_ != 0) run_f();
  }
};

// merlin::Transcript
class MerlinTranscript {
 public:
  explicit MerlinTranscript(const std::string& label) : strobe_("Merlin v1.0") { append_message("dom-sep", (const uint8_t*)label.data(), label.size()); }
  void append_message(const char* label, const uint8_t* msg, size_t n) {
    const uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    strobe_.meta_ad((const uint8_t*)label, strlen(label), false);
    strobe_.meta_ad(len, 4, true);
    strobe_.ad(msg, n, false);
  }
  void challenge_bytes(const char* label, uint8_t* out, size_t n) {
    const uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    strobe_.meta_ad((const uint8_t*)label, strlen(label), false);
    strobe_.meta_ad(len, 4, true);
    strobe_.prf(out, n, false);
  }

 private:
  Strobe128 strobe_;
};

}  // namespace bp
