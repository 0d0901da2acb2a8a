"""Merlin transcripts (merlin 3.0.0, Cargo.lock:384-386 of the reference) restated from the published specification:
STROBE-128 over Keccak-f[1600] (rate 166), framing  meta-AD(label) || meta-AD(len_le32, more) || AD / PRF.
The crate's source is not in the reference tree; this restatement is pinned by merlin's published conformance vector
(tests/test_merlin_transcript.py).  TEST INFRASTRUCTURE: the transcript is host-side protocol glue (about 20 Keccak-f
calls per proof), out of the product's scope, needed only to derive the reference's Fiat-Shamir challenges
(src/transcript.rs:4-86) for end-to-end proof bytes."""

_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
       0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
       0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
       0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M = (1 << 64) - 1


def _rol(v, n):
    return ((v << n) | (v >> (64 - n))) & _M if n else v


def keccak_f1600(state):
    """state: bytearray(200), in place"""
    a = [[int.from_bytes(state[8 * (x + 5 * y): 8 * (x + 5 * y) + 8], "little") for y in range(5)] for x in range(5)]
    for rc in _RC:
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _ROT[x][y])
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= rc
    for x in range(5):
        for y in range(5):
            state[8 * (x + 5 * y): 8 * (x + 5 * y) + 8] = a[x][y].to_bytes(8, "little")


_R = 166
_I, _A, _C, _T, _MF, _K = 1, 2, 4, 8, 16, 32


class Strobe128:
    def __init__(self, protocol_label):
        st = bytearray(200)
        st[0:6] = bytes([1, _R + 2, 1, 0, 1, 96])
        st[6:18] = b"STROBEv1.0.2"
        keccak_f1600(st)
        self.state, self.pos, self.pos_begin, self.cur_flags = st, 0, 0, 0
        self.meta_ad(protocol_label, False)

    def _run_f(self):
        self.state[self.pos] ^= self.pos_begin
        self.state[self.pos + 1] ^= 0x04
        self.state[_R + 1] ^= 0x80
        keccak_f1600(self.state)
        self.pos = self.pos_begin = 0

    def _absorb(self, data):
        for byte in data:
            self.state[self.pos] ^= byte
            self.pos += 1
            if self.pos == _R:
                self._run_f()

    def _squeeze(self, n):
        out = bytearray()
        for _ in range(n):
            out.append(self.state[self.pos])
            self.state[self.pos] = 0
            self.pos += 1
            if self.pos == _R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags, more):
        if more:
            assert self.cur_flags == flags
            return
        assert flags & _T == 0
        old_begin = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old_begin, flags]))
        if flags & (_C | _K) and self.pos != 0:
            self._run_f()

    def meta_ad(self, data, more):
        self._begin_op(_MF | _A, more)
        self._absorb(data)

    def ad(self, data, more):
        self._begin_op(_A, more)
        self._absorb(data)

    def prf(self, n, more):
        self._begin_op(_I | _A | _C, more)
        return self._squeeze(n)


class Transcript:
    """merlin::Transcript"""

    def __init__(self, label):
        self.strobe = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def append_message(self, label, message):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(len(message).to_bytes(4, "little"), True)
        self.strobe.ad(message, False)

    def challenge_bytes(self, label, n):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(n.to_bytes(4, "little"), True)
        return self.strobe.prf(n, False)


Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


class PlonkTranscript:
    """src/transcript.rs:4-86, including its label quirk (alpha is drawn under the label "z_1", :24).
    Points are absorbed as 48-byte compressed encodings (:66-69), scalars as 32-byte little-endian (:83-85);
    challenges are rejection-sampled until canonical and non-zero and then re-absorbed (:70-82)."""

    def __init__(self):
        self.t = Transcript(b"plonk")                       # prover.rs:112, verifier.rs:194

    def append_point(self, label, compressed48):
        self.t.append_message(label, compressed48)

    def append_scalar(self, label, value):
        self.t.append_message(label, (value % Q).to_bytes(32, "little"))

    def get_and_append_challenge(self, label):
        while True:
            b = self.t.challenge_bytes(label, 32)
            v = int.from_bytes(b, "little")
            if v < Q and v != 0:
                self.t.append_message(label, b)
                return v

    def round_1(self, a_1, b_1, c_1):
        self.append_point(b"a_1", a_1); self.append_point(b"b_1", b_1); self.append_point(b"c_1", c_1)
        return self.get_and_append_challenge(b"beta"), self.get_and_append_challenge(b"gamma")

    def round_2(self, z_1):
        self.append_point(b"z_1", z_1)
        return self.get_and_append_challenge(b"z_1")

    def round_3(self, t_lo_1, t_mid_1, t_hi_1):
        self.append_point(b"t_lo_1", t_lo_1); self.append_point(b"t_mid_1", t_mid_1); self.append_point(b"t_hi_1", t_hi_1)
        return self.get_and_append_challenge(b"zeta")

    def round_4(self, a_bar, b_bar, c_bar, s1_bar, s2_bar, z_omega_bar):
        for label, v in ((b"a_eval", a_bar), (b"b_eval", b_bar), (b"c_eval", c_bar), (b"s1_eval", s1_bar), (b"s2_eval", s2_bar),
                         (b"z_shifted_eval", z_omega_bar)):
            self.append_scalar(label, v)
        return self.get_and_append_challenge(b"nu")

    def round_5(self, w_zeta_1, w_zeta_omega_1):
        self.append_point(b"w_zeta_1", w_zeta_1); self.append_point(b"w_zeta_omega_1", w_zeta_omega_1)
        return self.get_and_append_challenge(b"mu")
