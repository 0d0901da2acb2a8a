"""Pins the from-spec Merlin restatement (tests/merlin_transcript.py): Keccak-f[1600] zero-state lane and merlin's
published conformance vector (merlin/src/transcript.rs test `equivalence_simple`)."""
from tests.merlin_transcript import Transcript, keccak_f1600


def test_keccak_f1600_zero_state():
    st = bytearray(200)
    keccak_f1600(st)
    assert int.from_bytes(st[0:8], "little") == 0xF1258F7940E1DDE7


def test_merlin_conformance_vector():
    t = Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
