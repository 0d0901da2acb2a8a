"""bp_prove -- the native prover (csrc/prover.hip: rounds 1-5 on the GPU, Fiat-Shamir transcript on the host) against
the reference-shaped restatement of src/prover.rs in tests/prover_rounds.py run on the CPU oracle, the committed golden
proof of the toy circuit of tests/verify_proof_test.rs, and the verifier's final equation (checked in G1 with the known tau).
The native path computes the same polynomials by a different route (coset quotient, fused linear combinations), so equal
proof BYTES are the parity statement."""
import hashlib
import os
import random

import numpy as np
import pytest

import baby_plonk_rust_amd as bp
from tests import bigint_model as M
from tests import prover_rounds as PR

Q = M.Q
HERE = os.path.dirname(__file__)
MERLIN_VECTOR = "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


def test_host_transcript_conformance_vector():
    """merlin's published test vector through the library's C++ transcript (no GPU): pins transcript.hpp like
    tests/test_merlin_transcript.py pins its Python twin"""
    assert bp.transcript_test_vector().hex() == MERLIN_VECTOR


def _split(blob):
    names = ("a_1", "b_1", "c_1", "z_1", "t_lo_1", "t_mid_1", "t_hi_1", "w_zeta_1", "w_zeta_omega_1")
    pts = {k: M.dec48(blob[48 * i: 48 * i + 48]) for i, k in enumerate(names)}
    ev = {k: int.from_bytes(blob[432 + 32 * i: 464 + 32 * i], "little")
          for i, k in enumerate(("a_bar", "b_bar", "c_bar", "s1_bar", "s2_bar", "z_omega_bar"))}
    return pts, ev


def _challenges(blob):
    from tests.merlin_transcript import PlonkTranscript
    tr = PlonkTranscript()
    pt = lambda i: blob[48 * i: 48 * i + 48]
    ev = [int.from_bytes(blob[432 + 32 * i: 464 + 32 * i], "little") for i in range(6)]
    beta, gamma = tr.round_1(pt(0), pt(1), pt(2))
    alpha = tr.round_2(pt(3))
    zeta = tr.round_3(pt(4), pt(5), pt(6))
    nu = tr.round_4(*ev)
    mu = tr.round_5(pt(7), pt(8))
    return dict(beta=beta, gamma=gamma, alpha=alpha, zeta=zeta, nu=nu, mu=mu)


@pytest.mark.gpu
def test_toy_circuit_native_proof_is_the_golden_proof():
    from oracle import oracle as O
    from tests.test_gpu_prover_rounds import g1_only_verify, prove_with_blinding, toy_circuit, decode
    n, tau = 8, 101
    cols, pk, public = toy_circuit(n)
    blinders = [random.Random(99).randrange(1, Q) for _ in range(11)]
    setup = bp.Setup.generate_srs(n + 6, tau)                          # verify_proof_test.rs:16
    circuit = bp.Circuit({k: PR.SV(v) for k, v in pk.items()})
    prover = bp.Prover(setup, circuit)
    blob = prover.prove_with_blinding(PR.SV(cols[0]), PR.SV(cols[1]), PR.SV(cols[2]), PR.SV(public), blinders)
    golden = open(os.path.join(HERE, "golden", "toy_proof_blinders_seed99.sha256")).read().strip()
    assert len(blob) == 624 and hashlib.sha256(blob).hexdigest() == golden      # fixture made by the oracle path on the CPU
    # the restatement of the reference's call sequence on the CPU oracle gives the same bytes
    cpu = PR.OracleBackend(O.proj_from_bytes96(setup.powers_of_x()))
    _, _, blob_c = prove_with_blinding(cpu, n, cols, pk, public, blinders)
    assert blob == blob_c
    # verifier (src/verifier.rs:80-209), challenges recomputed from the proof bytes
    gpu = PR.GpuBackend(setup)
    vk = {k: decode(gpu.commit(gpu.Polynomial(gpu.i_ntt_381(PR.SV(pk[k])), gpu.MONO))) for k in pk}
    assert {k: decode(v) for k, v in circuit.commitments(setup).items()} == vk      # Verifier::new through the library
    pts, ev = _split(blob)
    assert g1_only_verify(n, tau, pts, ev, _challenges(blob), vk, [80])
    assert not g1_only_verify(n, tau, pts, ev, _challenges(blob), vk, [81])
    # canonical-bytes input format and a different blinding
    st = prover.last_stats()
    assert len(st["round_ms"]) == 5 and st["total_ms"] > 0
    other = prover.prove_with_blinding(PR.SV(cols[0]), PR.SV(cols[1]), PR.SV(cols[2]), PR.SV(public), [b + 1 for b in blinders])
    assert other != blob and g1_only_verify(n, tau, *_split(other), _challenges(other), vk, [80])
    circuit.free()


@pytest.mark.gpu
def test_witness_that_does_not_satisfy_the_circuit_is_rejected():
    """the reference panics: assert z[n] == 1 (prover.rs:319) for a broken copy constraint, r(zeta) != 0 (:615) for a broken gate"""
    from tests.test_gpu_prover_rounds import toy_circuit
    n = 8
    cols, pk, public = toy_circuit(n)
    blinders = list(range(1, 12))
    setup = bp.Setup.generate_srs(n + 6, 101)
    circuit = bp.Circuit({k: PR.SV(v) for k, v in pk.items()})
    prover = bp.Prover(setup, circuit)
    good = [PR.SV(c) for c in cols]
    assert len(prover.prove_with_blinding(*good, PR.SV(public), blinders)) == 624
    bad_gate = [list(c) for c in cols]
    bad_gate[0][1], bad_gate[1][1] = 4, 3                  # a*b + b with a, b swapped: 16 != 15, both cells are free wires
    with pytest.raises(bp.BpError) as e:
        prover.prove_with_blinding(*[PR.SV(c) for c in bad_gate], PR.SV(public), blinders)
    assert e.value.code == -11
    bad_copy = [list(c) for c in cols]
    bad_copy[0][2] = 17                                     # row 2 reads c = 17 while row 1 wrote c = 16
    with pytest.raises(bp.BpError) as e:
        prover.prove_with_blinding(*[PR.SV(c) for c in bad_copy], PR.SV(public), blinders)
    assert e.value.code == -11
    with pytest.raises(bp.BpError) as e:                    # wrong public input: PI does not cancel e
        prover.prove_with_blinding(*good, PR.SV([(-81) % Q] + [0] * (n - 1)), blinders)
    assert e.value.code == -11
    # an SRS that is too short does not panic in the reference: Setup::commit zips and truncates (msm.rs:29); same bytes here
    from oracle import oracle as O
    from tests.test_gpu_prover_rounds import prove_with_blinding
    short = bp.Setup.generate_srs(n + 3, 101)
    cpu = PR.OracleBackend(O.proj_from_bytes96(short.powers_of_x()))
    assert bp.Prover(short, circuit).prove_with_blinding(*good, PR.SV(public), blinders) == prove_with_blinding(cpu, n, cols, pk, public, blinders)[2]
    with pytest.raises(bp.BpError):
        prover.prove_with_blinding(*good, PR.SV(public), blinders[:10])
    with pytest.raises(bp.BpError):
        bp.Circuit([PR.SV([0] * 8)] * 7)
    circuit.free()
    with pytest.raises(bp.BpError):
        prover.prove_with_blinding(*good, PR.SV(public), blinders)          # freed circuit handle


@pytest.mark.gpu
@pytest.mark.parametrize("logn", [4, 10])
def test_synthetic_circuit_native_vs_oracle_restatement(logn, monkeypatch):
    from oracle import oracle as O
    from tests.test_gpu_prover_rounds import decode, g1_only_verify, prove_with_blinding, synthetic_circuit
    n, tau = 1 << logn, 0x1234567
    cols, pk, public = synthetic_circuit(n, 5 + logn)
    blinders = [random.Random(7).randrange(1, Q) for _ in range(11)]
    setup = bp.Setup.generate_srs(n + 6, tau)
    circuit = bp.Circuit({k: PR.SV(v) for k, v in pk.items()})
    prover = bp.Prover(setup, circuit)
    blob = prover.prove_with_blinding(PR.SV(cols[0]), PR.SV(cols[1]), PR.SV(cols[2]), None, blinders)
    cpu = PR.OracleBackend(O.proj_from_bytes96(setup.powers_of_x()))
    cpu.threads = 16
    _, _, blob_c = prove_with_blinding(cpu, n, cols, pk, public, blinders, logging=False)
    assert blob == blob_c
    dev = PR.GpuDeviceBackend(setup)
    vk = {k: decode(dev.commit(dev.i_ntt_poly(dev.Polynomial(PR.SV(pk[k]), dev.LAG)))) for k in pk}
    assert {k: decode(v) for k, v in circuit.commitments(setup).items()} == vk
    assert g1_only_verify(n, tau, *_split(blob), _challenges(blob), vk, [])
    # HBM-resident witness: same bytes
    import torch
    t = [torch.from_numpy(PR.SV(c).view(np.int64)).cuda() for c in cols]
    torch.cuda.synchronize()
    assert prover.prove_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), None, blinders) == blob
    # no public inputs (None: PI's transforms are skipped) against an explicit all-zero PI column (the general path): same bytes
    assert prover.prove_with_blinding(PR.SV(cols[0]), PR.SV(cols[1]), PR.SV(cols[2]), PR.SV([0] * n), blinders) == blob
    # host witness staged column by column inside round 1 (the default from 2^18 gates) or copied in front of it, Montgomery limbs and
    # canonical bytes, with and without a PI column: same bytes
    le = [np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in c), dtype=np.uint8).copy() for c in cols]
    for staged_env in ("1", "0"):
        monkeypatch.setenv("BP_PROVE_STAGED", staged_env)
        assert prover.prove_with_blinding(PR.SV(cols[0]), PR.SV(cols[1]), PR.SV(cols[2]), None, blinders) == blob, staged_env
        assert prover.prove_with_blinding(PR.SV(cols[0]), PR.SV(cols[1]), PR.SV(cols[2]), PR.SV([0] * n), blinders) == blob, staged_env
        out = np.zeros(624, dtype=np.uint8)
        bl = np.frombuffer(b"".join((int(v) % Q).to_bytes(32, "little") for v in blinders), dtype=np.uint8).copy()
        c_ = prover.ctx
        c_.check(c_._lib.bp_prove(c_._h, setup.handle, circuit.handle, le[0].ctypes.data, le[1].ctypes.data, le[2].ctypes.data, None, bp.FR_BYTES_LE, 0,
                                  bl.ctypes.data, out.ctypes.data), "bp_prove")
        assert bytes(out) == blob, staged_env
    monkeypatch.delenv("BP_PROVE_STAGED")
    # round 3's challenge-free coset transforms on the side stream (the default below 2^20 gates) or in round 3 itself (from 2^20)
    for side in ("0", "1"):
        monkeypatch.setenv("BP_PROVE_SIDE", side)
        assert prover.prove_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), None, blinders) == blob, side
        assert prover.prove_with_blinding(PR.SV(cols[0]), PR.SV(cols[1]), PR.SV(cols[2]), PR.SV([0] * n), blinders) == blob, side
    circuit.free()


def _circuit_from_rows(wires, selectors, n):
    """what Program / Assembly produce for a list of gate rows: wire-name rows -> value columns are the
    caller's; sigma columns join equal names (empty cells included) into one cycle each, program.rs:92-99"""
    from tests.circuit_frontend import make_s_polynomials
    pk = {k: [x % Q for x in v] + [0] * (n - len(v)) for k, v in selectors.items()}
    rows, sig = make_s_polynomials(wires, n)
    pk.update(s1=sig[0], s2=sig[1], s3=sig[2])
    return rows, pk


@pytest.mark.gpu
def test_reference_prover_unit_test_configurations():
    """src/prover.rs:682-756: `initilization()` (program ["e public"], e = 3, SRS = 14 powers of tau = 1 -- every SRS point is G)
    and `test_prove` (the toy program on 14 powers of tau = 2), with fixed blinders instead of thread_rng"""
    from oracle import oracle as O
    from tests.test_gpu_prover_rounds import prove_with_blinding, toy_circuit
    n = 8
    blinders = [random.Random(3).randrange(1, Q) for _ in range(11)]
    # initilization(): one row (e, -, -) with ql = 1 and PI = -e
    rows, pk = _circuit_from_rows([("e", None, None)], dict(ql=[1], qr=[0], qm=[0], qo=[0], qc=[0]), n)
    cols = [[3 if r[j] == "e" else 0 for r in rows] for j in range(3)]
    public = [(-3) % Q] + [0] * (n - 1)
    setup = bp.Setup.generate_srs(n + 6, 1)
    circuit = bp.Circuit({k: PR.SV(v) for k, v in pk.items()})
    blob = bp.Prover(setup, circuit).prove_with_blinding(PR.SV(cols[0]), PR.SV(cols[1]), PR.SV(cols[2]), PR.SV(public), blinders)
    cpu = PR.OracleBackend(O.proj_from_bytes96(setup.powers_of_x()))
    assert blob == prove_with_blinding(cpu, n, cols, pk, public, blinders)[2]
    circuit.free()
    # test_prove: the toy program, tau = 2
    cols, pk, public = toy_circuit(n)
    setup = bp.Setup.generate_srs(n + 6, 2)
    circuit = bp.Circuit({k: PR.SV(v) for k, v in pk.items()})
    blob = bp.Prover(setup, circuit).prove_with_blinding(PR.SV(cols[0]), PR.SV(cols[1]), PR.SV(cols[2]), PR.SV(public), blinders)
    cpu = PR.OracleBackend(O.proj_from_bytes96(setup.powers_of_x()))
    assert blob == prove_with_blinding(cpu, n, cols, pk, public, blinders)[2]
    circuit.free()


@pytest.mark.gpu
def test_full_size_2p20_gates_proof_verifies():
    """BASELINE configs[4] size: one proof of a 2^20-gate synthetic circuit (chained multiplications, as bench.py's prove leg);
    too large for the oracle's restatement, so the check is the verifier's final equation (src/verifier.rs:80-192 in G1 with
    the known tau, challenges recomputed from the proof bytes) plus tamper rejection, and determinism across witness paths"""
    import torch
    from baby_plonk_rust_amd.synthetic import chained_multiplications
    from tests.test_gpu_prover_rounds import decode, g1_only_verify
    n, tau = 1 << 20, 0x1234567 + 20
    cols, pk = chained_multiplications(n, 2020)
    setup = bp.Setup.generate_srs(n + 6, tau)
    circuit = bp.Circuit(pk)
    prover = bp.Prover(setup, circuit)
    blinders = [random.Random(20).randrange(1, Q) for _ in range(11)]
    blob = prover.prove_with_blinding(cols[0], cols[1], cols[2], None, blinders)
    t = [torch.from_numpy(c.view(np.int64)).cuda() for c in cols]
    torch.cuda.synchronize()
    assert prover.prove_device(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), None, blinders) == blob
    vk = {k: decode(v) for k, v in circuit.commitments(setup).items()}
    pts, ev = _split(blob)
    ch = _challenges(blob)
    assert g1_only_verify(n, tau, pts, ev, ch, vk, [])
    assert not g1_only_verify(n, tau, pts, dict(ev, c_bar=(ev["c_bar"] + 1) % Q), ch, vk, [])
    # a witness with one wrong product is refused
    bad = cols[2].copy()
    bad[12345] = bp.scalar_from_int(7)
    with pytest.raises(bp.BpError) as e:
        prover.prove_with_blinding(cols[0], cols[1], bad, None, blinders)
    assert e.value.code == -11
    circuit.free()
    setup.ctx.srs_free(setup.handle)


def _large_proof_records():
    import json
    return json.load(open(os.path.join(HERE, "golden", "large_vectors.json")))["proofs"]


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [r["log_n"] for r in _large_proof_records()])
def test_large_proofs_match_the_committed_oracle_hashes(log_n):
    """byte parity of bp_prove at 2^12 ... 2^19 gates, where its size-dependent branches change (two-pass
    NTTs on the 4n coset from 2^12, the three-pass NTT at 4n = 2^21, chunked carry scans of the division by x^n - 1, full-width
    fixed-base tables).  The
    expected sha256 values come from the CPU oracle restatement of src/prover.rs (tests/golden/make_large_vectors.py, made once in the
    build container); inputs are regenerated here from the recorded recipe.  Every commitment is compared too, so a mismatch names
    the round it comes from.  Also through a two-shard context (bp_init_multi) and without tables: same bytes."""
    from tests.test_gpu_prover_rounds import synthetic_circuit
    rec = [r for r in _large_proof_records() if r["log_n"] == log_n][0]
    n = 1 << rec["log_n"]
    cols, pk, public = synthetic_circuit(n, rec["seed"])
    blinders = [random.Random(100 + rec["log_n"]).randrange(1, Q) for _ in range(11)]
    wit = [PR.SV(c) for c in cols]
    pkv = {k: PR.SV(v) for k, v in pk.items()}
    names = ("a_1", "b_1", "c_1", "z_1", "t_lo_1", "t_mid_1", "t_hi_1", "w_zeta_1", "w_zeta_omega_1")
    # group contexts split round 3 by COSET over their members (two cosets each on 2-3 members, one each on 4 and more): the
    # quotient comes back as four residues and is recombined -- the t commitments and the proof hash below cover it
    for devs, tables in ((0, True), (0, False), ([0, 0], True), ([0, 0, 0, 0], True)) if log_n <= 16 else ((0, True), ([0, 0, 0], True), ([0] * 5, True)):
        ctx = bp.Context(devs)
        setup = bp.Setup.generate_srs(n + 6, rec["tau"], ctx, tables=tables)
        assert ctx.srs_len(setup.handle) == rec["srs_powers"]
        blob = bp.Prover(setup, bp.Circuit(pkv, ctx)).prove_with_blinding(wit[0], wit[1], wit[2], None, blinders)
        for i, k in enumerate(names):                     # commitments are recorded as hashes of the 96-byte encodings
            assert hashlib.sha256(M.enc96(M.dec48(blob[48 * i: 48 * i + 48]))).hexdigest() == rec["commitment_sha256"][k], (k, devs, tables)
        for i, k in enumerate(("a_bar", "b_bar", "c_bar", "s1_bar", "s2_bar", "z_omega_bar")):
            assert "%064x" % int.from_bytes(blob[432 + 32 * i: 464 + 32 * i], "little") == rec["evaluations"][k], k
        assert hashlib.sha256(blob).hexdigest() == rec["proof_sha256"]
        ctx.close()
