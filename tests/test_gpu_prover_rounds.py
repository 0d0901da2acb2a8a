"""-m gpu: the reference prover's five rounds (src/prover.rs:177-647), restated call by call in tests/prover_rounds.py,
executed on the GPU library and on the CPU oracle for the toy circuit of tests/verify_proof_test.rs (group order 8,
SRS = 14 powers of tau = 101, witness a=3 b=4 c=16 d=5 e=80) with fixed blinders and challenges.  Every intermediate
polynomial, evaluation and commitment must agree byte for byte; the resulting proof must satisfy the verifier's final
equation (src/verifier.rs:80-192), checked in G1 with the known tau."""
import random

import numpy as np
import pytest

import baby_plonk_rust_amd as bp
from oracle import oracle as O
from tests import bigint_model as M
from tests import prover_rounds as PR

pytestmark = pytest.mark.gpu
Q = M.Q


def toy_circuit(n=8):
    """tests/verify_proof_test.rs:21-35 + SURVEY.md appendix A: rows (e,-,-) / (a,b,c) / (c,d,e), five empty rows"""
    w = {"a": 3, "b": 4, "c": 16, "d": 5, "e": 80}
    wires = [("e", None, None), ("a", "b", "c"), ("c", "d", "e")] + [(None, None, None)] * (n - 3)
    sel = dict(ql=[1, 0, 0], qr=[0, -1, 0], qm=[0, -1, -1], qo=[0, 1, 1], qc=[0, 0, 0])
    pk = {k: [x % Q for x in v] + [0] * (n - 3) for k, v in sel.items()}
    cols = [[w.get(row[j], 0) if row[j] else 0 for row in wires] for j in range(3)]
    from tests.circuit_frontend import make_s_polynomials
    _, sig = make_s_polynomials(wires[:3], n)                             # program.rs:76-147
    pk.update(s1=sig[0], s2=sig[1], s3=sig[2])
    public = [(-80) % Q] + [0] * (n - 1)                                 # prover.rs:114-127
    return cols, pk, public


def decode(b96):
    return None if b96[0] & 0x40 else (int.from_bytes(b96[:48], "big"), int.from_bytes(b96[48:], "big"))


def g1_only_verify(n, tau, proof, ev, ch, vk, public_inputs):
    """src/verifier.rs:80-192 with the pairing check e(A, [tau]_2) == e(B, [1]_2) replaced by tau * A == B"""
    beta, gamma, alpha, zeta, nu, mu = (ch[k] for k in ("beta", "gamma", "alpha", "zeta", "nu", "mu"))
    mul = lambda k, P: M.ec_mul(k % Q, P) if P is not None else None
    add = M.ec_add
    neg = lambda P: None if P is None else (P[0], (-P[1]) % M.P)
    z_h_zeta = (pow(zeta, n, Q) - 1) % Q
    omega = M.omega(n)
    # L_i(zeta) = omega^i (zeta^n - 1) / (n (zeta - omega^i)): the value verifier.rs:91-104 obtains by i_ntt + coeffs_evaluate
    lag = lambda i: pow(omega, i, Q) * z_h_zeta % Q * pow(n * (zeta - pow(omega, i, Q)) % Q, Q - 2, Q) % Q
    l_1_zeta = lag(0)
    pi_eval = sum((-x) * lag(i) for i, x in enumerate(public_inputs)) % Q
    a_bar, b_bar, c_bar, s1_bar, s2_bar, zw_bar = (ev[k] for k in ("a_bar", "b_bar", "c_bar", "s1_bar", "s2_bar", "z_omega_bar"))
    rl = lambda s, o: (s + o * beta + gamma) % Q
    r_0 = (pi_eval - l_1_zeta * alpha * alpha - alpha * rl(a_bar, s1_bar) * rl(b_bar, s2_bar) * (c_bar + gamma) * zw_bar) % Q
    d1 = add(add(add(add(mul(a_bar * b_bar, vk["qm"]), mul(a_bar, vk["ql"])), mul(b_bar, vk["qr"])), mul(c_bar, vk["qo"])), vk["qc"])
    d2 = mul(rl(a_bar, zeta) * rl(b_bar, PR.K1 * zeta) * rl(c_bar, PR.K2 * zeta) * alpha + l_1_zeta * alpha * alpha + mu, proof["z_1"])
    d3 = mul(rl(a_bar, s1_bar) * rl(b_bar, s2_bar) * alpha * beta * zw_bar, vk["s3"])
    d4 = mul(z_h_zeta, add(add(proof["t_lo_1"], mul(pow(zeta, n, Q), proof["t_mid_1"])), mul(pow(zeta, 2 * n, Q), proof["t_hi_1"])))
    d = add(add(add(d1, d2), neg(d3)), neg(d4))
    f = d
    for k, P in enumerate((proof["a_1"], proof["b_1"], proof["c_1"], vk["s1"], vk["s2"]), start=1):
        f = add(f, mul(pow(nu, k, Q), P))
    e_scalar = (nu * a_bar + nu**2 * b_bar + nu**3 * c_bar + nu**4 * s1_bar + nu**5 * s2_bar + mu * zw_bar - r_0) % Q
    e = M.ec_mul(e_scalar)
    lhs = mul(tau, add(proof["w_zeta_1"], mul(mu, proof["w_zeta_omega_1"])))
    rhs = add(add(add(mul(zeta, proof["w_zeta_1"]), mul(mu * zeta * omega, proof["w_zeta_omega_1"])), f), neg(e))
    return lhs == rhs


def run_rounds(B, n, cols, pk, public, blinders, ch, z_fn=None, logging=True):
    st = PR.ProverState(B, n, pk, blinders, logging)
    proof = {}
    proof["a_1"], proof["b_1"], proof["c_1"] = PR.round_1(st, cols[0], cols[1], cols[2], public)
    st.rand.update(beta=ch["beta"], gamma=ch["gamma"])
    proof["z_1"] = PR.round_2(st)
    st.rand["alpha"] = ch["alpha"]
    proof["t_lo_1"], proof["t_mid_1"], proof["t_hi_1"] = PR.round_3(st)
    st.rand["zeta"] = ch["zeta"]
    ev = PR.round_4(st)
    st.rand["nu"] = ch["nu"]
    proof["w_zeta_1"], proof["w_zeta_omega_1"] = PR.round_5(st)
    return st, proof, ev


def test_toy_circuit_rounds_gpu_vs_oracle_and_verify():
    n, tau = 8, 101
    cols, pk, public = toy_circuit(n)
    rnd = random.Random(2024)
    blinders = [rnd.randrange(1, Q) for _ in range(11)]
    ch = {k: rnd.randrange(1, Q) for k in ("beta", "gamma", "alpha", "zeta", "nu", "mu")}

    setup = bp.Setup.generate_srs(n + 6, tau)                           # verify_proof_test.rs:16
    gpu = PR.GpuBackend(setup)
    cpu = PR.OracleBackend(O.proj_from_bytes96(setup.powers_of_x()))

    st_g, proof_g, ev_g = run_rounds(gpu, n, cols, pk, public, blinders, ch, lambda *a: bp.round_2_z(*a))
    st_c, proof_c, ev_c = run_rounds(cpu, n, cols, pk, public, blinders, ch, lambda *a: O.round2_z(*a))

    # every logged intermediate agrees bit for bit
    for rk in ("round_1", "round_2", "round_3", "round_4", "round_5"):
        for name, val in st_c.log[rk].items():
            got = st_g.log[rk][name]
            if isinstance(val, np.ndarray):
                assert got.shape == val.shape and (got == val).all(), (rk, name)
            else:
                assert got == val, (rk, name)
    assert proof_g == proof_c and ev_g == ev_c
    # polynomial lengths of SURVEY.md appendix A
    assert len(st_g.log["round_1"]["a_coeff"]) == n + 2 and len(st_g.log["round_2"]["z_coeff"]) == n + 3
    assert len(st_g.log["round_3"]["t"]) == 3 * n + 6 and len(st_g.log["round_3"]["t_hi"]) == n + 6
    assert len(st_g.log["round_5"]["w_zeta_omega"]) == n + 2

    # 624-byte proof: 9 compressed points (verifier.rs:23-40 field order) then 6 little-endian scalars
    order = ("a_1", "b_1", "c_1", "z_1", "t_lo_1", "t_mid_1", "t_hi_1", "w_zeta_1", "w_zeta_omega_1")
    blob = b"".join(M.enc48(decode(proof_g[k])) for k in order) + b"".join(
        ev_g[k].to_bytes(32, "little") for k in ("a_bar", "b_bar", "c_bar", "s1_bar", "s2_bar", "z_omega_bar"))
    assert len(blob) == 624

    # the verifier's final equation holds for this proof (G1-only form for known tau)
    vk = {k: decode(gpu.commit(gpu.Polynomial(gpu.i_ntt_381(PR.SV(pk[k])), gpu.MONO))) for k in pk}    # verifier.rs:61-68
    pts = {k: decode(v) for k, v in proof_g.items()}
    assert g1_only_verify(n, tau, pts, ev_g, ch, vk, [80])
    # and fails when an evaluation is tampered with
    bad = dict(ev_g, a_bar=(ev_g["a_bar"] + 1) % Q)
    assert not g1_only_verify(n, tau, pts, bad, ch, vk, [80])


def prove_with_blinding(B, n, cols, pk, public, blinders, z_fn=None, logging=True):
    """src/prover.rs:106-176 with the blinders as an argument (the reference draws them from thread_rng, :108-110) and the
    Fiat-Shamir challenges from the Merlin transcript exactly as src/transcript.rs derives them"""
    from tests.merlin_transcript import PlonkTranscript
    comp = lambda b96: M.enc48(decode(b96))
    tr = PlonkTranscript()
    st = PR.ProverState(B, n, pk, blinders, logging)
    proof = {}
    proof["a_1"], proof["b_1"], proof["c_1"] = PR.round_1(st, cols[0], cols[1], cols[2], public)
    st.rand["beta"], st.rand["gamma"] = tr.round_1(comp(proof["a_1"]), comp(proof["b_1"]), comp(proof["c_1"]))
    proof["z_1"] = PR.round_2(st)
    st.rand["alpha"] = tr.round_2(comp(proof["z_1"]))
    proof["t_lo_1"], proof["t_mid_1"], proof["t_hi_1"] = PR.round_3(st)
    st.rand["zeta"] = tr.round_3(comp(proof["t_lo_1"]), comp(proof["t_mid_1"]), comp(proof["t_hi_1"]))
    ev = PR.round_4(st)
    st.rand["nu"] = tr.round_4(ev["a_bar"], ev["b_bar"], ev["c_bar"], ev["s1_bar"], ev["s2_bar"], ev["z_omega_bar"])
    proof["w_zeta_1"], proof["w_zeta_omega_1"] = PR.round_5(st)
    order = ("a_1", "b_1", "c_1", "z_1", "t_lo_1", "t_mid_1", "t_hi_1", "w_zeta_1", "w_zeta_omega_1")
    blob = b"".join(comp(proof[k]) for k in order) + b"".join(
        ev[k].to_bytes(32, "little") for k in ("a_bar", "b_bar", "c_bar", "s1_bar", "s2_bar", "z_omega_bar"))
    return proof, ev, blob


def compute_challenges(proof, ev):
    """src/verifier.rs:193-209"""
    from tests.merlin_transcript import PlonkTranscript
    comp = lambda b96: M.enc48(decode(b96))
    tr = PlonkTranscript()
    beta, gamma = tr.round_1(comp(proof["a_1"]), comp(proof["b_1"]), comp(proof["c_1"]))
    alpha = tr.round_2(comp(proof["z_1"]))
    zeta = tr.round_3(comp(proof["t_lo_1"]), comp(proof["t_mid_1"]), comp(proof["t_hi_1"]))
    nu = tr.round_4(ev["a_bar"], ev["b_bar"], ev["c_bar"], ev["s1_bar"], ev["s2_bar"], ev["z_omega_bar"])
    mu = tr.round_5(comp(proof["w_zeta_1"]), comp(proof["w_zeta_omega_1"]))
    return dict(beta=beta, gamma=gamma, alpha=alpha, zeta=zeta, nu=nu, mu=mu)


def test_toy_circuit_deterministic_proof_bytes_and_verify():
    """prove -> 624 proof bytes on the GPU path == on the oracle path; the verifier (challenges recomputed from the proof)
    accepts -- the reference's tests/verify_proof_test.rs, made reproducible"""
    n, tau = 8, 101
    cols, pk, public = toy_circuit(n)
    blinders = [random.Random(99).randrange(1, Q) for _ in range(11)]
    setup = bp.Setup.generate_srs(n + 6, tau)
    gpu = PR.GpuBackend(setup)
    cpu = PR.OracleBackend(O.proj_from_bytes96(setup.powers_of_x()))
    proof_g, ev_g, blob_g = prove_with_blinding(gpu, n, cols, pk, public, blinders, lambda *a: bp.round_2_z(*a))
    proof_c, ev_c, blob_c = prove_with_blinding(cpu, n, cols, pk, public, blinders, lambda *a: O.round2_z(*a))
    assert blob_g == blob_c and len(blob_g) == 624
    import hashlib, os
    golden = open(os.path.join(os.path.dirname(__file__), "golden", "toy_proof_blinders_seed99.sha256")).read().strip()
    assert hashlib.sha256(blob_g).hexdigest() == golden          # committed fixture (made by the oracle path on the CPU)
    vk = {k: decode(gpu.commit(gpu.Polynomial(gpu.i_ntt_381(PR.SV(pk[k])), gpu.MONO))) for k in pk}
    ch = compute_challenges(proof_g, ev_g)
    assert g1_only_verify(n, tau, {k: decode(v) for k, v in proof_g.items()}, ev_g, ch, vk, [80])
    # a different public input is rejected
    assert not g1_only_verify(n, tau, {k: decode(v) for k, v in proof_g.items()}, ev_g, ch, vk, [81])


def synthetic_circuit(n, seed):
    """n multiplication gates z_i = x_i * y_i chained by copy constraints x_{i+1} = z_i (qm = -1, qo = 1, as the
    reference's parser emits for `c <== a * b`, assembly.rs:30-81); no public inputs"""
    rnd = random.Random(seed)
    x = rnd.randrange(Q)
    cols = [[0] * n for _ in range(3)]
    for i in range(n):
        y = rnd.randrange(Q)
        cols[0][i], cols[1][i], cols[2][i] = x, y, x * y % Q
        x = cols[2][i]
    zero = [0] * n
    pk = dict(ql=zero, qr=zero, qm=[Q - 1] * n, qo=[1] * n, qc=zero)
    om = M.omega(n)
    pw = [1] * n
    for i in range(1, n):
        pw[i] = pw[i - 1] * om % Q
    lab = lambda col, row: (col + 1) * pw[row] % Q
    s1, s2, s3 = [lab(0, i) for i in range(n)], [lab(1, i) for i in range(n)], [lab(2, i) for i in range(n)]
    for i in range(n - 1):                      # cycle {(C, i), (A, i+1)}
        s3[i], s1[i + 1] = lab(0, i + 1), lab(2, i)
    pk.update(s1=s1, s2=s2, s3=s3)
    return cols, pk, [0] * n


def test_device_resident_pipeline_toy_and_synthetic():
    """the same five rounds with every polynomial resident in HBM (DevicePolynomial)"""
    import hashlib, os
    # toy circuit: identical proof bytes (committed fixture)
    n, tau = 8, 101
    cols, pk, public = toy_circuit(n)
    blinders = [random.Random(99).randrange(1, Q) for _ in range(11)]
    setup = bp.Setup.generate_srs(n + 6, tau)
    dev = PR.GpuDeviceBackend(setup)
    proof, ev, blob = prove_with_blinding(dev, n, cols, pk, public, blinders)
    golden = open(os.path.join(os.path.dirname(__file__), "golden", "toy_proof_blinders_seed99.sha256")).read().strip()
    assert hashlib.sha256(blob).hexdigest() == golden
    # synthetic 2^10-gate circuit: device-resident == host-pointer GPU path == oracle path, and the proof verifies
    n, tau = 1 << 10, 0x1234567
    cols, pk, public = synthetic_circuit(n, 5)
    setup = bp.Setup.generate_srs(n + 6, tau)
    dev, gpu = PR.GpuDeviceBackend(setup), PR.GpuBackend(setup)
    cpu = PR.OracleBackend(O.proj_from_bytes96(setup.powers_of_x()))
    cpu.threads = 16
    proof_d, ev_d, blob_d = prove_with_blinding(dev, n, cols, pk, public, blinders, logging=False)
    proof_g, ev_g, blob_g = prove_with_blinding(gpu, n, cols, pk, public, blinders, logging=False)
    proof_c, ev_c, blob_c = prove_with_blinding(cpu, n, cols, pk, public, blinders, logging=False)
    assert blob_d == blob_g == blob_c
    vk = {k: decode(dev.commit(dev.i_ntt_poly(dev.Polynomial(PR.SV(pk[k]), dev.LAG)))) for k in pk}
    ch = compute_challenges(proof_d, ev_d)
    assert g1_only_verify(n, tau, {k: decode(v) for k, v in proof_d.items()}, ev_d, ch, vk, [])
