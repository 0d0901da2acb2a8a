"""The reference prover's rounds 1-5 (src/prover.rs:177-647) restated line by line OVER AN ABSTRACT BACKEND, so that the
same call sequence -- every `Polynomial` operator, `i_ntt_381`, `coeffs_evaluate`, `Setup::commit` -- runs once on the
GPU library (through baby_plonk_rust_amd's mirror of the reference interface) and once on the CPU oracle, and every
intermediate polynomial, evaluation and commitment can be compared byte for byte.

Test infrastructure: the prover itself (transcript, circuit front-end) is out of the product's scope.  Challenges are
inputs here (the reference's round_k functions are `pub` and read them from `random_nums`), blinders are inputs too
(`prove` draws them from thread_rng, prover.rs:108-110, so the reference's proofs are not reproducible).

A backend provides:  Polynomial(values[n,4], basis) with + - * / (Polynomial or scalar limbs), coeffs_evaluate, i_ntt;
i_ntt_381(values); roots_of_unity(n); commit(poly) -> 96 bytes;  basis constants MONO / LAG.
Scalars travel as Python ints mod q at this level and are converted with the helpers below.
"""
import numpy as np

from oracle import oracle as O

Q = O.Q
K1, K2 = 2, 3                                   # prover.rs:99-100


def S(v):
    """int -> Montgomery limbs"""
    return O.fr_from_int(v % Q)


def SV(vals):
    """list of ints -> Montgomery limbs [n, 4]; zeros (the bulk of x^n - 1, L_1, x^n ...) cost nothing"""
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        if v:
            out[i] = O.fr_from_int(v % Q)
    return out


def sparse(length, entries):
    """limbs of a length-`length` vector that is zero except for {index: int}"""
    out = np.zeros((length, 4), dtype=np.uint64)
    for i, v in entries.items():
        out[i] = O.fr_from_int(v % Q)
    return out


def ints(values):
    return O.fr_array_to_ints(np.ascontiguousarray(values, dtype=np.uint64).reshape(-1, 4))


def root_of_unity(n):
    return pow(pow(7, (Q - 1) >> 32, Q), (1 << 32) // n, Q)      # utils.rs:39-43


class OracleBackend:
    """the CPU oracle behind the reference-shaped interface"""
    MONO, LAG = 1, 0

    class Polynomial:
        def __init__(self, values, basis):
            self.values = np.ascontiguousarray(values, dtype=np.uint64).reshape(-1, 4).copy()
            self.basis = basis

        def __len__(self):
            return len(self.values)

        def _wrap(self, vals, basis=None):
            return OracleBackend.Polynomial(vals, self.basis if basis is None else basis)

        def _scalar(self, name, s):
            out = O.u64(self.values.shape)
            s = np.ascontiguousarray(s, dtype=np.uint64)
            args = [out.ctypes.data, self.values.ctypes.data, len(self.values), s.ctypes.data]
            getattr(O.lib, name)(*(args + ([self.basis] if name != "poly_mul_scalar" else [])))
            return self._wrap(out)

        def __add__(self, o):
            if isinstance(o, OracleBackend.Polynomial):
                assert self.basis == o.basis
                return self._wrap(O.poly_binop("poly_add", self.values, o.values, self.basis))
            return self._scalar("poly_add_scalar", o)

        def __sub__(self, o):
            if isinstance(o, OracleBackend.Polynomial):
                assert self.basis == o.basis
                return self._wrap(O.poly_binop("poly_sub", self.values, o.values, self.basis))
            return self._scalar("poly_sub_scalar", o)

        def __mul__(self, o):
            if isinstance(o, OracleBackend.Polynomial):
                assert self.basis == o.basis == 1
                return self._wrap(O.poly_binop("poly_mul_fast", self.values, o.values))
            return self._scalar("poly_mul_scalar", o)

        def __truediv__(self, o):
            assert self.basis == o.basis == 1
            return self._wrap(O.poly_binop("poly_div", self.values, o.values))

        def coeffs_evaluate(self, x):
            assert self.basis == 1
            return O.poly_eval(self.values, x, fast=True)

        def i_ntt(self):
            assert self.basis == 0
            return self._wrap(O.ntt_fast(self.values, inverse=True), 1)

    def __init__(self, srs_proj):
        self.srs = srs_proj                                     # [m, 18] projective points

    def i_ntt_381(self, values):
        return O.ntt_fast(values, inverse=True)

    def roots_of_unity(self, n):
        out = O.u64((n, 4))
        O.lib.ntt_roots_of_unity(out.ctypes.data, n)
        return out

    def commit(self, poly):                                      # setup.rs:32-37
        assert poly.basis == 1
        return O.g1_bytes96(O.bucket_msm(self.srs, poly.values, 256, 4, threads=self.threads))

    threads = 1

    def i_ntt_poly(self, values_poly):
        return self.Polynomial(O.ntt_fast(values_poly.values, inverse=True), 1)

    def roots_poly_lagrange(self, n):
        return self.Polynomial(self.roots_of_unity(n), 0)

    def round2_z(self, a, b, c, s1, s2, s3, beta, gamma):
        return self.Polynomial(O.round2_z(a.values, b.values, c.values, s1.values, s2.values, s3.values, beta, gamma), 0)

    def scale_powers(self, z, omega):
        return generic_scale_powers(self, z, omega)

    def slice(self, p, lo, hi=None):
        return self.Polynomial(p.values[lo:hi], p.basis)


class GpuBackend:
    """the HIP library behind the same interface (baby_plonk_rust_amd.api)"""
    MONO, LAG = 1, 0

    def __init__(self, setup):
        import baby_plonk_rust_amd as bp
        self.bp, self.setup = bp, setup
        self.Polynomial = bp.Polynomial

    def i_ntt_381(self, values):
        return self.bp.i_ntt_381(values)

    def roots_of_unity(self, n):
        return self.bp.roots_of_unity(n)

    def commit(self, poly):
        return self.setup.commit(poly)

    def i_ntt_poly(self, values_poly):
        return self.Polynomial(self.bp.i_ntt_381(values_poly.values), 1)

    def roots_poly_lagrange(self, n):
        return self.Polynomial(self.bp.roots_of_unity(n), 0)

    def round2_z(self, a, b, c, s1, s2, s3, beta, gamma):
        return self.Polynomial(self.bp.round_2_z(a.values, b.values, c.values, s1.values, s2.values, s3.values, beta, gamma), 0)

    def scale_powers(self, z, omega):
        return generic_scale_powers(self, z, omega)

    def slice(self, p, lo, hi=None):
        return self.Polynomial(p.values[lo:hi], p.basis)


class GpuDeviceBackend:
    """the HIP library with every polynomial resident in HBM (baby_plonk_rust_amd.DevicePolynomial): the quotient /
    linearisation pipeline of SURVEY.md section 8f row 1 -- only O(1) scalars and the 96-byte commitments cross PCIe"""
    MONO, LAG = 1, 0

    def __init__(self, setup):
        import baby_plonk_rust_amd as bp
        self.bp, self.setup = bp, setup
        self.Polynomial = lambda values, basis: bp.DevicePolynomial(values, basis, setup.ctx)

    def commit(self, poly):
        return self.bp.commit_device(self.setup, poly)

    def i_ntt_poly(self, values_poly):
        return values_poly.i_ntt()

    def roots_poly_lagrange(self, n):
        return self.bp.roots_of_unity_device(n, self.setup.ctx)

    def round2_z(self, a, b, c, s1, s2, s3, beta, gamma):
        return self.bp.round_2_z_device(a, b, c, s1, s2, s3, beta, gamma)

    def scale_powers(self, z, omega):
        return z.scale_powers(S(omega))

    def slice(self, p, lo, hi=None):
        return p.slice(lo, hi)


# ------------------------------------------------------------------------------------------------------------------
def generic_scale_powers(B, z, omega):
    """prover.rs:661-674 monomial_z_to_z_omega: coefficient i times omega^i"""
    vals, p = [], 1
    for v in ints(z.values):
        vals.append(v * p % Q)
        p = p * omega % Q
    return B.Polynomial(SV(vals), B.MONO)


class ProverState:
    """prover.rs:50-62 minus the circuit front-end: the preprocessed columns are given as Lagrange value lists (ints)"""

    def __init__(self, B, n, pk, blinders, logging=True):
        self.B, self.n, self.b = B, n, blinders                 # blinders: 11 ints (prover.rs:110)
        P = B.Polynomial
        as_limbs = lambda v: v if isinstance(v, np.ndarray) else SV(v)
        self.pk = {k: P(as_limbs(v), B.LAG) for k, v in pk.items()}   # ql qr qm qo qc s1 s2 s3 (program.rs:34-50)
        self.rand, self.wp, self.evals, self.log, self.logging = {}, {}, {}, {}, logging

    def record(self, rk, **kw):
        """keep intermediates for byte-for-byte comparison (polynomials are downloaded; off for large-n timing runs)"""
        if self.logging:
            self.log[rk] = {k: (v.values if hasattr(v, "values") else v) for k, v in kw.items()}

    def z_h(self):                                               # x^n - 1 (prover.rs:229-235)
        return self.B.Polynomial(sparse(self.n + 1, {0: -1, self.n: 1}), self.B.MONO)


def rlc(p, other, beta, gamma):
    """impl Rlc for Polynomial (utils.rs:170-175): self + other * beta + gamma"""
    return p + other * S(beta) + S(gamma)


def round_1(st, a_values, b_values, c_values, public_values):
    """prover.rs:177-277 (witness lookup replaced by the three value columns); public_values: prover.rs:114-127"""
    B, P = st.B, st.B.Polynomial
    as_limbs = lambda v: v if isinstance(v, np.ndarray) else SV(v)
    st.public_input_poly = P(as_limbs(public_values), B.LAG)
    z_h = st.z_h()
    b1, b2, b3, b4, b5, b6 = st.b[0:6]
    a, b, c = P(as_limbs(a_values), B.LAG), P(as_limbs(b_values), B.LAG), P(as_limbs(c_values), B.LAG)
    a_coeff = P(SV([b2, b1]), B.MONO) * z_h + a.i_ntt()
    b_coeff = P(SV([b4, b3]), B.MONO) * z_h + b.i_ntt()
    c_coeff = P(SV([b6, b5]), B.MONO) * z_h + c.i_ntt()
    st.wp.update(a=a, b=b, c=c, a_coeff=a_coeff, b_coeff=b_coeff, c_coeff=c_coeff, z_h_coeff=z_h)
    out = (B.commit(a_coeff), B.commit(b_coeff), B.commit(c_coeff))
    st.record("round_1", a_coeff=a_coeff, b_coeff=b_coeff, c_coeff=c_coeff, commits=out)
    return out


def round_2(st, z_values_fn=None):
    """prover.rs:279-368 (the grand product loop :286-319 is the backend's round2_z)"""
    B, P = st.B, st.B.Polynomial
    beta, gamma = st.rand["beta"], st.rand["gamma"]
    z = B.round2_z(st.wp["a"], st.wp["b"], st.wp["c"], st.pk["s1"], st.pk["s2"], st.pk["s3"], S(beta), S(gamma))
    b7, b8, b9 = st.b[6:9]
    z_coeff = P(SV([b9, b8, b7]), B.MONO) * st.wp["z_h_coeff"] + z.i_ntt()
    st.wp.update(z=z, z_coeff=z_coeff)
    z_1 = B.commit(z_coeff)
    st.record("round_2", z=z, z_coeff=z_coeff, commit=z_1)
    return z_1


def round_3(st):
    """prover.rs:370-500"""
    B, P, n = st.B, st.B.Polynomial, st.n
    coeff = {k: B.i_ntt_poly(st.pk[k]) for k in ("s1", "s2", "s3", "ql", "qr", "qm", "qo", "qc")}
    a, b, c, z = st.wp["a_coeff"], st.wp["b_coeff"], st.wp["c_coeff"], st.wp["z_coeff"]
    l1 = P(sparse(n, {0: 1}), B.LAG)
    z_h = st.z_h()
    gate = (a * coeff["ql"] + b * coeff["qr"] + a * b * coeff["qm"] + c * coeff["qo"]
            + st.public_input_poly.i_ntt() + coeff["qc"])
    roots_poly = B.i_ntt_poly(B.roots_poly_lagrange(n))
    omega = root_of_unity(n)
    z_omega = B.scale_powers(z, omega)                                     # prover.rs:661-674
    beta, gamma, alpha = st.rand["beta"], st.rand["gamma"], st.rand["alpha"]
    perm = ((rlc(a, roots_poly, beta, gamma) * rlc(b, roots_poly * S(K1), beta, gamma) * rlc(c, roots_poly * S(K2), beta, gamma)) * z
            - (rlc(a, coeff["s1"], beta, gamma) * rlc(b, coeff["s2"], beta, gamma) * rlc(c, coeff["s3"], beta, gamma)) * z_omega)
    l1_coeff = B.i_ntt_poly(l1)
    first_row = (z - S(1)) * l1_coeff
    all_constraints = gate + perm * S(alpha) + first_row * S(alpha * alpha % Q)
    t = all_constraints / z_h
    t_lo, t_mid, t_hi = B.slice(t, 0, n), B.slice(t, n, 2 * n), B.slice(t, 2 * n)                # prover.rs:649-659
    b10, b11 = st.b[9], st.b[10]
    x_pow_n = P(sparse(n + 1, {n: 1}), B.MONO)
    t_lo = t_lo + x_pow_n * S(b10)
    t_mid = t_mid + (x_pow_n * S(b11) - S(b10))
    t_hi = t_hi + S(-b11)
    st.pk_coeff = coeff
    st.wp.update(z_omega_coeff=z_omega, t_lo_coeff=t_lo, t_mid_coeff=t_mid, t_hi_coeff=t_hi)
    out = (B.commit(t_lo), B.commit(t_mid), B.commit(t_hi))
    st.record("round_3", t=t, t_lo=t_lo, t_mid=t_mid, t_hi=t_hi, commits=out)
    return out


def round_4(st):
    """prover.rs:502-541"""
    zeta = S(st.rand["zeta"])
    ev = dict(a_bar=st.wp["a_coeff"].coeffs_evaluate(zeta), b_bar=st.wp["b_coeff"].coeffs_evaluate(zeta),
              c_bar=st.wp["c_coeff"].coeffs_evaluate(zeta), s1_bar=st.pk_coeff["s1"].coeffs_evaluate(zeta),
              s2_bar=st.pk_coeff["s2"].coeffs_evaluate(zeta), z_omega_bar=st.wp["z_omega_coeff"].coeffs_evaluate(zeta))
    st.evals = {k: O.fr_to_int(v) for k, v in ev.items()}
    st.record("round_4", **st.evals)
    return st.evals


def round_5(st):
    """prover.rs:543-647"""
    B, P, n = st.B, st.B.Polynomial, st.n
    e, r = st.evals, st.rand
    a_bar, b_bar, c_bar, s1_bar, s2_bar, z_omega_bar = (e[k] for k in ("a_bar", "b_bar", "c_bar", "s1_bar", "s2_bar", "z_omega_bar"))
    alpha, beta, gamma, zeta, nu = r["alpha"], r["beta"], r["gamma"], r["zeta"], r["nu"]
    a, b, c, z = st.wp["a_coeff"], st.wp["b_coeff"], st.wp["c_coeff"], st.wp["z_coeff"]
    s1, s2 = st.pk_coeff["s1"], st.pk_coeff["s2"]
    pk = st.pk
    r1 = (pk["qm"].i_ntt() * S(a_bar) * S(b_bar) + pk["ql"].i_ntt() * S(a_bar) + pk["qr"].i_ntt() * S(b_bar)
          + pk["qo"].i_ntt() * S(c_bar) + st.public_input_poly.i_ntt().coeffs_evaluate(S(zeta)) + pk["qc"].i_ntt())
    r2 = (z * S(a_bar + zeta * beta + gamma) * S(b_bar + zeta * beta * K1 + gamma) * S(c_bar + zeta * beta * K2 + gamma)
          - (pk["s3"].i_ntt() * S(beta) + S(c_bar) + S(gamma)) * S(a_bar + s1_bar * beta + gamma) * S(b_bar + s2_bar * beta + gamma)
          * S(z_omega_bar))
    l1_coeff = B.i_ntt_poly(P(sparse(n, {0: 1}), B.LAG))
    r3 = (z - S(1)) * l1_coeff.coeffs_evaluate(S(zeta))
    z_h = st.z_h()
    omega = root_of_unity(n)
    assert O.fr_to_int(z_h.coeffs_evaluate(S(omega))) == 0                                   # prover.rs:602
    r4 = ((st.wp["t_lo_coeff"] + st.wp["t_mid_coeff"] * S(pow(zeta, n, Q)) + st.wp["t_hi_coeff"] * S(pow(zeta, 2 * n, Q)))
          * z_h.coeffs_evaluate(S(zeta)))
    r_coeff = r1 + r2 * S(alpha) + r3 * S(alpha) * S(alpha) - r4
    assert O.fr_to_int(r_coeff.coeffs_evaluate(S(zeta))) == 0                                # prover.rs:615
    w_zeta = ((r_coeff + (a - S(a_bar)) * S(nu) + (b - S(b_bar)) * S(nu) * S(nu) + (c - S(c_bar)) * S(pow(nu, 3, Q))
               + (s1 - S(s1_bar)) * S(pow(nu, 4, Q)) + (s2 - S(s2_bar)) * S(pow(nu, 5, Q)))
              / P(SV([-zeta, 1]), B.MONO))
    w_zeta_omega = (z - S(z_omega_bar)) / P(SV([-(zeta * omega), 1]), B.MONO)
    out = (B.commit(w_zeta), B.commit(w_zeta_omega))
    st.record("round_5", r=r_coeff, w_zeta=w_zeta, w_zeta_omega=w_zeta_omega, commits=out)
    return out
