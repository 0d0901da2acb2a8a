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
    return O.fr_array_from_ints([v % Q for v in vals])


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
        return O.g1_bytes96(O.bucket_msm(self.srs, poly.values, 256, 4))


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


# ------------------------------------------------------------------------------------------------------------------
class ProverState:
    """prover.rs:50-62 minus the circuit front-end: the preprocessed columns are given as Lagrange value lists (ints)"""

    def __init__(self, B, n, pk, blinders):
        self.B, self.n, self.b = B, n, blinders                 # blinders: 11 ints (prover.rs:110)
        P = B.Polynomial
        self.pk = {k: P(SV(v), B.LAG) for k, v in pk.items()}   # ql qr qm qo qc s1 s2 s3 (program.rs:34-50)
        self.rand, self.wp, self.evals, self.log = {}, {}, {}, {}

    def z_h(self):                                               # x^n - 1 (prover.rs:229-235)
        return self.B.Polynomial(SV([-1] + [0] * (self.n - 1) + [1]), self.B.MONO)


def rlc(p, other, beta, gamma):
    """impl Rlc for Polynomial (utils.rs:170-175): self + other * beta + gamma"""
    return p + other * S(beta) + S(gamma)


def round_1(st, a_values, b_values, c_values, public_values):
    """prover.rs:177-277 (witness lookup replaced by the three value columns); public_values: prover.rs:114-127"""
    B, P = st.B, st.B.Polynomial
    st.public_input_poly = P(SV(public_values), B.LAG)
    z_h = st.z_h()
    b1, b2, b3, b4, b5, b6 = st.b[0:6]
    a, b, c = P(SV(a_values), B.LAG), P(SV(b_values), B.LAG), P(SV(c_values), B.LAG)
    a_coeff = P(SV([b2, b1]), B.MONO) * z_h + a.i_ntt()
    b_coeff = P(SV([b4, b3]), B.MONO) * z_h + b.i_ntt()
    c_coeff = P(SV([b6, b5]), B.MONO) * z_h + c.i_ntt()
    st.wp.update(a=a, b=b, c=c, a_coeff=a_coeff, b_coeff=b_coeff, c_coeff=c_coeff, z_h_coeff=z_h)
    out = (B.commit(a_coeff), B.commit(b_coeff), B.commit(c_coeff))
    st.log["round_1"] = dict(a_coeff=a_coeff.values, b_coeff=b_coeff.values, c_coeff=c_coeff.values, commits=out)
    return out


def round_2(st, z_values_fn):
    """prover.rs:279-368; z_values_fn(a, b, c, s1, s2, s3, beta, gamma) -> z Lagrange values [n, 4]"""
    B, P = st.B, st.B.Polynomial
    beta, gamma = st.rand["beta"], st.rand["gamma"]
    zv = z_values_fn(st.wp["a"].values, st.wp["b"].values, st.wp["c"].values, st.pk["s1"].values, st.pk["s2"].values,
                     st.pk["s3"].values, S(beta), S(gamma))
    z = P(zv, B.LAG)
    b7, b8, b9 = st.b[6:9]
    z_coeff = P(SV([b9, b8, b7]), B.MONO) * st.wp["z_h_coeff"] + z.i_ntt()
    st.wp.update(z=z, z_coeff=z_coeff)
    z_1 = B.commit(z_coeff)
    st.log["round_2"] = dict(z=zv, z_coeff=z_coeff.values, commit=z_1)
    return z_1


def monomial_z_to_z_omega(B, z, omega):
    """prover.rs:661-674: coefficient i times omega^i"""
    vals, p = [], 1
    for v in ints(z.values):
        vals.append(v * p % Q)
        p = p * omega % Q
    return B.Polynomial(SV(vals), B.MONO)


def round_3(st):
    """prover.rs:370-500"""
    B, P, n = st.B, st.B.Polynomial, st.n
    coeff = {k: P(B.i_ntt_381(st.pk[k].values), B.MONO) for k in ("s1", "s2", "s3", "ql", "qr", "qm", "qo", "qc")}
    a, b, c, z = st.wp["a_coeff"], st.wp["b_coeff"], st.wp["c_coeff"], st.wp["z_coeff"]
    l1 = P(SV([1] + [0] * (n - 1)), B.LAG)
    z_h = st.z_h()
    gate = (a * coeff["ql"] + b * coeff["qr"] + a * b * coeff["qm"] + c * coeff["qo"]
            + st.public_input_poly.i_ntt() + coeff["qc"])
    roots_poly = P(B.i_ntt_381(B.roots_of_unity(n)), B.MONO)
    omega = root_of_unity(n)
    z_omega = monomial_z_to_z_omega(B, z, omega)
    beta, gamma, alpha = st.rand["beta"], st.rand["gamma"], st.rand["alpha"]
    perm = ((rlc(a, roots_poly, beta, gamma) * rlc(b, roots_poly * S(K1), beta, gamma) * rlc(c, roots_poly * S(K2), beta, gamma)) * z
            - (rlc(a, coeff["s1"], beta, gamma) * rlc(b, coeff["s2"], beta, gamma) * rlc(c, coeff["s3"], beta, gamma)) * z_omega)
    l1_coeff = P(B.i_ntt_381(l1.values), B.MONO)
    first_row = (z - S(1)) * l1_coeff
    all_constraints = gate + perm * S(alpha) + first_row * S(alpha * alpha % Q)
    t = all_constraints / z_h
    tv = t.values
    t_lo, t_mid, t_hi = P(tv[0:n], B.MONO), P(tv[n:2 * n], B.MONO), P(tv[2 * n:], B.MONO)      # prover.rs:649-659
    b10, b11 = st.b[9], st.b[10]
    x_pow_n = P(SV([0] * n + [1]), B.MONO)
    t_lo = t_lo + x_pow_n * S(b10)
    t_mid = t_mid + (x_pow_n * S(b11) - S(b10))
    t_hi = t_hi + S(-b11)
    st.pk_coeff = coeff
    st.wp.update(z_omega_coeff=z_omega, t_lo_coeff=t_lo, t_mid_coeff=t_mid, t_hi_coeff=t_hi)
    out = (B.commit(t_lo), B.commit(t_mid), B.commit(t_hi))
    st.log["round_3"] = dict(t=tv, t_lo=t_lo.values, t_mid=t_mid.values, t_hi=t_hi.values, commits=out)
    return out


def round_4(st):
    """prover.rs:502-541"""
    zeta = S(st.rand["zeta"])
    ev = dict(a_bar=st.wp["a_coeff"].coeffs_evaluate(zeta), b_bar=st.wp["b_coeff"].coeffs_evaluate(zeta),
              c_bar=st.wp["c_coeff"].coeffs_evaluate(zeta), s1_bar=st.pk_coeff["s1"].coeffs_evaluate(zeta),
              s2_bar=st.pk_coeff["s2"].coeffs_evaluate(zeta), z_omega_bar=st.wp["z_omega_coeff"].coeffs_evaluate(zeta))
    st.evals = {k: O.fr_to_int(v) for k, v in ev.items()}
    st.log["round_4"] = dict(st.evals)
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
    l1_coeff = P(B.i_ntt_381(SV([1] + [0] * (n - 1))), B.MONO)
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
    st.log["round_5"] = dict(r=r_coeff.values, w_zeta=w_zeta.values, w_zeta_omega=w_zeta_omega.values, commits=out)
    return out
