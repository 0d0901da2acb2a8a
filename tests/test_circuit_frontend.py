"""The test-side restatement of the reference's preprocessing (tests/circuit_frontend.py) against the reference's own unit
test for it (src/program.rs:206-239) and the orientation rule of program.rs:126-137."""
from tests import bigint_model as M
from tests.circuit_frontend import make_gate_polynomials, make_s_polynomials

Q = M.Q


def test_make_s_polynomials_reference_test():
    """program.rs:206-239: constraints "c <== a * b", "b <== a * e", group order 8: s1[0] = w^1, s2[0] = 3 w^1"""
    n = 8
    rows, (s1, s2, s3) = make_s_polynomials([("a", "b", "c"), ("a", "e", "b")], n)
    w = [pow(M.omega(n), i, Q) for i in range(n)]
    assert s1[0] == w[1]
    assert s2[0] == 3 * w[1] % Q
    assert s1[1] == w[0] and s3[1] == 2 * w[0] % Q               # the other halves of the two 2-cycles
    assert s3[0] == 3 * w[0] % Q and s2[1] == 2 * w[1] % Q       # c and e are used once: fixed points
    # every sigma value is a cell label, and sigma is a permutation of the 3n labels
    labels = sorted((c + 1) * w[r] % Q for c in range(3) for r in range(n))
    assert sorted(s1 + s2 + s3) == labels


def test_cycle_orientation_of_longer_cycles():
    """program.rs:126-137: `s[next.column][next.row] = cell.label`: in a cycle c0 -> c1 -> c2 the cell c1 holds label(c0)"""
    n = 8
    rows, (s1, s2, s3) = make_s_polynomials([("x", None, None), ("x", None, None), ("x", None, None)], n)
    w = [pow(M.omega(n), i, Q) for i in range(n)]
    assert s1[1] == w[0] and s1[2] == w[1] and s1[0] == w[2]
    # the empty cells (rows 0..2 columns R, O, then rows 3..7) form one cycle in row-major order
    assert s2[0] == 3 * w[7] % Q                                  # first empty cell gets the last one's label (O, row 7)
    assert s3[0] == 2 * w[0] % Q and s2[1] == 3 * w[0] % Q


def test_make_gate_polynomials_shape():
    g = make_gate_polynomials([(1, 0, 0, 0, 0), (0, -1, -1, 1, 0)], 8)
    assert g["ql"] == [1] + [0] * 7 and g["qr"][1] == Q - 1 and g["qo"][1] == 1 and len(g["qc"]) == 8


def test_library_make_s_polynomials_matches_the_restatement():
    """bp_make_s_polynomials (host-side entry of the library, no GPU): the reference's own test case, the toy program, and random
    wirings with shared variables, against the line-by-line restatement above"""
    import random

    import numpy as np

    import baby_plonk_rust_amd as bp
    from oracle import oracle as O

    def run(wires, n):
        names = {}
        ids = np.zeros((n, 3), dtype=np.uint32)
        for r, ws in enumerate(wires):
            for c, name in enumerate(ws):
                if name is not None:
                    ids[r, c] = names.setdefault(name, 7 + 3 * len(names))     # arbitrary distinct non-zero ids
        got = [O.fr_array_to_ints(x) for x in bp.make_s_polynomials(ids)]
        _, want = make_s_polynomials(wires, n)
        assert got == want

    run([("a", "b", "c"), ("a", "e", "b")], 8)                                  # program.rs:206-239
    run([("e", None, None), ("a", "b", "c"), ("c", "d", "e")], 8)               # tests/verify_proof_test.rs
    run([("x", None, None), ("x", None, None), ("x", "x", "x")], 8)
    run([], 8)
    rnd = random.Random(4)
    for n in (8, 64, 1024):
        pool = ["v%d" % i for i in range(max(2, n // 3))]
        wires = [tuple(rnd.choice(pool + [None]) for _ in range(3)) for _ in range(rnd.randrange(1, n + 1))]
        run(wires, n)
    # a 2^16-row chain: the 2-cycles {(O, i), (L, i + 1)} of baby_plonk_rust_amd.synthetic
    n = 1 << 16
    ids = np.zeros((n, 3), dtype=np.uint32)
    ids[:, 0] = np.arange(1, n + 1)
    ids[:, 1] = np.arange(n + 1, 2 * n + 1)
    ids[:, 2] = np.arange(2, n + 2)
    s1, s2, s3 = (O.fr_array_to_ints(x) for x in bp.make_s_polynomials(ids))
    w = M.omega(n)
    assert s1[0] == 1 and s1[5] == 3 * pow(w, 4, Q) % Q and s3[4] == pow(w, 5, Q) and s2[9] == 2 * pow(w, 9, Q) % Q
    assert s3[n - 1] == 2 and s2[0] == 3 * pow(w, n - 1, Q) % Q     # id n + 1 is both (O, n - 1) and (R, 0): a 2-cycle
