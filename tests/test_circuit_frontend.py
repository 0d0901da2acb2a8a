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
