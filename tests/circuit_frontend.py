"""Restatement of the two preprocessing functions of the reference's circuit front-end that the prover's inputs depend on
(src/program.rs:52-147), for building test circuits.  TEST INFRASTRUCTURE: the front-end (parser, Program, Assembly) is
outside the accelerated path; the library receives its OUTPUT (eight Lagrange columns)."""
from tests import bigint_model as M

Q = M.Q


def make_s_polynomials(wires, n):
    """program.rs:76-147.  wires: one (L, R, O) tuple of variable names per constraint, None = empty wire; rows beyond the
    constraints are empty.  Cells of one variable form a cycle in row-major order (L, R, O inside a row, :80-88; the empty
    cells of constraint rows and of the unused rows share the variable None, :92-99); walking the cycle, the NEXT cell
    receives THIS cell's label (:126-137); label(column, row) = column * w^row with columns 1, 2, 3 (utils.rs:29-36)."""
    rows = list(wires) + [(None, None, None)] * (n - len(wires))
    om = M.omega(n)
    uses = {}
    for row, ws in enumerate(rows):
        for col, name in enumerate(ws):
            uses.setdefault(name, []).append((col, row))
    s = [[pow(om, i, Q) for i in range(n)], [2 * pow(om, i, Q) % Q for i in range(n)], [0] * n]      # :101-118 initial values
    for cells in uses.values():
        for i, (col, row) in enumerate(cells):
            ncol, nrow = cells[(i + 1) % len(cells)]
            s[ncol][nrow] = (col + 1) * pow(om, row, Q) % Q
    return rows, s


def make_gate_polynomials(gates, n):
    """program.rs:52-75: gates = one (L, R, M, O, C) coefficient tuple per constraint, zero rows after them"""
    cols = [[g[k] % Q for g in gates] + [0] * (n - len(gates)) for k in range(5)]
    return dict(ql=cols[0], qr=cols[1], qm=cols[2], qo=cols[3], qc=cols[4])
