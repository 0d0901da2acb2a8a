"""-m gpu: "next" rows of SURVEY.md section 8(f) -- building blocks of the prover rounds on the GPU, checked against the
oracle's restatement of src/prover.rs."""
import random

import numpy as np
import pytest

import baby_plonk_rust_amd as bp
from oracle import oracle as O
from tests import bigint_model as M

pytestmark = pytest.mark.gpu
Q = M.Q


def permutation_instance(n, rnd):
    """random copy-constraint cycles over the 3n wire cells; returns witness columns and sigma columns (ints)"""
    w = M.omega(n)
    labels = [[k * pow(w, i, Q) % Q for i in range(n)] for k in (1, 2, 3)]     # utils.rs:29-36 Cell::label, k1 = 2, k2 = 3
    cells = [(col, row) for col in range(3) for row in range(n)]
    rnd.shuffle(cells)
    wit = [[0] * n for _ in range(3)]
    sig = [[0] * n for _ in range(3)]
    pos = 0
    while pos < len(cells):
        size = min(len(cells) - pos, rnd.choice([1, 1, 2, 3, 5, 8]))
        cyc = cells[pos:pos + size]
        val = rnd.randrange(Q)
        for j, (col, row) in enumerate(cyc):
            ncol, nrow = cyc[(j + 1) % size]
            wit[col][row] = val
            sig[col][row] = labels[ncol][nrow]
        pos += size
    return wit, sig


@pytest.mark.parametrize("n", [8, 64, 4096, 1 << 16])
def test_round2_grand_product(n):
    rnd = random.Random(n)
    wit, sig = permutation_instance(n, rnd)
    cols = [bp.scalars_from_ints(v) if n <= 4096 else O.fr_array_from_ints(v) for v in wit + sig]
    beta, gamma = bp.scalar_from_int(rnd.randrange(Q)), bp.scalar_from_int(rnd.randrange(Q))
    z = bp.round_2_z(*cols, beta, gamma)
    assert (z == O.round2_z(*cols, beta, gamma)).all()
    assert bp.scalar_to_int(z[0]) == 1
    # a broken copy constraint: the reference's assert_eq!(z_n, 1) fires (prover.rs:319)
    bad = [c.copy() for c in cols]
    bad[0][n // 2] = bp.scalar_from_int(bp.scalar_to_int(bad[0][n // 2]) + 1)
    with pytest.raises(AssertionError):
        O.round2_z(*bad, beta, gamma)
    with pytest.raises(bp.BpError) as e:
        bp.round_2_z(*bad, beta, gamma)
    assert e.value.code == -11
    # a zero denominator: gamma = -(a_3 + beta * s1_3)  -> invert().unwrap() panics
    b_int = bp.scalar_to_int(beta)
    g0 = (-(wit[0][3] + b_int * sig[0][3])) % Q
    with pytest.raises(bp.BpError) as e:
        bp.round_2_z(*cols, beta, bp.scalar_from_int(g0))
    assert e.value.code == -7


def test_round2_identity_permutation_is_all_ones():
    n, rnd = 32, random.Random(5)
    w = M.omega(n)
    s1 = [pow(w, i, Q) for i in range(n)]
    cols = [bp.scalars_from_ints([rnd.randrange(Q) for _ in range(n)]) for _ in range(3)]
    cols += [bp.scalars_from_ints([k * x % Q for x in s1]) for k in (1, 2, 3)]
    z = bp.round_2_z(*cols, bp.scalar_from_int(11), bp.scalar_from_int(13))
    assert bp.scalars_to_ints(z) == [1] * n
