"""shared helpers for the -m gpu parity tests (HIP path through the C ABI vs the CPU oracle)"""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests import bigint_model as M

Q = M.Q
# Tests that set BP_* knobs need the experiment build (make -C baby_plonk_rust_amd/csrc exp; run with BABY_PLONK_LIBRARY=exp):
# the shipped library reads no environment.  Under the default library they are skipped.
from baby_plonk_rust_amd import _lib as _bp_lib  # noqa: E402
EXPERIMENT = _bp_lib.EXPERIMENT
experiment = pytest.mark.skipif(not EXPERIMENT, reason="needs the experiment build: BABY_PLONK_LIBRARY=exp (make -C baby_plonk_rust_amd/csrc exp)")
NTHREADS = min(32, os.cpu_count() or 1)


def progression_bytes(n, a, d):
    return bytes(O.points_to_bytes96(O.points_progression(n, a, d)))


def closed_form(scalars_int, a, d, first=0):
    """sum_i s_i * (a + (first+i) d) * G as 96 bytes (independent big-int arithmetic)"""
    k = sum(s * (a + (first + i) * d) for i, s in enumerate(scalars_int)) % Q
    return M.enc96(M.ec_mul(k))


def oracle_dot(sc_mont, a, d):
    """k = sum_i s_i (a + i d) mod q; small inputs by independent Python big-ints, large ones in C"""
    sc_mont = np.ascontiguousarray(sc_mont, dtype=np.uint64).reshape(-1, 4)
    k = O.dot_progression(sc_mont, a, d)
    if len(sc_mont) <= 4096:
        ints = O.fr_array_to_ints(sc_mont)
        assert k == sum(s * (a + i * d) for i, s in enumerate(ints)) % Q
    return k
