"""Synthetic PLONK circuit for benchmarks (BASELINE.json configs[4]: "synthetic witness, random SRS"): n multiplication
gates z_i = x_i * y_i chained by copy constraints x_{i+1} = z_i -- the rows the reference's parser emits for
`c <== a * b` (src/assembly.rs:30-81: qm = -1, qo = 1), sigma columns as src/program.rs:76-147 would build them
(cell labels (column + 1) * w^row, utils.rs:29-36).  Host-side Python big-int code: it stands in for the reference's
circuit front-end, which is outside the accelerated path."""
import random

import numpy as np

from . import _lib

Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ROOT_OF_UNITY = pow(7, (Q - 1) >> 32, Q)          # scalar.rs ROOT_OF_UNITY: 2^32-th root


def ints_to_mont(vals):
    """list of ints -> [n, 4] uint64 Montgomery limbs (host-side conversion entry of the library)"""
    raw = np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint8).copy()
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    rc = _lib.load().bp_fr_convert(raw.ctypes.data, len(vals), 0, 1, out.ctypes.data)
    assert rc == 0
    return out


def chained_multiplications(n, seed):
    """-> (wire columns [a, b, c], circuit columns {ql qr qm qo qc s1 s2 s3}), all Lagrange, Montgomery limbs"""
    rnd = random.Random(seed)
    x = rnd.randrange(Q)
    a, b, c = [0] * n, [0] * n, [0] * n
    for i in range(n):
        y = rnd.getrandbits(250)
        a[i], b[i], c[i] = x, y, x * y % Q
        x = c[i]
    # wires: row i = (x_i, y_i, z_i) with x_{i+1} = z_i -> variable ids: x_i = z_{i-1} shares id i, y_i is used once, z_i has id i + 1
    ids = np.zeros((n, 3), dtype=np.uint32)
    ids[:, 0] = np.arange(1, n + 1, dtype=np.uint32)
    ids[:, 1] = np.arange(n + 2, 2 * n + 2, dtype=np.uint32)
    ids[:, 2] = np.arange(2, n + 2, dtype=np.uint32)
    from .api import make_s_polynomials
    s1m, s2m, s3m = make_s_polynomials(ids)                      # Program::make_s_polynomials (program.rs:76-147)
    zero = np.zeros((n, 4), dtype=np.uint64)
    pk = dict(ql=zero, qr=zero, qm=ints_to_mont([Q - 1] * n), qo=ints_to_mont([1] * n), qc=zero,
              s1=s1m, s2=s2m, s3=s3m)
    return [ints_to_mont(a), ints_to_mont(b), ints_to_mont(c)], pk
