"""Independent big-integer model of the mathematics (NOT a port of the Rust, NOT the oracle):
plain Python ints, affine short-Weierstrass formulas, textbook DFT.  Used to cross-check the
C oracle and to derive closed-form answers for the GPU tests."""

Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
GX = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
GY = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
ROOT_OF_UNITY = pow(7, (Q - 1) >> 32, Q)


def inv(a, m):
    return pow(a, m - 2, m)


def ec_add(p1, p2):
    """affine addition on y^2 = x^3 + 4; None = identity"""
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * inv(2 * y1, P) % P
    else:
        lam = (y2 - y1) * inv(x2 - x1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return x3, (lam * (x1 - x3) - y1) % P


def ec_mul(k, pt=(GX, GY)):
    k %= Q
    acc = None
    while k:
        if k & 1:
            acc = ec_add(acc, pt)
        pt = ec_add(pt, pt)
        k >>= 1
    return acc


def enc96(pt):
    if pt is None:
        return bytes([0x40]) + bytes(95)
    return pt[0].to_bytes(48, "big") + pt[1].to_bytes(48, "big")


def enc48(pt):
    if pt is None:
        return bytes([0xC0]) + bytes(47)
    b = bytearray(pt[0].to_bytes(48, "big"))
    b[0] |= 0x80
    if pt[1] > (P - 1) // 2:
        b[0] |= 0x20
    return bytes(b)


def dec48(b):
    """G1Affine::from_compressed (g1.rs:324-388): y = sqrt(x^3 + 4) with the sign bit choosing the larger root"""
    if b[0] & 0x40:
        return None
    x = int.from_bytes(bytes([b[0] & 0x1F]) + bytes(b[1:48]), "big")
    y = pow((x * x * x + 4) % P, (P + 1) // 4, P)            # p = 3 mod 4
    assert y * y % P == (x * x * x + 4) % P
    if (y > (P - 1) // 2) != bool(b[0] & 0x20):
        y = P - y
    return (x, y)


def omega(n):
    return pow(ROOT_OF_UNITY, (1 << 32) // n, Q)


def dft(vals, inverse=False):
    n = len(vals)
    w = omega(n)
    if inverse:
        w = inv(w, Q)
    out = []
    for x in range(n):
        s = 0
        for y, v in enumerate(vals):
            s += v * pow(w, x * y, Q)
        s %= Q
        if inverse:
            s = s * inv(n, Q) % Q
        out.append(s)
    return out


def splitmix64(state):
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


def splitmix_scalar(i, seed):
    """element i of the synthetic scalar stream"""
    s = (seed + 0x9E3779B97F4A7C15 * 8 * i) & 0xFFFFFFFFFFFFFFFF
    v = 0
    for k in range(8):
        s, z = splitmix64(s)
        v |= z << (64 * k)
    return v % Q
