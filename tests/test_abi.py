"""The C-ABI library loads without a GPU and exports every symbol include/bp_msm_ntt.h declares; the
host-only entry points (partials, encodings, roots of unity) work on the CPU and agree with the oracle."""
import os
import re

import numpy as np
import pytest

import baby_plonk_rust_amd as bp
from baby_plonk_rust_amd import _lib
from oracle import oracle as O
from tests import bigint_model as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "bp_msm_ntt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bp_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_header_symbol():
    lib = bp.load()
    names = header_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SIGNATURES) == names          # python binding table and header agree


def test_rust_bindings_follow_the_header():
    """include/bp_msm_ntt.rs (the extern "C" block a maintainer of the Rust reference drops in as src/gpu.rs) is
    generated from the header: it is current, declares every entry point, and its arity agrees with the ctypes table the parity
    tests call through (same count of arguments, same pointer-vs-value pattern)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_rust_bindings", os.path.join(ROOT, "tools", "gen_rust_bindings.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    text = open(os.path.join(ROOT, "include", "bp_msm_ntt.rs")).read()
    assert text == gen.generate(), "run python tools/gen_rust_bindings.py"
    protos = {name: (args, ret) for name, args, ret in gen.prototypes(open(gen.HEADER).read())}
    assert sorted(protos) == header_symbols()
    import ctypes as C
    for name, (args, ret) in protos.items():
        res, argtypes = _lib.SIGNATURES[name]
        assert len(args) == len(argtypes), name
        for (pname, rty), cty in zip(args, argtypes):
            is_ptr_rust = rty.startswith("*")
            is_ptr_c = cty in (C.c_void_p, C.c_char_p) or hasattr(cty, "contents") or getattr(cty, "_type_", None) == "P"
            assert is_ptr_rust == is_ptr_c, (name, pname, rty, cty)
        assert (ret is None) == (res is None), name
    for const in ("BP_FR_MONT: c_int = 1", "BP_FR_BYTES_LE: c_int = 0", "BP_ERR_ASSERT: c_int = -11", "BP_MSM_BLOB_BYTES: usize = %d" % _lib.MSM_BLOB_BYTES):
        assert "pub const " + const + ";" in text, const


def test_library_links_rccl_and_comm_calls_fail_loudly_without_a_context():
    """the one-process-per-GPU collectives live UNDER the C ABI (capi_comm.hip): the shipped library is linked against librccl
    (DT_NEEDED), and without a context every comm entry point is an error, not a no-op"""
    import subprocess
    dyn = subprocess.check_output(["readelf", "-d", _lib.SO_PATH], text=True)
    assert re.search(r"NEEDED.*librccl\.so", dyn), dyn
    lib = bp.load()
    assert lib.bp_comm_init_rank(None, None, 0, 1) == -1 and lib.bp_comm_destroy(None) == -1
    assert lib.bp_msm_g1_allgather(None, 0, 0, None, 0, 1, 0, None) == -1 and lib.bp_ntt_columns_allgather(None, None, 10, 1) == -1
    assert lib.bp_comm_unique_id(None) == -1
    text = open(os.path.join(ROOT, "include", "bp_msm_ntt.h")).read()
    assert "#define BP_COMM_ID_BYTES %d" % _lib.COMM_ID_BYTES in text and "BP_ERR_COMM = -12" in text


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(bp.BpError) as e:
        bp.Context()
    assert e.value.code == -8


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "baby_plonk_rust_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                for needle in ("import oracle", "from oracle", "oracle/", "liboracle", "bp_oracle", "oracle."):
                    assert needle not in src, (needle, os.path.join(dirpath, f))


def test_host_partials_and_encodings():
    g = O.g1_generator()
    pts = [O.g1_mul(g, O.fr_from_int(k)) for k in (5, 11, M.Q - 3)] + [O.g1_identity()]
    raw = b"".join(p.tobytes() for p in pts)                     # G1Projective memory image = 144-byte partial
    assert bp.sum_partials(raw) == M.enc96(M.ec_mul(5 + 11 - 3))
    assert bp.sum_partials(b"") == M.enc96(None)
    assert bp.sum_partials(pts[3].tobytes()) == M.enc96(None)
    # bytes96 -> partial -> bytes96 round trip, and rejection of a non-canonical x
    enc = M.enc96(M.ec_mul(77))
    assert bp.sum_partials(bp.bytes96_to_partial(enc)) == enc
    assert bp.sum_partials(bp.bytes96_to_partial(M.enc96(None))) == M.enc96(None)
    with pytest.raises(bp.BpError):
        bp.bytes96_to_partial(bytes([0x1F]) + bytes([0xFF] * 95))


def test_root_of_unity_host():
    for n in (1, 2, 4, 8, 1 << 16, 1 << 24, 3, 1000):      # non powers of two: integer division, as written
        want = pow(M.ROOT_OF_UNITY, (1 << 32) // n, M.Q)
        assert bp.scalar_to_int(bp.root_of_unity(n)) == want
    with pytest.raises(bp.BpError):
        bp.root_of_unity(0)
    assert (bp.scalar_from_int(12345) == O.fr_from_int(12345)).all()


def test_host_point_compression_against_the_reference_fixture():
    """bp_g1_bytes96_to_compressed48 (G1Affine::to_compressed, g1.rs:221-244 -- the encoding the transcript absorbs and bp_prove
    emits) over the reference's own 1000 + 1000 wire vectors (src/tests/mod.rs:3-55); host-side, no GPU"""
    import numpy as np
    from baby_plonk_rust_amd import _lib
    here = os.path.dirname(__file__)
    unc = open(os.path.join(here, "golden", "g1_uncompressed_valid_test_vectors.dat"), "rb").read()
    comp = open(os.path.join(here, "golden", "g1_compressed_valid_test_vectors.dat"), "rb").read()
    lib = _lib.load()
    for i in range(1000):
        a = np.frombuffer(unc[96 * i: 96 * i + 96], dtype=np.uint8).copy()
        out = np.zeros(48, dtype=np.uint8)
        assert lib.bp_g1_bytes96_to_compressed48(a.ctypes.data, out.ctypes.data) == 0
        assert bytes(out) == comp[48 * i: 48 * i + 48], i
    bad = np.full(96, 0xFF, dtype=np.uint8)
    assert lib.bp_g1_bytes96_to_compressed48(bad.ctypes.data, out.ctypes.data) == -3


def _slot(point_int):
    """an accumulator slot as the MSM kernels leave it (msm_kernels.hpp proj28_slot): x | y | z, 14 limbs of 28 bits each of
    v * 2^392 mod p, + 2 pad words; built with Python integers only"""
    P = M.P
    x, y, z = (point_int[0], point_int[1], 1) if point_int is not None else (0, 1, 0)
    words = []
    for v in (x, y, z):
        m = v * pow(2, 392, P) % P
        words += [(m >> (28 * i)) & 0xFFFFFFF for i in range(14)]
    return np.array(words + [0, 0], dtype=np.uint32).tobytes()


def _blob(c, slots, tables, status=0):
    hdr = np.zeros(16, dtype=np.uint32)
    hdr[:6] = [0x424D5042, c, 1 if tables else len(slots), len(slots), int(tables), status]
    body = b"".join(slots)
    return hdr.tobytes() + body + bytes(_lib.MSM_BLOB_BYTES - 64 - len(body))


def test_blob_combine_host_side():
    """bp_msm_blobs_combine (the host end of the one-process-per-GPU exchange) on records written by hand: window sums
    T_w = k_w G combine to sum_w 2^(c w) k_w G (msm.rs:107-115); bit-plane records (fixed-base tables) to
    (A + sum_j 2^j T_j) G; records of equal layout are added slot by slot, others separately; no GPU"""
    rnd = __import__("random").Random(8)
    c, W = 5, 4
    total, blobs = 0, []
    for _ in range(3):                                   # three ranks, per-window layout
        ks = [rnd.randrange(1, M.Q) for _ in range(W)]
        ks[1] = 0                                        # an empty window (identity slot)
        total += sum(k << (c * w) for w, k in enumerate(ks))
        blobs.append(_blob(c, [_slot(M.ec_mul(k)) for k in ks], False))
    assert bp.combine_blobs(b"".join(blobs)) == M.enc96(M.ec_mul(total % M.Q))
    # a rank with the bit-plane layout (A, T_0 .. T_{c-2}) and one with a different window width
    ks = [rnd.randrange(1, M.Q) for _ in range(c)]
    total += ks[0] + sum(k << j for j, k in enumerate(ks[1:]))
    blobs.append(_blob(c, [_slot(M.ec_mul(k)) for k in ks], True))
    ks = [rnd.randrange(1, M.Q) for _ in range(3)]
    total += sum(k << (7 * w) for w, k in enumerate(ks))
    blobs.append(_blob(7, [_slot(M.ec_mul(k)) for k in ks], False))
    blobs.append(_blob(0, [], False))                    # an empty shard
    assert bp.combine_blobs(b"".join(blobs)) == M.enc96(M.ec_mul(total % M.Q))
    assert bp.combine_blobs(b"") == M.enc96(None)
    with pytest.raises(bp.BpError) as e:
        bp.combine_blobs(_blob(c, [_slot(M.ec_mul(1))] * W, False, status=1))
    assert e.value.code == -4
    with pytest.raises(bp.BpError):
        bp.combine_blobs(bytes(_lib.MSM_BLOB_BYTES))


def test_window_scalars_are_what_the_reference_walks():
    """msm.rs:83,119-139: bucket_msm(points, scalars, b, c) walks floor(b / c) windows of c bits from the top of the 256-bit image, so
    it multiplies by K >> (256 - c * floor(b / c)).  bp_msm_window_scalars (host-side, no GPU) against the oracle's literal
    get_c_bit_chunk walk recombined by Horner, for window parameters that drop bits and for those that do not; the reference's
    panics are errors."""
    from oracle import oracle as O
    lib = bp.load()
    sc = O.splitmix_scalars(40, 0xB17)
    sc[0] = 0
    sc[1] = bp.scalar_from_int(M.Q - 1)
    ints = O.fr_array_to_ints(sc)
    for b, c in ((256, 4), (256, 16), (256, 3), (256, 5), (256, 7), (255, 5), (128, 4), (200, 8), (64, 63), (17, 16), (256, 20)):
        out = np.zeros((len(sc), 32), dtype=np.uint8)
        assert lib.bp_msm_window_scalars(sc.ctypes.data, len(sc), bp.FR_MONT, b, c, out.ctypes.data) == 0, (b, c)
        k = b // c
        for i, v in enumerate(ints):
            walked = 0
            for w in range(k):
                walked = (walked << c) | int(O.lib.msm_get_c_bit_chunk(sc[i].ctypes.data, w, c))      # most significant window first
            assert walked == v >> (256 - k * c)
            assert int.from_bytes(out[i].tobytes(), "little") == walked, (b, c, i)
        le = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in ints), dtype=np.uint8).reshape(-1, 32).copy()
        out2 = np.zeros_like(out)
        assert lib.bp_msm_window_scalars(le.ctypes.data, len(le), bp.FR_BYTES_LE, b, c, out2.ctypes.data) == 0 and (out2 == out).all()
    out = np.zeros((len(sc), 32), dtype=np.uint8)
    for b, c in ((256, 0), (3, 4), (300, 4), (260, 13), (256, 64), (512, 100)):       # division by zero; t_points[0]; slice past bit 256 (twice); 1 << 64
        assert lib.bp_msm_window_scalars(sc.ctypes.data, len(sc), bp.FR_MONT, b, c, out.ctypes.data) == -1, (b, c)
    bad = np.full((1, 32), 0xFF, dtype=np.uint8)
    assert lib.bp_msm_window_scalars(bad.ctypes.data, 1, bp.FR_BYTES_LE, 256, 4, out.ctypes.data) == -4


def test_new_entry_points_reject_bad_arguments_without_a_gpu():
    """argument checks of the round-2 entry points that need no device: null pointers and bad counts are BP_ERR_INVALID_ARG (-1),
    a device list on a machine without GPUs is BP_ERR_NO_DEVICE (-8) -- never a crash, never a silent fallback"""
    import ctypes as C
    import torch
    lib = bp.load()
    h = C.c_void_p()
    ids = (C.c_int * 2)(0, 0)
    assert lib.bp_init_multi(None, ids, 2) == -1
    assert lib.bp_init_multi(C.byref(h), None, 2) == -1
    assert lib.bp_init_multi(C.byref(h), ids, 0) == -1
    assert lib.bp_init_multi(C.byref(h), ids, 65) == -1
    if not torch.cuda.is_available():
        assert lib.bp_init_multi(C.byref(h), ids, 2) == -8 and not h.value
    assert lib.bp_ctx_devices(None, None, 0) == -1
    out = np.zeros(96, dtype=np.uint8)
    assert lib.bp_msm_blobs_combine(None, 1, out.ctypes.data) == -1
    assert lib.bp_msm_blobs_combine(out.ctypes.data, 0, None) == -1
    assert lib.bp_msm_g1_blob_device(None, 1, 0, None, 0, 1, 0, None) == -1
    assert lib.bp_srs_load_projective144(None, None, 0, None) == -1
    assert lib.bp_msm_g1_projective144(None, None, 0, None, 0, 1, None) == -1
    assert lib.bp_srs_export_projective144(None, 1, 0, 0, None) == -1


def test_shipped_library_reads_no_environment():
    """the shipped library imports no getenv at all (every experiment knob is compiled out); the experiment
    build of the same sources does -- it is the only one that can read BP_* variables"""
    import subprocess
    here = os.path.join(ROOT, "baby_plonk_rust_amd")
    undefined = subprocess.run(["nm", "-D", "--undefined-only", os.path.join(here, "libbp_msm_ntt.so")], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undefined
    exp = os.path.join(here, "libbp_msm_ntt_exp.so")
    if os.path.exists(exp):
        assert "getenv" in subprocess.run(["nm", "-D", "--undefined-only", exp], capture_output=True, text=True, check=True).stdout
    lib = bp.load()
    lib.bp_version.restype = __import__("ctypes").c_char_p
    assert (b"+experiment" in lib.bp_version()) == _lib.EXPERIMENT
