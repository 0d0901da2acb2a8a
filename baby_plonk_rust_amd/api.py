"""Host-side mirror of the reference's interface for the hot path, on top of the C ABI.

Names, argument meaning and error behaviour follow the reference (Rust) so the parity tests read like its
own tests; where the reference panics (assert!/unwrap) these raise BpError/AssertionError.

  reference (file:line)                               here
  ---------------------------------------------------------------------------------------------
  Setup::generate_srs(powers, tau)   setup.rs:12-31   Setup.generate_srs(powers, tau)
  Setup::commit(&poly)               setup.rs:32-37   Setup.commit(poly)
  BucketMSM::bucket_msm(p, s, b, c)  msm.rs:76-118    BucketMSM.bucket_msm(points, scalars, b, c)
  ntt_381 / i_ntt_381                utils.rs:63,106  ntt_381(values) / i_ntt_381(values)
  root_of_unity / roots_of_unity     utils.rs:39-52   root_of_unity(n) / roots_of_unity(n)
  Polynomial{values,basis} + ops     polynomial.rs    Polynomial(values, basis) with + - * / and methods

Scalars travel as numpy uint64 arrays of shape [n, 4]: the reference's Montgomery limbs (Scalar::to_array,
scalar.rs:35-40).  Points travel in the 96-byte uncompressed encoding (g1.rs:246-260)."""
import ctypes as C

import os

import numpy as np

from . import _lib
from ._lib import BASIS_LAGRANGE, BASIS_MONOMIAL, FR_BYTES_LE, FR_MONT, BpError

SRS_TABLES_OFF = 1      # bp_srs_precompute(window_bits): drop the fixed-base tables

Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
_R = pow(2, 256, Q)
_RINV = pow(_R, Q - 2, Q)


def scalar_from_int(v):
    """Scalar::from / from_raw: canonical integer -> Montgomery limbs (host-side big-int, O(1))"""
    m = (v % Q) * _R % Q
    return np.array([(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def scalar_to_int(limbs):
    m = sum(int(x) << (64 * i) for i, x in enumerate(np.asarray(limbs, dtype=np.uint64).reshape(4)))
    return m * _RINV % Q


def scalars_from_ints(vals):
    return np.stack([scalar_from_int(v) for v in vals]) if len(vals) else np.zeros((0, 4), dtype=np.uint64)


def scalars_to_ints(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return [scalar_to_int(a[i]) for i in range(len(a))]


def _fr_array(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.ndim == 1:
        a = a.reshape(-1, 4)
    assert a.ndim == 2 and a.shape[1] == 4, a.shape
    return a


class Context:
    """bp_ctx: one GPU (device = int, bp_init) or one context over several GPUs (device = list of ids, bp_init_multi:
    SRS and MSMs sharded by point range inside the library, everything else on the first device).
    The module keeps a default context per device."""

    def __init__(self, device=0):
        self._lib = _lib.load()
        h = C.c_void_p()
        if isinstance(device, (list, tuple)):
            ids = (C.c_int * len(device))(*device)
            rc = self._lib.bp_init_multi(C.byref(h), ids, len(device))
            self.devices = list(device)
        else:
            rc = self._lib.bp_init(C.byref(h), device)
            self.devices = [device]
        if rc != 0:
            raise BpError(rc, "bp_init", "no usable GPU: this package has no CPU fallback")
        self._h = h
        self.device = self.devices[0]
        self._async_keepalive = []       # host buffers handed to *_async calls: referenced until the next synchronize()

    def n_shards(self):
        return self._lib.bp_ctx_devices(self._h, None, 0)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc, where):
        if rc != 0:
            raise BpError(rc, where, (self._lib.bp_last_error(self._h) or b"").decode())

    # ---- thin wrappers -----------------------------------------------------------------------
    def srs_load(self, points96):
        buf = np.ascontiguousarray(np.frombuffer(bytes(points96), dtype=np.uint8))
        n, h = len(buf) // 96, C.c_uint64()
        self.check(self._lib.bp_srs_load(self._h, buf.ctypes.data, n, C.byref(h)), "bp_srs_load")
        return h.value

    def srs_load_projective144(self, points144):
        """n x 144 bytes: the in-memory image of G1Projective (x | y | z Montgomery limbs), normalised on the GPU"""
        buf = points144 if isinstance(points144, np.ndarray) else np.frombuffer(bytes(points144), dtype=np.uint8)
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        n, h = len(buf) // 144, C.c_uint64()
        self.check(self._lib.bp_srs_load_projective144(self._h, buf.ctypes.data, n, C.byref(h)), "bp_srs_load_projective144")
        return h.value

    def msm_projective144(self, points144, scalars, fmt=FR_MONT):
        """BucketMSM::bucket_msm(&[G1Projective], &[Scalar]) in one call, nothing cached: upload in two pieces, multiply the first behind the upload of the second"""
        buf = points144 if isinstance(points144, np.ndarray) else np.frombuffer(bytes(points144), dtype=np.uint8)
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        s = _fr_array(scalars) if fmt == FR_MONT else np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
        out = np.zeros(96, dtype=np.uint8)
        self.check(self._lib.bp_msm_g1_projective144(self._h, buf.ctypes.data, len(buf) // 144, s.ctypes.data, len(s), fmt, out.ctypes.data),
                   "bp_msm_g1_projective144")
        return out.tobytes()

    def srs_generate(self, powers, tau_int):
        t, h = np.frombuffer((tau_int % Q).to_bytes(32, "little"), dtype=np.uint8).copy(), C.c_uint64()
        self.check(self._lib.bp_srs_generate(self._h, powers, t.ctypes.data, C.byref(h)), "bp_srs_generate")
        return h.value

    def srs_generate_progression(self, n, a_int, d_int):
        a = np.frombuffer((a_int % Q).to_bytes(32, "little"), dtype=np.uint8).copy()
        d = np.frombuffer((d_int % Q).to_bytes(32, "little"), dtype=np.uint8).copy()
        h = C.c_uint64()
        self.check(self._lib.bp_srs_generate_progression(self._h, n, a.ctypes.data, d.ctypes.data, C.byref(h)),
                   "bp_srs_generate_progression")
        return h.value

    def srs_len(self, handle):
        n = C.c_size_t()
        self.check(self._lib.bp_srs_len(self._h, handle, C.byref(n)), "bp_srs_len")
        return n.value

    def srs_export(self, handle, first=0, n=None):
        n = self.srs_len(handle) - first if n is None else n
        out = np.zeros(96 * n, dtype=np.uint8)
        self.check(self._lib.bp_srs_export(self._h, handle, first, n, out.ctypes.data), "bp_srs_export")
        return bytes(out)

    def srs_export_projective144(self, handle, first=0, n=None):
        """the points as G1Projective memory images (144 bytes each, z = 1): what a Rust Vec<G1Projective> holds"""
        n = self.srs_len(handle) - first if n is None else n
        out = np.zeros(144 * n, dtype=np.uint8)
        self.check(self._lib.bp_srs_export_projective144(self._h, handle, first, n, out.ctypes.data), "bp_srs_export_projective144")
        return out

    def srs_free(self, handle):
        self.check(self._lib.bp_srs_free(self._h, handle), "bp_srs_free")

    def srs_precompute(self, handle, window_bits=0):
        """fixed-base window tables T[w][i] = 2^(c w) P_i for an SRS that serves many MSMs (0 = auto width,
        SRS_TABLES_OFF drops them; 256 + w = tables of every bit position for width-w NAF digits: experiment build only,
        the shipped library raises BpError(BP_ERR_INVALID_ARG) for it)"""
        self.check(self._lib.bp_srs_precompute(self._h, handle, window_bits), "bp_srs_precompute")
        return self.srs_table_info(handle)

    def srs_table_info(self, handle):
        c, w, b = C.c_uint32(), C.c_uint32(), C.c_uint64()
        self.check(self._lib.bp_srs_table_info(self._h, handle, C.byref(c), C.byref(w), C.byref(b)), "bp_srs_table_info")
        return {"window_bits": c.value, "windows": w.value, "bytes": b.value}

    def msm(self, handle, scalars, fmt=FR_MONT):
        s = _fr_array(scalars) if fmt == FR_MONT else np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
        out = np.zeros(96, dtype=np.uint8)
        self.check(self._lib.bp_msm_g1(self._h, handle, s.ctypes.data, len(s), fmt, out.ctypes.data), "bp_msm_g1")
        return bytes(out)

    def msm_partial(self, handle, scalars, first=0, fmt=FR_MONT, device_ptr=None, n=None):
        """144-byte projective partial over SRS[first:first+n]; scalars = host array or (device_ptr, n)"""
        out = np.zeros(144, dtype=np.uint8)
        if device_ptr is not None:
            rc = self._lib.bp_msm_g1_partial(self._h, handle, first, device_ptr, n, fmt, 1, out.ctypes.data)
        else:
            s = _fr_array(scalars)
            rc = self._lib.bp_msm_g1_partial(self._h, handle, first, s.ctypes.data, len(s), fmt, 0, out.ctypes.data)
        self.check(rc, "bp_msm_g1_partial")
        return bytes(out)

    def msm_blob_device(self, handle, d_blob_ptr, scalars=None, first=0, fmt=FR_MONT, device_ptr=None, n=None, wait=True):
        """one process per GPU: this rank's partial sums stay in HBM at d_blob_ptr (MSM_BLOB_BYTES), ready for the all-gather.
        wait=False: stream-ordered (bp_msm_g1_blob_device_async): enqueued on the context's stream (set_stream), not waited for"""
        fn = self._lib.bp_msm_g1_blob_device if wait else self._lib.bp_msm_g1_blob_device_async
        if device_ptr is not None:
            rc = fn(self._h, handle, first, device_ptr, n, fmt, 1, d_blob_ptr)
        else:
            s = _fr_array(scalars) if fmt == FR_MONT else np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
            if not wait:
                self._async_keepalive.append(s)      # the copy is stream-ordered (bp_msm_ntt.h): alive until synchronize()
            rc = fn(self._h, handle, first, s.ctypes.data, len(s), fmt, 0, d_blob_ptr)
        self.check(rc, "bp_msm_g1_blob_device" if wait else "bp_msm_g1_blob_device_async")

    def msm_blobs_sum_device(self, d_blobs_ptr, n_blobs, d_out_ptr, wait=True):
        """gathered records of equal layout -> one record, added slot by slot on the GPU (see bp_msm_blobs_sum_device)"""
        fn = self._lib.bp_msm_blobs_sum_device if wait else self._lib.bp_msm_blobs_sum_device_async
        self.check(fn(self._h, d_blobs_ptr, n_blobs, d_out_ptr), "bp_msm_blobs_sum_device")

    # ---- one process per GPU: the collective under the C ABI (capi_comm.hip) ----
    @staticmethod
    def comm_unique_id():
        """rank 0: the 128-byte id every rank passes to comm_init_rank (ncclGetUniqueId); the host carries it between the ranks"""
        out = np.zeros(_lib.COMM_ID_BYTES, dtype=np.uint8)
        rc = _lib.load().bp_comm_unique_id(out.ctypes.data)
        if rc != 0:
            raise BpError(rc, "bp_comm_unique_id")
        return bytes(out)

    def comm_init_rank(self, comm_id, rank, world):
        buf = np.frombuffer(bytes(comm_id), dtype=np.uint8).copy()
        assert len(buf) == _lib.COMM_ID_BYTES
        self.check(self._lib.bp_comm_init_rank(self._h, buf.ctypes.data, rank, world), "bp_comm_init_rank")

    def comm_info(self):
        r, w = C.c_int(), C.c_int()
        self.check(self._lib.bp_comm_info(self._h, C.byref(r), C.byref(w)), "bp_comm_info")
        return r.value, w.value

    def comm_destroy(self):
        self.check(self._lib.bp_comm_destroy(self._h), "bp_comm_destroy")

    def comm_set_timeout_ms(self, ms):
        """bound of every wait behind a collective and of ncclCommInitRank (default 120 000 ms; 0 = none): when it expires the
        communicator is aborted and the call returns BP_ERR_COMM -- an error, not a stall"""
        self.check(self._lib.bp_comm_set_timeout_ms(self._h, int(ms)), "bp_comm_set_timeout_ms")

    def comm_stats(self):
        n, ms = C.c_uint64(), C.c_uint32()
        self.check(self._lib.bp_comm_stats(self._h, C.byref(n), C.byref(ms)), "bp_comm_stats")
        return {"collectives": n.value, "timeout_ms": ms.value}

    def msm_allgather(self, handle, scalars=None, first=0, fmt=FR_MONT, device_ptr=None, n=None):
        """sum over ALL ranks' (point, scalar) pairs (bp_msm_g1_allgather): record -> ONE ncclAllGather -> device pre-sum -> one D2H"""
        out = np.zeros(96, dtype=np.uint8)
        if device_ptr is not None:
            rc = self._lib.bp_msm_g1_allgather(self._h, handle, first, device_ptr, n, fmt, 1, out.ctypes.data)
        else:
            s = _fr_array(scalars) if fmt == FR_MONT else np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
            rc = self._lib.bp_msm_g1_allgather(self._h, handle, first, s.ctypes.data, len(s), fmt, 0, out.ctypes.data)
        self.check(rc, "bp_msm_g1_allgather")
        return bytes(out)

    def comm_last_exchange_ms(self):
        ms = C.c_float()
        self.check(self._lib.bp_comm_last_exchange_ms(self._h, C.byref(ms)), "bp_comm_last_exchange_ms")
        return ms.value

    def ntt_columns_allgather(self, d_columns_ptr, log_n, columns_per_rank):
        """world x columns_per_rank columns of 2^log_n elements in HBM, this rank's block filled: ONE in-place ncclAllGather"""
        self.check(self._lib.bp_ntt_columns_allgather(self._h, d_columns_ptr, log_n, columns_per_rank), "bp_ntt_columns_allgather")

    def msm_stats(self):
        a, t, adds, c = C.c_float(), C.c_float(), C.c_uint64(), C.c_uint32()
        self.check(self._lib.bp_msm_last_stats(self._h, C.byref(a), C.byref(t), C.byref(adds), C.byref(c)), "bp_msm_last_stats")
        return {"accumulate_ms": a.value, "device_ms": t.value, "mixed_adds": adds.value, "window_bits": c.value,
                "tables": self._lib.bp_msm_last_used_tables(self._h) == 1}

    def msm_member_stats(self):
        """per member of a group context: upload (host scalars only), accumulate and whole-pipeline device times of the last MSM"""
        out = []
        for r in range(self.n_shards()):
            u, a, t, adds = C.c_float(), C.c_float(), C.c_float(), C.c_uint64()
            self.check(self._lib.bp_msm_last_member_stats(self._h, r, C.byref(u), C.byref(a), C.byref(t), C.byref(adds)), "bp_msm_last_member_stats")
            out.append({"member": r, "upload_ms": u.value, "accumulate_ms": a.value, "device_ms": t.value, "mixed_adds": adds.value})
        return out

    def ntt(self, values, inverse=False, fmt=FR_MONT):
        """one vector; raises like the reference's assert!(is_power_of_two(n)) (utils.rs:65,108)"""
        a = _fr_array(values).copy()
        n = len(a)
        if n == 0 or n & (n - 1):
            raise BpError(-2, "ntt_381", "length %d is not a power of two" % n)
        self.check(self._lib.bp_ntt_fr(self._h, a.ctypes.data, n.bit_length() - 1, int(inverse), fmt, 1, n), "bp_ntt_fr")
        return a

    def ntt_batch(self, values, inverse=False, stride=None, fmt=FR_MONT):
        """values [batch, stride, 4]: transforms the first 2^k entries of every row (independent columns)"""
        a = np.ascontiguousarray(values, dtype=np.uint64).copy()
        assert a.ndim == 3 and a.shape[2] == 4
        batch, row = a.shape[0], a.shape[1]
        n = stride if stride is not None else row
        if n == 0 or n & (n - 1) or n > row:
            raise BpError(-2, "ntt_381", "length %d is not a power of two" % n)
        self.check(self._lib.bp_ntt_fr(self._h, a.ctypes.data, n.bit_length() - 1, int(inverse), fmt, batch, row), "bp_ntt_fr")
        return a

    def ntt_device(self, ptr, log_n, inverse=False, batch=1, stride=None):
        self.check(self._lib.bp_ntt_fr_device(self._h, ptr, log_n, int(inverse), batch, stride if stride else (1 << log_n)),
                   "bp_ntt_fr_device")

    def ntt_device_async(self, ptr, log_n, inverse=False, batch=1, stride=None):
        """enqueue only (bp_ntt_fr_device_async); synchronize() waits"""
        self.check(self._lib.bp_ntt_fr_device_async(self._h, ptr, log_n, int(inverse), batch, stride if stride else (1 << log_n)),
                   "bp_ntt_fr_device_async")

    def ntt_stats(self):
        ms, p = C.c_float(), C.c_uint32()
        self.check(self._lib.bp_ntt_last_stats(self._h, C.byref(ms), C.byref(p)), "bp_ntt_last_stats")
        return {"device_ms": ms.value, "passes": p.value, "members": self._lib.bp_ntt_last_members(self._h)}

    def synthetic_scalars_device(self, ptr, n, seed):
        """fill HBM at `ptr` with n synthetic Montgomery scalars (the same stream the CPU baseline uses)"""
        self.check(self._lib.bp_fr_synthetic_device(self._h, ptr, n, seed), "bp_fr_synthetic_device")

    def set_stream(self, stream_ptr):
        self.check(self._lib.bp_set_stream(self._h, stream_ptr), "bp_set_stream")

    def synchronize(self):
        self.check(self._lib.bp_synchronize(self._h), "bp_synchronize")
        self._async_keepalive.clear()


_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


def sum_partials(partials144):
    """host-side: add 144-byte projective partials and return the 96-byte affine encoding"""
    lib = _lib.load()
    buf = np.ascontiguousarray(np.frombuffer(bytes(partials144), dtype=np.uint8))
    out = np.zeros(96, dtype=np.uint8)
    rc = lib.bp_g1_sum_partials(buf.ctypes.data, len(buf) // 144, out.ctypes.data)
    if rc != 0:
        raise BpError(rc, "bp_g1_sum_partials")
    return bytes(out)


def combine_blobs(blobs):
    """host-side: the gathered MSM records of all ranks (n x MSM_BLOB_BYTES, host bytes) -> 96-byte affine encoding"""
    lib = _lib.load()
    buf = np.ascontiguousarray(np.frombuffer(bytes(blobs), dtype=np.uint8))
    assert len(buf) % _lib.MSM_BLOB_BYTES == 0
    out = np.zeros(96, dtype=np.uint8)
    rc = lib.bp_msm_blobs_combine(buf.ctypes.data, len(buf) // _lib.MSM_BLOB_BYTES, out.ctypes.data)
    if rc != 0:
        raise BpError(rc, "bp_msm_blobs_combine")
    return bytes(out)


def bytes96_to_partial(b96):
    lib = _lib.load()
    src, out = np.frombuffer(bytes(b96), dtype=np.uint8).copy(), np.zeros(144, dtype=np.uint8)
    rc = lib.bp_g1_bytes96_to_partial(src.ctypes.data, out.ctypes.data)
    if rc != 0:
        raise BpError(rc, "bp_g1_bytes96_to_partial")
    return bytes(out)


# ------------------------------------------------------------------------------------------------
# reference-shaped API
# ------------------------------------------------------------------------------------------------
def root_of_unity(group_order, ctx=None):
    """utils.rs:39-43"""
    lib = _lib.load()
    out = np.zeros(4, dtype=np.uint64)
    rc = lib.bp_root_of_unity(group_order, FR_MONT, out.ctypes.data)
    if rc != 0:
        raise BpError(rc, "root_of_unity")
    return out


def roots_of_unity(group_order, ctx=None):
    """utils.rs:45-52"""
    ctx = ctx or default_context()
    out = np.zeros((group_order, 4), dtype=np.uint64)
    ctx.check(ctx._lib.bp_roots_of_unity(ctx._h, group_order, FR_MONT, out.ctypes.data), "roots_of_unity")
    return out


def ntt_381(elements, ctx=None):
    """utils.rs:63-81: natural-order DFT; asserts a power-of-two length"""
    return (ctx or default_context()).ntt(elements, inverse=False)


def i_ntt_381(elements, ctx=None):
    """utils.rs:106-129: inverse DFT including the 1/n factor"""
    return (ctx or default_context()).ntt(elements, inverse=True)


class BucketMSM:
    """src/msm.rs:8"""

    @staticmethod
    def bucket_msm(points96, scalars, b=256, c=4, ctx=None):
        """msm.rs:76-118.  points96: concatenated 96-byte encodings; scalars [n,4] Montgomery limbs.
        The reference walks floor(b / c) windows of c bits from the most significant end of the scalar's 256-bit image
        (msm.rs:83, 119-139): with b = 256 and c dividing 256 (its only call, setup.rs:36: b = 256, c = 4) that is the whole
        scalar; any other (b, c) silently drops the low 256 - c * floor(b / c) bits.  bp_msm_window_scalars reproduces exactly
        that (and the reference's panics as errors), so every (b, c) gives the reference's group element."""
        ctx = ctx or default_context()
        s = _fr_array(scalars)
        k = b // c if c > 0 else 0
        fmt = FR_MONT
        if not (0 < c <= 63 and k * c == 256):        # (c = 64 divides 256 too, but the reference's 1 << c bucket vector panics)
            eff = np.zeros((len(s), 32), dtype=np.uint8)
            rc = ctx._lib.bp_msm_window_scalars(s.ctypes.data, len(s), FR_MONT, b, c, eff.ctypes.data)
            if rc != 0:
                raise BpError(rc, "BucketMSM.bucket_msm", "the reference panics for b=%d, c=%d (msm.rs:24,105,132)" % (b, c))
            s, fmt = eff, FR_BYTES_LE
        h = ctx.srs_load(points96)
        try:
            return ctx.msm(h, s, fmt=fmt)
        finally:
            ctx.srs_free(h)


class Polynomial:
    """polynomial.rs:14-17; value semantics (every operator returns a new object)"""

    def __init__(self, values, basis, ctx=None):
        self.values = _fr_array(values).copy() if len(values) else np.zeros((0, 4), dtype=np.uint64)
        assert basis in (BASIS_LAGRANGE, BASIS_MONOMIAL)
        self.basis = basis
        self.ctx = ctx or default_context()

    def __len__(self):
        return len(self.values)

    def __eq__(self, other):
        return self.basis == other.basis and self.values.shape == other.values.shape and bool((self.values == other.values).all())

    def _binop(self, other, fn, name):
        if self.basis != other.basis:
            raise BpError(-5, name, "Basis must be the same")
        out = np.zeros((max(len(self) + len(other), 1), 4), dtype=np.uint64)
        n = C.c_size_t()
        c = self.ctx
        c.check(fn(c._h, self.values.ctypes.data, len(self), other.values.ctypes.data, len(other), self.basis, FR_MONT,
                   out.ctypes.data, C.byref(n)), name)
        return Polynomial(out[: n.value], self.basis, c)

    def _scalar(self, s, op, name):
        s = np.ascontiguousarray(s, dtype=np.uint64).reshape(4)
        out = np.zeros_like(self.values)
        c = self.ctx
        c.check(c._lib.bp_poly_scalar_op(c._h, self.values.ctypes.data, len(self), self.basis, s.ctypes.data, op, FR_MONT,
                                         out.ctypes.data), name)
        return Polynomial(out, self.basis, c)

    def __add__(self, other):            # polynomial.rs:57-117
        if isinstance(other, Polynomial):
            return self._binop(other, self.ctx._lib.bp_poly_add, "Polynomial + Polynomial")
        return self._scalar(other, 0, "Polynomial + Scalar")

    def __sub__(self, other):            # polynomial.rs:119-174
        if isinstance(other, Polynomial):
            return self._binop(other, self.ctx._lib.bp_poly_sub, "Polynomial - Polynomial")
        return self._scalar(other, 1, "Polynomial - Scalar")

    def __mul__(self, other):            # polynomial.rs:176-312
        if isinstance(other, Polynomial):
            return self._binop(other, self.ctx._lib.bp_poly_mul, "Polynomial * Polynomial")
        return self._scalar(other, 2, "Polynomial * Scalar")

    def __truediv__(self, other):        # polynomial.rs:314-380
        return self._binop(other, self.ctx._lib.bp_poly_div, "Polynomial / Polynomial")

    def rlc(self, other, beta, gamma):   # impl Rlc for Polynomial (utils.rs:170-175): self + other * beta + gamma
        return self + other * beta + gamma

    def coeffs_evaluate(self, x):        # polynomial.rs:34-45
        x = np.ascontiguousarray(x, dtype=np.uint64).reshape(4)
        out = np.zeros(4, dtype=np.uint64)
        c = self.ctx
        c.check(c._lib.bp_poly_evaluate(c._h, self.values.ctypes.data, len(self), self.basis, x.ctypes.data, FR_MONT,
                                        out.ctypes.data), "coeffs_evaluate")
        return out

    def ntt(self):                       # polynomial.rs:47-51
        if self.basis != BASIS_MONOMIAL:
            raise BpError(-5, "Polynomial.ntt", "needs the Monomial basis")
        return Polynomial(self.ctx.ntt(self.values, inverse=False), BASIS_LAGRANGE, self.ctx)

    def i_ntt(self):                     # polynomial.rs:52-55
        if self.basis != BASIS_LAGRANGE:
            raise BpError(-5, "Polynomial.i_ntt", "needs the Lagrange basis")
        return Polynomial(self.ctx.ntt(self.values, inverse=True), BASIS_MONOMIAL, self.ctx)

    def shift_left(self, n):             # polynomial.rs:22-33 (index permutation, host side)
        if self.basis != BASIS_LAGRANGE:
            raise BpError(-5, "Polynomial.shift_left", "needs the Lagrange basis")
        return np.roll(self.values, -(n % len(self)), axis=0)


class DevicePolynomial:
    """polynomial.rs:14-17 with the coefficient / value vector resident in HBM (torch CUDA tensor, int64 [n, 4] =
    Montgomery limbs).  Same operators and rules as Polynomial; nothing but O(1) scalars crosses PCIe."""

    def __init__(self, values, basis, ctx=None):
        import torch
        self.ctx = ctx or default_context()
        assert basis in (BASIS_LAGRANGE, BASIS_MONOMIAL)
        self.basis = basis
        if isinstance(values, torch.Tensor):
            self.t = values
        else:
            host = _fr_array(values) if len(values) else np.zeros((0, 4), dtype=np.uint64)
            self.t = torch.from_numpy(host.view(np.int64).copy()).to(torch.device("cuda", self.ctx.device))
            torch.cuda.current_stream().synchronize()

    @staticmethod
    def empty(n, basis, ctx):
        import torch
        return DevicePolynomial(torch.empty((max(n, 1), 4), dtype=torch.int64, device=torch.device("cuda", ctx.device))[:n], basis, ctx)

    def __len__(self):
        return self.t.shape[0]

    @property
    def values(self):
        return self.t.cpu().numpy().view(np.uint64)

    def _ptr(self):
        return self.t.data_ptr() if len(self) else 0

    def _binop(self, other, fn, name, out_len):
        if self.basis != other.basis:
            raise BpError(-5, name, "Basis must be the same")
        out = DevicePolynomial.empty(out_len, self.basis, self.ctx)
        n = C.c_size_t()
        c = self.ctx
        c.check(fn(c._h, self._ptr(), len(self), other._ptr(), len(other), self.basis, out._ptr(), C.byref(n)), name)
        out.t = out.t[: n.value]
        return out

    def _scalar(self, s, op, name):
        s = np.ascontiguousarray(s, dtype=np.uint64).reshape(4)
        out = DevicePolynomial.empty(len(self), self.basis, self.ctx)
        c = self.ctx
        c.check(c._lib.bp_poly_scalar_op_device(c._h, self._ptr(), len(self), self.basis, s.ctypes.data, op, out._ptr()), name)
        return out

    def __add__(self, o):
        if isinstance(o, DevicePolynomial):
            return self._binop(o, self.ctx._lib.bp_poly_add_device, "Polynomial + Polynomial", max(len(self), len(o)))
        return self._scalar(o, 0, "Polynomial + Scalar")

    def __sub__(self, o):
        if isinstance(o, DevicePolynomial):
            return self._binop(o, self.ctx._lib.bp_poly_sub_device, "Polynomial - Polynomial", max(len(self), len(o)))
        return self._scalar(o, 1, "Polynomial - Scalar")

    def __mul__(self, o):
        if isinstance(o, DevicePolynomial):
            return self._binop(o, self.ctx._lib.bp_poly_mul_device, "Polynomial * Polynomial", len(self) + len(o))
        return self._scalar(o, 2, "Polynomial * Scalar")

    def __truediv__(self, o):
        return self._binop(o, self.ctx._lib.bp_poly_div_device, "Polynomial / Polynomial", max(len(self), 1))

    def rlc(self, other, beta, gamma):   # impl Rlc for Polynomial (utils.rs:170-175)
        return self + other * beta + gamma

    def coeffs_evaluate(self, x):
        x = np.ascontiguousarray(x, dtype=np.uint64).reshape(4)
        out = np.zeros(4, dtype=np.uint64)
        c = self.ctx
        c.check(c._lib.bp_poly_evaluate_device(c._h, self._ptr(), len(self), self.basis, x.ctypes.data, out.ctypes.data), "coeffs_evaluate")
        return out

    def _transform(self, inverse, need, to):
        if self.basis != need:
            raise BpError(-5, "Polynomial.ntt/i_ntt", "wrong basis")
        n = len(self)
        if n == 0 or n & (n - 1):
            raise BpError(-2, "ntt_381", "length %d is not a power of two" % n)
        out = DevicePolynomial(self.t.clone(), to, self.ctx)
        import torch
        torch.cuda.current_stream().synchronize()
        self.ctx.ntt_device(out._ptr(), n.bit_length() - 1, inverse=inverse)
        return out

    def ntt(self):
        return self._transform(False, BASIS_MONOMIAL, BASIS_LAGRANGE)

    def i_ntt(self):
        return self._transform(True, BASIS_LAGRANGE, BASIS_MONOMIAL)

    def scale_powers(self, w):
        """p(x) -> p(w x)  (prover.rs:661-674)"""
        w = np.ascontiguousarray(w, dtype=np.uint64).reshape(4)
        out = DevicePolynomial.empty(len(self), self.basis, self.ctx)
        c = self.ctx
        c.check(c._lib.bp_poly_scale_powers_device(c._h, self._ptr(), len(self), w.ctypes.data, out._ptr()), "scale_powers")
        return out

    def slice(self, lo, hi=None):
        return DevicePolynomial(self.t[lo:hi].contiguous(), self.basis, self.ctx)


def commit_device(setup, poly):
    """Setup::commit (setup.rs:32-37) of an HBM-resident polynomial"""
    c = setup.ctx
    out = np.zeros(96, dtype=np.uint8)
    c.check(c._lib.bp_commit_device(c._h, setup.handle, poly._ptr(), len(poly), poly.basis, out.ctypes.data), "Setup.commit")
    return bytes(out)


def commit_many_device(setup, polys):
    """several Setup::commit calls in one (bp_commit_many_device): the pipelines of the commitments overlap"""
    c = setup.ctx
    k = len(polys)
    if k == 0:
        return []
    ptrs = (C.c_void_p * k)(*[p._ptr() for p in polys])
    lens = (C.c_size_t * k)(*[len(p) for p in polys])
    out = np.zeros(96 * k, dtype=np.uint8)
    c.check(c._lib.bp_commit_many_device(c._h, setup.handle, ptrs, lens, k, polys[0].basis, out.ctypes.data), "Setup.commit (many)")
    return [bytes(out[96 * i: 96 * (i + 1)]) for i in range(k)]


def round_2_z_device(a, b, c, s1, s2, s3, beta, gamma, k1=None, k2=None):
    """prover.rs:279-319 on DevicePolynomial columns -> DevicePolynomial (Lagrange)"""
    ctx = a.ctx
    n = len(a)
    k1 = scalar_from_int(2) if k1 is None else np.ascontiguousarray(k1, dtype=np.uint64)
    k2 = scalar_from_int(3) if k2 is None else np.ascontiguousarray(k2, dtype=np.uint64)
    beta, gamma = np.ascontiguousarray(beta, dtype=np.uint64), np.ascontiguousarray(gamma, dtype=np.uint64)
    out = DevicePolynomial.empty(n, BASIS_LAGRANGE, ctx)
    ctx.check(ctx._lib.bp_grand_product_device(ctx._h, a._ptr(), b._ptr(), c._ptr(), s1._ptr(), s2._ptr(), s3._ptr(), n, beta.ctypes.data,
                                               gamma.ctypes.data, k1.ctypes.data, k2.ctypes.data, out._ptr()), "round_2 grand product")
    return out


def roots_of_unity_device(n, ctx=None):
    ctx = ctx or default_context()
    out = DevicePolynomial.empty(n, BASIS_LAGRANGE, ctx)
    ctx.check(ctx._lib.bp_roots_of_unity_device(ctx._h, n, out._ptr()), "roots_of_unity")
    return out


def round_2_z(a, b, c, s1, s2, s3, beta, gamma, k1=None, k2=None, ctx=None):
    """prover.rs:279-319: Lagrange values z_0..z_{n-1} of the permutation grand product (raises where the reference panics)"""
    ctx = ctx or default_context()
    cols = [_fr_array(x) for x in (a, b, c, s1, s2, s3)]
    n = len(cols[0])
    assert all(len(x) == n for x in cols)
    k1 = scalar_from_int(2) if k1 is None else np.ascontiguousarray(k1, dtype=np.uint64)
    k2 = scalar_from_int(3) if k2 is None else np.ascontiguousarray(k2, dtype=np.uint64)
    beta, gamma = np.ascontiguousarray(beta, dtype=np.uint64), np.ascontiguousarray(gamma, dtype=np.uint64)
    out = np.zeros((n, 4), dtype=np.uint64)
    ctx.check(ctx._lib.bp_grand_product(ctx._h, *[x.ctypes.data for x in cols], n, beta.ctypes.data, gamma.ctypes.data, k1.ctypes.data,
                                        k2.ctypes.data, FR_MONT, out.ctypes.data), "round_2 grand product")
    return out


class Setup:
    """src/setup.rs:7-10 (G1 part; x_2 in G2 belongs to the verifier's pairing, out of scope).
    A Setup serves every commitment of a prover, so it builds the SRS's fixed-base window tables once
    (bp_srs_precompute) unless tables=False or BP_SRS_TABLES=0."""

    def __init__(self, handle, ctx, tables=True):
        self.handle, self.ctx, self.tables_error = handle, ctx, None
        if tables and os.environ.get("BP_SRS_TABLES", "1") != "0":
            try:
                ctx.srs_precompute(handle, 0)
            except BpError as e:          # tables are an optimisation (they need windows x 128 B per point of HBM): commits work without
                self.tables_error = e

    @staticmethod
    def generate_srs(powers, tau_int, ctx=None, tables=True):
        """setup.rs:12-31: [G, tau G, ..., tau^(powers-1) G], generated and kept on the GPU"""
        ctx = ctx or default_context()
        return Setup(ctx.srs_generate(powers, tau_int), ctx, tables)

    @staticmethod
    def from_points(points96, ctx=None, tables=True):
        ctx = ctx or default_context()
        return Setup(ctx.srs_load(points96), ctx, tables)

    def powers_of_x(self):
        return self.ctx.srs_export(self.handle)

    def commit(self, polynomial):
        """setup.rs:32-37: asserts the Monomial basis, then bucket_msm over the whole SRS"""
        c = self.ctx
        out = np.zeros(96, dtype=np.uint8)
        c.check(c._lib.bp_commit(c._h, self.handle, polynomial.values.ctypes.data, len(polynomial), polynomial.basis, FR_MONT,
                                 out.ctypes.data), "Setup.commit")
        return bytes(out)


CIRCUIT_COLUMNS = ("ql", "qr", "qm", "qo", "qc", "s1", "s2", "s3")


class Circuit:
    """CommonPreprocessedInput (src/program.rs:34-50) resident in HBM: the eight Lagrange columns QL QR QM QO QC S1 S2 S3
    of a 2^log_n-row circuit, uploaded once (bp_circuit_load)"""

    def __init__(self, columns, ctx=None):
        self.ctx = ctx or default_context()
        cols = [_fr_array(columns[k]) for k in CIRCUIT_COLUMNS] if isinstance(columns, dict) else [_fr_array(c) for c in columns]
        n = len(cols[0])
        if len(cols) != 8 or any(len(c) != n for c in cols) or n < 8 or n & (n - 1):
            raise BpError(-2, "Circuit", "eight columns of equal power-of-two length >= 8 expected")
        self.group_order = n
        ptrs = (C.c_void_p * 8)(*[c.ctypes.data for c in cols])
        h = C.c_uint64()
        self.ctx.check(self.ctx._lib.bp_circuit_load(self.ctx._h, n.bit_length() - 1, ptrs, FR_MONT, 0, C.byref(h)), "bp_circuit_load")
        self.handle = h.value

    def free(self):
        self.ctx.check(self.ctx._lib.bp_circuit_free(self.ctx._h, self.handle), "bp_circuit_free")

    def commitments(self, setup):
        """Verifier::new (src/verifier.rs:61-68): {column name: 96-byte commitment of its coefficient form}"""
        out = np.zeros(768, dtype=np.uint8)
        self.ctx.check(self.ctx._lib.bp_circuit_commitments(self.ctx._h, setup.handle, self.handle, out.ctypes.data), "bp_circuit_commitments")
        return {k: bytes(out[96 * i: 96 * i + 96]) for i, k in enumerate(CIRCUIT_COLUMNS)}


class Prover:
    """src/prover.rs:50-175 behind bp_prove: rounds 1-5 on the GPU, Fiat-Shamir transcript on the host"""

    def __init__(self, setup, circuit):
        assert setup.ctx is circuit.ctx
        self.setup, self.circuit, self.ctx = setup, circuit, setup.ctx

    def prove_with_blinding(self, a, b, c, public_input, blinders):
        """a, b, c: the three Lagrange wire columns (prover.rs:186-227); public_input: the Lagrange column of prover.rs:114-127
        or None; blinders: 11 ints b1..b11 (the reference draws them from thread_rng, prover.rs:108-110).  Returns the 624-byte
        proof: 9 compressed G1 points in Proof field order (verifier.rs:23-40) then the 6 evaluations, 32 bytes LE each."""
        n = self.circuit.group_order
        cols = [_fr_array(v) for v in (a, b, c)]
        pi = None if public_input is None else _fr_array(public_input)
        if any(len(v) != n for v in cols) or (pi is not None and len(pi) != n):
            raise BpError(-6, "prove", "witness columns must have group_order entries")
        bl = np.frombuffer(b"".join((int(v) % Q).to_bytes(32, "little") for v in blinders), dtype=np.uint8).copy()
        if len(bl) != 352:
            raise BpError(-2, "prove", "11 blinders expected")
        out = np.zeros(624, dtype=np.uint8)
        c_ = self.ctx
        c_.check(c_._lib.bp_prove(c_._h, self.setup.handle, self.circuit.handle, cols[0].ctypes.data, cols[1].ctypes.data, cols[2].ctypes.data,
                                  None if pi is None else pi.ctypes.data, FR_MONT, 0, bl.ctypes.data, out.ctypes.data), "bp_prove")
        return bytes(out)

    def prove_device(self, a_ptr, b_ptr, c_ptr, pi_ptr, blinders):
        """same with the witness columns already in HBM (Montgomery limbs)"""
        bl = np.frombuffer(b"".join((int(v) % Q).to_bytes(32, "little") for v in blinders), dtype=np.uint8).copy()
        out = np.zeros(624, dtype=np.uint8)
        c_ = self.ctx
        c_.check(c_._lib.bp_prove(c_._h, self.setup.handle, self.circuit.handle, a_ptr, b_ptr, c_ptr, pi_ptr, FR_MONT, 1, bl.ctypes.data,
                                  out.ctypes.data), "bp_prove")
        return bytes(out)

    def last_stats(self):
        r, t = (C.c_float * 5)(), C.c_float()
        self.ctx.check(self.ctx._lib.bp_prove_last_stats(self.ctx._h, r, C.byref(t)), "bp_prove_last_stats")
        return {"round_ms": list(r), "total_ms": t.value}


def make_s_polynomials(wire_ids):
    """Program::make_s_polynomials (src/program.rs:76-147) in O(n log n) on the host (no GPU): wire_ids [n, 3] uint32, 0 = empty
    wire, equal ids = the same variable -> (s1, s2, s3) Lagrange columns as [n, 4] Montgomery limbs"""
    ids = np.ascontiguousarray(wire_ids, dtype=np.uint32)
    n = ids.shape[0]
    if ids.ndim != 2 or ids.shape[1] != 3 or n == 0 or n & (n - 1):
        raise BpError(-2, "make_s_polynomials", "wire_ids must be [2^k, 3]")
    out = [np.zeros((n, 4), dtype=np.uint64) for _ in range(3)]
    rc = _lib.load().bp_make_s_polynomials(n.bit_length() - 1, ids.ctypes.data, out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data)
    if rc:
        raise BpError(rc, "bp_make_s_polynomials", "")
    return tuple(out)


def transcript_test_vector():
    """merlin's conformance vector through the library's host transcript (no GPU needed)"""
    out = np.zeros(32, dtype=np.uint8)
    rc = _lib.load().bp_transcript_test_vector(out.ctypes.data)
    if rc:
        raise BpError(rc, "bp_transcript_test_vector", "")
    return bytes(out)
