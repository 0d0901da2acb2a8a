"""Loader of libbp_msm_ntt.so (the C ABI of include/bp_msm_ntt.h).

There is no CPU fallback: if the HIP library is missing or no GPU is usable, every compute entry point
raises.  Loading the library and resolving its symbols works without a GPU (the CPU test-suite checks
that every symbol of the header is exported)."""
import ctypes as C
import os

# torch bundles its own libamdhip64.so.7; importing it first makes the dynamic loader reuse that one runtime
# for libbp_msm_ntt.so too (same SONAME).  Two HIP runtimes in one process cannot both see the GPU.
import torch  # noqa: F401  (plumbing: device memory, streams, torch.distributed)

_HERE = os.path.dirname(os.path.abspath(__file__))
# BABY_PLONK_LIBRARY=exp selects the experiment build (csrc/Makefile `make exp`: the only build that reads BP_* knobs); the
# shipped library reads no environment at all, so the choice is made here, on the Python side of the boundary.
# (a value ending in .so names a library file directly: A/B runs of two builds of the shipped library on one box)
_choice = os.environ.get("BABY_PLONK_LIBRARY", "")
SO_PATH = _choice if _choice.endswith(".so") else os.path.join(_HERE, "libbp_msm_ntt_exp.so" if _choice == "exp" else "libbp_msm_ntt.so")


def _is_experiment_build():
    """what the library itself says (bp_version() ends in "+experiment" for -DBP_EXPERIMENT builds), so that a library named by path
    is classified by its contents, not by the spelling of the environment value; bp_version needs no GPU.
    torch is imported above, so the loader has already mapped torch's libamdhip64 / librccl and reuses them for this library's
    DT_NEEDED entries of the same SONAME.  A library that exists but cannot be loaded here (librccl or HIP unresolvable on a box without
    ROCm) must not break `import baby_plonk_rust_amd` for code that never calls load(): classified by the environment value then, and
    load() raises the loader's error when it is really needed."""
    if not os.path.exists(SO_PATH):
        return _choice == "exp"
    try:
        fn = C.CDLL(SO_PATH).bp_version
    except OSError:
        return _choice == "exp"
    fn.restype = C.c_char_p
    return b"+experiment" in fn()


EXPERIMENT = _is_experiment_build()

BP_OK = 0
ERRORS = {
    -1: "BP_ERR_INVALID_ARG", -2: "BP_ERR_NOT_POW2", -3: "BP_ERR_BAD_POINT", -4: "BP_ERR_BAD_SCALAR",
    -5: "BP_ERR_BASIS", -6: "BP_ERR_LENGTH", -7: "BP_ERR_DIV_ZERO", -8: "BP_ERR_NO_DEVICE", -9: "BP_ERR_HIP",
    -10: "BP_ERR_TOO_LARGE", -11: "BP_ERR_ASSERT", -12: "BP_ERR_COMM",
}
FR_BYTES_LE, FR_MONT = 0, 1
MSM_BLOB_BYTES = 22592          # BP_MSM_BLOB_BYTES
COMM_ID_BYTES = 128             # BP_COMM_ID_BYTES (RCCL's ncclUniqueId)
BASIS_LAGRANGE, BASIS_MONOMIAL = 0, 1

_vp, _sz, _u64, _u32, _int, _cp = C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_int, C.c_char_p
_pp = C.POINTER

# name -> (restype, argtypes); mirrors include/bp_msm_ntt.h one to one
SIGNATURES = {
    "bp_init": (_int, [_pp(_vp), _int]),
    "bp_init_multi": (_int, [_pp(_vp), _pp(_int), _int]),
    "bp_ctx_devices": (_int, [_vp, _pp(_int), _int]),
    "bp_destroy": (None, [_vp]),
    "bp_last_error": (_cp, [_vp]),
    "bp_version": (_cp, []),
    "bp_set_stream": (_int, [_vp, _vp]),
    "bp_synchronize": (_int, [_vp]),
    "bp_srs_load": (_int, [_vp, _vp, _sz, _pp(_u64)]),
    "bp_srs_load_projective144": (_int, [_vp, _vp, _sz, _pp(_u64)]),
    "bp_srs_generate": (_int, [_vp, _sz, _vp, _pp(_u64)]),
    "bp_srs_generate_progression": (_int, [_vp, _sz, _vp, _vp, _pp(_u64)]),
    "bp_srs_len": (_int, [_vp, _u64, _pp(_sz)]),
    "bp_srs_export": (_int, [_vp, _u64, _sz, _sz, _vp]),
    "bp_srs_export_projective144": (_int, [_vp, _u64, _sz, _sz, _vp]),
    "bp_srs_free": (_int, [_vp, _u64]),
    "bp_srs_precompute": (_int, [_vp, _u64, _u32]),
    "bp_srs_table_info": (_int, [_vp, _u64, _pp(_u32), _pp(_u32), _pp(_u64)]),
    "bp_msm_last_used_tables": (_int, [_vp]),
    "bp_g1_bytes96_to_compressed48": (_int, [_vp, _vp]),
    "bp_circuit_load": (_int, [_vp, _u32, _vp, _int, _int, _pp(_u64)]),
    "bp_circuit_free": (_int, [_vp, _u64]),
    "bp_circuit_commitments": (_int, [_vp, _u64, _u64, _vp]),
    "bp_make_s_polynomials": (_int, [_u32, _vp, _vp, _vp, _vp]),
    "bp_prove": (_int, [_vp, _u64, _u64, _vp, _vp, _vp, _vp, _int, _int, _vp, _vp]),
    "bp_prove_last_stats": (_int, [_vp, _vp, _pp(C.c_float)]),
    "bp_transcript_test_vector": (_int, [_vp]),
    "bp_msm_g1": (_int, [_vp, _u64, _vp, _sz, _int, _vp]),
    "bp_msm_g1_projective144": (_int, [_vp, _vp, _sz, _vp, _sz, _int, _vp]),
    "bp_msm_g1_partial": (_int, [_vp, _u64, _sz, _vp, _sz, _int, _int, _vp]),
    "bp_msm_g1_blob_device": (_int, [_vp, _u64, _sz, _vp, _sz, _int, _int, _vp]),
    "bp_msm_g1_blob_device_async": (_int, [_vp, _u64, _sz, _vp, _sz, _int, _int, _vp]),
    "bp_msm_blobs_sum_device_async": (_int, [_vp, _vp, _sz, _vp]),
    "bp_msm_blobs_sum_device": (_int, [_vp, _vp, _sz, _vp]),
    "bp_msm_blobs_combine": (_int, [_vp, _sz, _vp]),
    "bp_comm_unique_id": (_int, [_vp]),
    "bp_comm_init_rank": (_int, [_vp, _vp, _int, _int]),
    "bp_comm_info": (_int, [_vp, _pp(_int), _pp(_int)]),
    "bp_comm_set_timeout_ms": (_int, [_vp, _u32]),
    "bp_comm_stats": (_int, [_vp, _pp(_u64), _pp(_u32)]),
    "bp_comm_destroy": (_int, [_vp]),
    "bp_msm_g1_allgather": (_int, [_vp, _u64, _sz, _vp, _sz, _int, _int, _vp]),
    "bp_comm_last_exchange_ms": (_int, [_vp, _pp(C.c_float)]),
    "bp_ntt_columns_allgather": (_int, [_vp, _vp, _u32, _sz]),
    "bp_g1_sum_partials": (_int, [_vp, _sz, _vp]),
    "bp_g1_partial_to_bytes96": (_int, [_vp, _vp]),
    "bp_g1_bytes96_to_partial": (_int, [_vp, _vp]),
    "bp_msm_last_stats": (_int, [_vp, _pp(C.c_float), _pp(C.c_float), _pp(_u64), _pp(_u32)]),
    "bp_msm_last_member_stats": (_int, [_vp, _int, _pp(C.c_float), _pp(C.c_float), _pp(C.c_float), _pp(_u64)]),
    "bp_ntt_fr": (_int, [_vp, _vp, _u32, _int, _int, _sz, _sz]),
    "bp_ntt_fr_device": (_int, [_vp, _vp, _u32, _int, _sz, _sz]),
    "bp_ntt_fr_device_async": (_int, [_vp, _vp, _u32, _int, _sz, _sz]),
    "bp_ntt_last_stats": (_int, [_vp, _pp(C.c_float), _pp(_u32)]),
    "bp_ntt_last_members": (_int, [_vp]),
    "bp_fr_convert": (_int, [_vp, _sz, _int, _int, _vp]),
    "bp_msm_window_scalars": (_int, [_vp, _sz, _int, _sz, _sz, _vp]),
    "bp_fr_synthetic_device": (_int, [_vp, _vp, _sz, _u64]),
    "bp_root_of_unity": (_int, [_u64, _int, _vp]),
    "bp_roots_of_unity": (_int, [_vp, _u64, _int, _vp]),
    "bp_poly_evaluate": (_int, [_vp, _vp, _sz, _int, _vp, _int, _vp]),
    "bp_poly_add": (_int, [_vp, _vp, _sz, _vp, _sz, _int, _int, _vp, _pp(_sz)]),
    "bp_poly_sub": (_int, [_vp, _vp, _sz, _vp, _sz, _int, _int, _vp, _pp(_sz)]),
    "bp_poly_scalar_op": (_int, [_vp, _vp, _sz, _int, _vp, _int, _int, _vp]),
    "bp_poly_mul": (_int, [_vp, _vp, _sz, _vp, _sz, _int, _int, _vp, _pp(_sz)]),
    "bp_poly_div": (_int, [_vp, _vp, _sz, _vp, _sz, _int, _int, _vp, _pp(_sz)]),
    "bp_grand_product": (_int, [_vp] + [_vp] * 6 + [_sz] + [_vp] * 4 + [_int, _vp]),
    "bp_commit": (_int, [_vp, _u64, _vp, _sz, _int, _int, _vp]),
    "bp_poly_add_device": (_int, [_vp, _vp, _sz, _vp, _sz, _int, _vp, _pp(_sz)]),
    "bp_poly_sub_device": (_int, [_vp, _vp, _sz, _vp, _sz, _int, _vp, _pp(_sz)]),
    "bp_poly_scalar_op_device": (_int, [_vp, _vp, _sz, _int, _vp, _int, _vp]),
    "bp_poly_mul_device": (_int, [_vp, _vp, _sz, _vp, _sz, _int, _vp, _pp(_sz)]),
    "bp_poly_div_device": (_int, [_vp, _vp, _sz, _vp, _sz, _int, _vp, _pp(_sz)]),
    "bp_poly_evaluate_device": (_int, [_vp, _vp, _sz, _int, _vp, _vp]),
    "bp_poly_scale_powers_device": (_int, [_vp, _vp, _sz, _vp, _vp]),
    "bp_roots_of_unity_device": (_int, [_vp, _u64, _vp]),
    "bp_grand_product_device": (_int, [_vp] + [_vp] * 6 + [_sz] + [_vp] * 4 + [_vp]),
    "bp_commit_device": (_int, [_vp, _u64, _vp, _sz, _int, _vp]),
    "bp_commit_many_device": (_int, [_vp, _u64, _vp, _vp, _sz, _int, _vp]),
}

_lib = None


class BpError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__("%s failed: %s (%d)%s" % (where, ERRORS.get(code, "?"), code, (": " + detail) if detail else ""))


def load():
    """dlopen the HIP library; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            "libbp_msm_ntt.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C baby_plonk_rust_amd/csrc`. There is no CPU fallback." % SO_PATH)
    lib = C.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library drift
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib
