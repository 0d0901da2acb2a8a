"""baby_plonk_rust_amd -- MI355X-native MSM + Fr-NTT hot path of ChainUpZero/baby-plonk-rust.

Only what the path needs: csrc/ (HIP kernels + C ABI), this host-side mirror of the reference interface
(api.py), the multi-GPU sharding helpers (dist.py) and a C++ mirror (host/).  No CPU fallback exists."""
from ._lib import BASIS_LAGRANGE, BASIS_MONOMIAL, FR_BYTES_LE, FR_MONT, BpError, load  # noqa: F401
from .api import (SRS_TABLES_OFF, BucketMSM, Circuit, Prover, make_s_polynomials, transcript_test_vector, Context, DevicePolynomial, Polynomial, commit_device, commit_many_device, roots_of_unity_device, round_2_z_device, Setup, bytes96_to_partial, default_context, i_ntt_381,  # noqa: F401
                  ntt_381, root_of_unity, roots_of_unity, round_2_z, scalar_from_int, scalar_to_int, scalars_from_ints,
                  scalars_to_ints, sum_partials, combine_blobs)
