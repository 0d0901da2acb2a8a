// Generated from include/bp_msm_ntt.h by tools/gen_rust_bindings.py -- do not edit; the header's comments are the documentation.
// Drop into the reference crate as src/gpu.rs (`pub mod gpu;` in src/lib.rs) and link with `-l bp_msm_ntt`.
#![allow(dead_code)]
use std::os::raw::{c_char, c_int, c_void};
#[repr(C)]
pub struct BpCtx {
    _private: [u8; 0],
}
pub const BP_SRS_TABLES_OFF: u32 = 1;
pub const BP_MSM_BLOB_BYTES: usize = 22592;
pub const BP_OK: c_int = 0;
pub const BP_ERR_INVALID_ARG: c_int = -1;
pub const BP_ERR_NOT_POW2: c_int = -2;
pub const BP_ERR_BAD_POINT: c_int = -3;
pub const BP_ERR_BAD_SCALAR: c_int = -4;
pub const BP_ERR_BASIS: c_int = -5;
pub const BP_ERR_LENGTH: c_int = -6;
pub const BP_ERR_DIV_ZERO: c_int = -7;
pub const BP_ERR_NO_DEVICE: c_int = -8;
pub const BP_ERR_HIP: c_int = -9;
pub const BP_ERR_TOO_LARGE: c_int = -10;
pub const BP_ERR_ASSERT: c_int = -11;
pub const BP_FR_BYTES_LE: c_int = 0;
pub const BP_FR_MONT: c_int = 1;
pub const BP_BASIS_LAGRANGE: c_int = 0;
pub const BP_BASIS_MONOMIAL: c_int = 1;
extern "C" {
    pub fn bp_init(out: *mut *mut BpCtx, device_id: c_int) -> c_int;
    pub fn bp_init_multi(out: *mut *mut BpCtx, device_ids: *const c_int, n_devices: c_int) -> c_int;
    pub fn bp_ctx_devices(ctx: *mut BpCtx, device_ids: *mut c_int, cap: c_int) -> c_int;
    pub fn bp_destroy(ctx: *mut BpCtx);
    pub fn bp_last_error(ctx: *mut BpCtx) -> *const c_char;
    pub fn bp_version() -> *const c_char;
    pub fn bp_set_stream(ctx: *mut BpCtx, hip_stream: *mut c_void) -> c_int;
    pub fn bp_synchronize(ctx: *mut BpCtx) -> c_int;
    pub fn bp_srs_load(ctx: *mut BpCtx, points96: *const u8, n: usize, srs_handle: *mut u64) -> c_int;
    pub fn bp_srs_load_projective144(ctx: *mut BpCtx, points144: *const u8, n: usize, srs_handle: *mut u64) -> c_int;
    pub fn bp_msm_g1_projective144(ctx: *mut BpCtx, points144: *const u8, n_points: usize, scalars: *const c_void, n_scalars: usize, scalar_fmt: c_int, out96: *mut u8) -> c_int;
    pub fn bp_srs_generate(ctx: *mut BpCtx, powers: usize, tau32: *const u8, srs_handle: *mut u64) -> c_int;
    pub fn bp_srs_generate_progression(ctx: *mut BpCtx, n: usize, a32: *const u8, d32: *const u8, srs_handle: *mut u64) -> c_int;
    pub fn bp_srs_len(ctx: *mut BpCtx, srs_handle: u64, n: *mut usize) -> c_int;
    pub fn bp_srs_export(ctx: *mut BpCtx, srs_handle: u64, first: usize, n: usize, points96: *mut u8) -> c_int;
    pub fn bp_srs_export_projective144(ctx: *mut BpCtx, srs_handle: u64, first: usize, n: usize, points144: *mut u8) -> c_int;
    pub fn bp_srs_free(ctx: *mut BpCtx, srs_handle: u64) -> c_int;
    pub fn bp_srs_precompute(ctx: *mut BpCtx, srs_handle: u64, window_bits: u32) -> c_int;
    pub fn bp_srs_table_info(ctx: *mut BpCtx, srs_handle: u64, window_bits: *mut u32, windows: *mut u32, bytes: *mut u64) -> c_int;
    pub fn bp_msm_g1(ctx: *mut BpCtx, srs_handle: u64, scalars: *const c_void, n_scalars: usize, scalar_fmt: c_int, out96: *mut u8) -> c_int;
    pub fn bp_msm_g1_partial(ctx: *mut BpCtx, srs_handle: u64, first: usize, scalars: *const c_void, n_scalars: usize, scalar_fmt: c_int, scalars_on_device: c_int, out144: *mut u8) -> c_int;
    pub fn bp_msm_g1_blob_device(ctx: *mut BpCtx, srs_handle: u64, first: usize, scalars: *const c_void, n_scalars: usize, scalar_fmt: c_int, scalars_on_device: c_int, d_blob: *mut c_void) -> c_int;
    pub fn bp_msm_g1_blob_device_async(ctx: *mut BpCtx, srs_handle: u64, first: usize, scalars: *const c_void, n_scalars: usize, scalar_fmt: c_int, scalars_on_device: c_int, d_blob: *mut c_void) -> c_int;
    pub fn bp_msm_blobs_sum_device(ctx: *mut BpCtx, d_blobs: *const c_void, n_blobs: usize, d_out_blob: *mut c_void) -> c_int;
    pub fn bp_msm_blobs_sum_device_async(ctx: *mut BpCtx, d_blobs: *const c_void, n_blobs: usize, d_out_blob: *mut c_void) -> c_int;
    pub fn bp_msm_blobs_combine(blobs: *const c_void, n_blobs: usize, out96: *mut u8) -> c_int;
    pub fn bp_msm_window_scalars(scalars: *const c_void, n: usize, scalar_fmt: c_int, b: usize, c: usize, out_le32: *mut c_void) -> c_int;
    pub fn bp_g1_sum_partials(partials144: *const u8, n: usize, out96: *mut u8) -> c_int;
    pub fn bp_g1_partial_to_bytes96(in144: *const u8, out96: *mut u8) -> c_int;
    pub fn bp_g1_bytes96_to_partial(in96: *const u8, out144: *mut u8) -> c_int;
    pub fn bp_g1_bytes96_to_compressed48(in96: *const u8, out48: *mut u8) -> c_int;
    pub fn bp_msm_last_stats(ctx: *mut BpCtx, accumulate_ms: *mut f32, total_device_ms: *mut f32, mixed_adds: *mut u64, window_bits: *mut u32) -> c_int;
    pub fn bp_msm_last_member_stats(ctx: *mut BpCtx, member: c_int, upload_ms: *mut f32, accumulate_ms: *mut f32, total_device_ms: *mut f32, mixed_adds: *mut u64) -> c_int;
    pub fn bp_msm_last_used_tables(ctx: *mut BpCtx) -> c_int;
    pub fn bp_ntt_fr(ctx: *mut BpCtx, data: *mut c_void, log_n: u32, inverse: c_int, scalar_fmt: c_int, batch: usize, stride: usize) -> c_int;
    pub fn bp_ntt_fr_device(ctx: *mut BpCtx, d_data: *mut c_void, log_n: u32, inverse: c_int, batch: usize, stride: usize) -> c_int;
    pub fn bp_ntt_fr_device_async(ctx: *mut BpCtx, d_data: *mut c_void, log_n: u32, inverse: c_int, batch: usize, stride: usize) -> c_int;
    pub fn bp_ntt_last_stats(ctx: *mut BpCtx, device_ms: *mut f32, passes: *mut u32) -> c_int;
    pub fn bp_ntt_last_members(ctx: *mut BpCtx) -> c_int;
    pub fn bp_root_of_unity(group_order: u64, scalar_fmt: c_int, out32: *mut u8) -> c_int;
    pub fn bp_roots_of_unity(ctx: *mut BpCtx, group_order: u64, scalar_fmt: c_int, out: *mut c_void) -> c_int;
    pub fn bp_fr_convert(in_: *const c_void, n: usize, from_fmt: c_int, to_fmt: c_int, out: *mut c_void) -> c_int;
    pub fn bp_fr_synthetic_device(ctx: *mut BpCtx, d_out: *mut c_void, n: usize, seed: u64) -> c_int;
    pub fn bp_poly_evaluate(ctx: *mut BpCtx, coeffs: *const c_void, n: usize, basis: c_int, x32: *const c_void, scalar_fmt: c_int, out32: *mut c_void) -> c_int;
    pub fn bp_poly_add(ctx: *mut BpCtx, a: *const c_void, na: usize, b: *const c_void, nb: usize, basis: c_int, scalar_fmt: c_int, out: *mut c_void, n_out: *mut usize) -> c_int;
    pub fn bp_poly_sub(ctx: *mut BpCtx, a: *const c_void, na: usize, b: *const c_void, nb: usize, basis: c_int, scalar_fmt: c_int, out: *mut c_void, n_out: *mut usize) -> c_int;
    pub fn bp_poly_scalar_op(ctx: *mut BpCtx, a: *const c_void, n: usize, basis: c_int, s32: *const c_void, op: c_int, scalar_fmt: c_int, out: *mut c_void) -> c_int;
    pub fn bp_poly_mul(ctx: *mut BpCtx, a: *const c_void, na: usize, b: *const c_void, nb: usize, basis: c_int, scalar_fmt: c_int, out: *mut c_void, n_out: *mut usize) -> c_int;
    pub fn bp_poly_div(ctx: *mut BpCtx, a: *const c_void, na: usize, b: *const c_void, nb: usize, basis: c_int, scalar_fmt: c_int, out: *mut c_void, n_out: *mut usize) -> c_int;
    pub fn bp_grand_product(ctx: *mut BpCtx, a: *const c_void, b: *const c_void, c: *const c_void, s1: *const c_void, s2: *const c_void, s3: *const c_void, n: usize, beta32: *const c_void, gamma32: *const c_void, k1_32: *const c_void, k2_32: *const c_void, scalar_fmt: c_int, z_out: *mut c_void) -> c_int;
    pub fn bp_poly_add_device(ctx: *mut BpCtx, d_a: *const c_void, na: usize, d_b: *const c_void, nb: usize, basis: c_int, d_out: *mut c_void, n_out: *mut usize) -> c_int;
    pub fn bp_poly_sub_device(ctx: *mut BpCtx, d_a: *const c_void, na: usize, d_b: *const c_void, nb: usize, basis: c_int, d_out: *mut c_void, n_out: *mut usize) -> c_int;
    pub fn bp_poly_scalar_op_device(ctx: *mut BpCtx, d_a: *const c_void, n: usize, basis: c_int, s32_mont: *const c_void, op: c_int, d_out: *mut c_void) -> c_int;
    pub fn bp_poly_mul_device(ctx: *mut BpCtx, d_a: *const c_void, na: usize, d_b: *const c_void, nb: usize, basis: c_int, d_out: *mut c_void, n_out: *mut usize) -> c_int;
    pub fn bp_poly_div_device(ctx: *mut BpCtx, d_a: *const c_void, na: usize, d_b: *const c_void, nb: usize, basis: c_int, d_out: *mut c_void, n_out: *mut usize) -> c_int;
    pub fn bp_poly_evaluate_device(ctx: *mut BpCtx, d_coeffs: *const c_void, n: usize, basis: c_int, x32_mont: *const c_void, out32_mont: *mut c_void) -> c_int;
    pub fn bp_poly_scale_powers_device(ctx: *mut BpCtx, d_a: *const c_void, n: usize, w32_mont: *const c_void, d_out: *mut c_void) -> c_int;
    pub fn bp_roots_of_unity_device(ctx: *mut BpCtx, group_order: u64, d_out: *mut c_void) -> c_int;
    pub fn bp_grand_product_device(ctx: *mut BpCtx, d_a: *const c_void, d_b: *const c_void, d_c: *const c_void, d_s1: *const c_void, d_s2: *const c_void, d_s3: *const c_void, n: usize, beta32: *const c_void, gamma32: *const c_void, k1_32: *const c_void, k2_32: *const c_void, d_z: *mut c_void) -> c_int;
    pub fn bp_commit_device(ctx: *mut BpCtx, srs_handle: u64, d_coeffs: *const c_void, n: usize, basis: c_int, out96: *mut u8) -> c_int;
    pub fn bp_commit_many_device(ctx: *mut BpCtx, srs_handle: u64, d_coeffs: *mut *const c_void, n: *const usize, count: usize, basis: c_int, out96: *mut u8) -> c_int;
    pub fn bp_commit(ctx: *mut BpCtx, srs_handle: u64, coeffs: *const c_void, n: usize, basis: c_int, scalar_fmt: c_int, out96: *mut u8) -> c_int;
    pub fn bp_circuit_load(ctx: *mut BpCtx, log_n: u32, columns: *mut *const c_void, scalar_fmt: c_int, columns_on_device: c_int, circuit_handle: *mut u64) -> c_int;
    pub fn bp_circuit_free(ctx: *mut BpCtx, circuit_handle: u64) -> c_int;
    pub fn bp_circuit_commitments(ctx: *mut BpCtx, srs_handle: u64, circuit_handle: u64, out768: *mut u8) -> c_int;
    pub fn bp_make_s_polynomials(log_n: u32, wire_ids: *const u32, s1: *mut c_void, s2: *mut c_void, s3: *mut c_void) -> c_int;
    pub fn bp_prove(ctx: *mut BpCtx, srs_handle: u64, circuit_handle: u64, a: *const c_void, b: *const c_void, c: *const c_void, public_input: *const c_void, scalar_fmt: c_int, witness_on_device: c_int, blinders: *const u8, proof: *mut u8) -> c_int;
    pub fn bp_prove_last_stats(ctx: *mut BpCtx, round_ms: *mut f32, total_ms: *mut f32) -> c_int;
    pub fn bp_transcript_test_vector(out32: *mut u8) -> c_int;
}
