/*
 * bp_msm_ntt.h -- C ABI of the MI355X-native PLONK hot path (BLS12-381 G1 MSM + Fr NTT + polynomial ops).
 *
 * This is the drop-in boundary.  The reference (ChainUpZero/baby-plonk-rust) has no FFI of its own; the
 * entry points below are what a Rust `extern "C"` block in src/msm.rs / src/utils.rs / src/polynomial.rs /
 * src/setup.rs would bind (INTEGRATION.md shows those stubs).  Each entry names the reference interface it
 * replaces (paths relative to the reference repository root).
 *
 * Conventions
 *   - plain pointers and sizes only; no ownership transfer; every call returns BP_OK (0) or a negative code
 *     and never throws.  Where the reference panics (assert!, unwrap), the call returns an error instead and
 *     the Rust shim turns it back into a panic.
 *   - a bp_ctx is driven by one host thread (the reference is single-threaded); it owns its HIP stream(s),
 *     workspaces, SRS and circuit handles, so several contexts (one host thread each) may share a GPU and overlap
 *     their work.  Every entry point restores the calling thread's current HIP device before it returns.
 *   - multi-GPU, two ways:
 *       in one process   bp_init_multi(device_ids, n): ONE context drives n GPUs.  An SRS loaded or generated through it is
 *                        sharded by contiguous point range over the GPUs; bp_msm_g1 / bp_commit / bp_prove split the scalars
 *                        the same way, run the shards concurrently and add the partial sums -- the single-threaded Rust caller
 *                        of Setup::commit (src/setup.rs:32-37) uses all GPUs without knowing about them.  Batched host NTTs
 *                        (bp_ntt_fr with batch > 1) spread their independent columns over the GPUs.  Every GPU beyond the
 *                        first is driven by a persistent host thread of the library (uploads, waits, epilogues in parallel).
 *                        A list that names a device twice ({0,0}: what a one-GPU box can rehearse) makes a REHEARSAL group: its
 *                        members beyond the first take the GPU-to-GPU branches (hipMemcpyPeerAsync behind the leader's event)
 *                        exactly as members on other cards do.  STATUS: that is the only way those branches have run so far;
 *                        no run on distinct GPUs is recorded yet.
 *   - the library reads no environment variable (experiment knobs exist in the separate -DBP_EXPERIMENT build only).
 *       one process per GPU (torch.distributed / MPI launchers): every rank owns a point range in a plain bp_init context and
 *                        joins the library's own communicator (bp_comm_unique_id on rank 0, the host carries 128 bytes to the
 *                        others, bp_comm_init_rank everywhere).  bp_msm_g1_allgather then is the whole exchange under this ABI:
 *                        the rank's partial sums stay in HBM as a record, ONE ncclAllGather of the records over xGMI on the
 *                        context's stream, the slot-wise sum of the gathered records on the device, ONE 22-KB device-to-host
 *                        copy, host Horner; bp_ntt_columns_allgather gathers finished NTT columns in place.  A rank never skips
 *                        a collective (a local failure travels as a poisoned record / agreement word and every rank returns it),
 *                        and every wait behind a collective is bounded (bp_comm_set_timeout_ms): errors, not stalls.
 *                        (bp_msm_g1_blob_device + bp_msm_blobs_combine remain for hosts that bring their own collective.)
 *   - wire formats are the reference's own:
 *       scalar  fmt BP_FR_BYTES_LE : 32-byte little-endian canonical  (Scalar::to_bytes,  scalar.rs:292-304)
 *               fmt BP_FR_MONT     : 4 x u64 Montgomery limbs          (Scalar::to_array,  scalar.rs:35-40)
 *       point   96-byte uncompressed affine, x||y big-endian, bit 6 of byte 0 = infinity
 *                                                                      (G1Affine::to_uncompressed, g1.rs:246-260)
 *       projective partial: 144 bytes = x|y|z, 6 x u64 Montgomery limbs each = the memory image of
 *               G1Projective (g1.rs:442-446); only used between our own ranks.
 *   - *_device variants take pointers to HBM (hipMalloc / torch CUDA tensors) and leave results there.
 */
#ifndef BP_MSM_NTT_H
#define BP_MSM_NTT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bp_ctx bp_ctx;

enum {
  BP_OK = 0,
  BP_ERR_INVALID_ARG = -1,   /* null pointer, bad format flag, bad handle */
  BP_ERR_NOT_POW2 = -2,      /* utils.rs:65,108   assert!(is_power_of_two(n)) */
  BP_ERR_BAD_POINT = -3,     /* non-canonical / off-curve / bad flags (g1.rs:273-322) */
  BP_ERR_BAD_SCALAR = -4,    /* 32-byte value >= q (scalar.rs:264-288) */
  BP_ERR_BASIS = -5,         /* polynomial.rs:35,48,53,319, setup.rs:34   assert_eq!(basis, ..) */
  BP_ERR_LENGTH = -6,        /* polynomial.rs:85-89,142-146   Lagrange operands of different length */
  BP_ERR_DIV_ZERO = -7,      /* polynomial.rs:347-348   division by the zero polynomial (unwrap on None) */
  BP_ERR_NO_DEVICE = -8,     /* no usable GPU / HIP runtime error at init */
  BP_ERR_HIP = -9,           /* HIP runtime error; bp_last_error() has the text */
  BP_ERR_TOO_LARGE = -10,    /* size beyond the supported range (NTT > 2^28, MSM >= 2^31 points) */
  BP_ERR_ASSERT = -11,       /* a protocol assert_eq! of the reference failed (prover.rs:319  z_n == 1) */
  BP_ERR_COMM = -12          /* an RCCL call failed, the communicator reported an asynchronous error, or a collective / ncclCommInitRank
                                did not finish within the bound of bp_comm_set_timeout_ms (the communicator is then aborted); text in bp_last_error */
};
enum { BP_FR_BYTES_LE = 0, BP_FR_MONT = 1 };
enum { BP_BASIS_LAGRANGE = 0, BP_BASIS_MONOMIAL = 1 };   /* polynomial.rs:8-11 */

/* ---- context ------------------------------------------------------------------------------------- */
int  bp_init(bp_ctx** out, int device_id);
/* One context over n_devices GPUs (`bp_init(out, device_ids, n_devices)`); device_ids[0] is the primary
 * device: polynomial operators, the prover's rounds and *_device pointers live there, MSMs and SRS storage span all of
 * them.  A device may be listed more than once (each listing is an independent shard; used to test the path on one GPU).
 * n_devices == 1 is the same as bp_init. */
int  bp_init_multi(bp_ctx** out, const int* device_ids, int n_devices);
/* number of shards of the context (1 for bp_init); device_ids, if not NULL, receives up to cap of their device ids */
int  bp_ctx_devices(bp_ctx* ctx, int* device_ids, int cap);
void bp_destroy(bp_ctx* ctx);
const char* bp_last_error(bp_ctx* ctx);            /* text of the last error on this ctx ("" if none) */
const char* bp_version(void);
/* Run all subsequent work of this ctx on an existing hipStream_t (e.g. torch's current stream). NULL = own stream.
 * Every entry point that reads or writes CALLER device memory (the *_device family) is ordered on that stream: work the caller
 * enqueued on the same stream before the call is seen, work enqueued after sees the call's results -- the blocking forms
 * additionally wait for the stream, the *_async forms do not.  On the context's own stream (the default) the caller must
 * have finished its pending work on those buffers before the call. */
int  bp_set_stream(bp_ctx* ctx, void* hip_stream);
int  bp_synchronize(bp_ctx* ctx);

/* ---- SRS: Setup.powers_of_x (src/setup.rs:7-10), resident in HBM ----------------------------------- */
/* Upload n points in the 96-byte encoding.  Replaces handing &self.powers_of_x to bucket_msm on every
 * commit (setup.rs:36): the SRS is immutable after construction, so it is uploaded once and cached.
 * Checks = G1Affine::from_uncompressed_unchecked + is_on_curve (g1.rs:273-322, 101-106): canonical coordinates, flag
 * bits, curve equation -> BP_ERR_BAD_POINT.  Subgroup membership (is_torsion_free, the extra check of the checked
 * decoder from_uncompressed) is NOT tested: an SRS comes from a trusted setup, as the reference's own Setup does. */
int  bp_srs_load(bp_ctx* ctx, const uint8_t* points96, size_t n, uint64_t* srs_handle);
/* The literal bucket_msm(points: &[G1Projective], ..) seam (src/msm.rs:76-81): n points as the in-memory image of
 * G1Projective (x | y | z, 6 x u64 Montgomery limbs each = 144 bytes, g1.rs:442-446), e.g. the Vec<G1Projective> of
 * Setup::powers_of_x handed over as a byte slice.  Normalised on the GPU (one shared inversion per 8 points, what
 * G1Projective::batch_normalize does on the host, g1.rs:806-839); z = 0 is the identity.  No curve check: the type
 * guarantees it (as G1Affine::from(&G1Projective) assumes). */
int  bp_srs_load_projective144(bp_ctx* ctx, const uint8_t* points144, size_t n, uint64_t* srs_handle);
/* The same seam as ONE call, nothing cached: result = sum_i scalars[i] * points[i] over min(n_points, n_scalars) pairs (the zip of
 * msm.rs:85), both operands in host memory.  Equivalent to bp_srs_load_projective144 + bp_msm_g1 + bp_srs_free, but the operands cross
 * PCIe in two pieces and the multiplication of the first runs while the second is uploaded and normalised, out of workspaces instead
 * of an SRS entry (from 2^17 pairs; below that, and on bp_init_multi contexts, it is the three calls): 8.3 against 8.9 ms at 2^20.  Statistics afterwards (bp_msm_last_stats): additions of all pieces,
 * times of the last piece. */
int  bp_msm_g1_projective144(bp_ctx* ctx, const uint8_t* points144, size_t n_points, const void* scalars, size_t n_scalars, int scalar_fmt,
                             uint8_t out96[96]);
/* Setup::generate_srs(powers, tau) (setup.rs:12-31): P_i = tau^i * G, generated on the GPU. tau: 32-byte LE. */
int  bp_srs_generate(bp_ctx* ctx, size_t powers, const uint8_t tau32[32], uint64_t* srs_handle);
/* Synthetic benchmark points P_i = (a + i*d) * G, generated on the GPU. */
int  bp_srs_generate_progression(bp_ctx* ctx, size_t n, const uint8_t a32[32], const uint8_t d32[32], uint64_t* srs_handle);
int  bp_srs_len(bp_ctx* ctx, uint64_t srs_handle, size_t* n);
/* Read points [first, first+n) back in the 96-byte encoding (G1Affine::to_uncompressed). */
int  bp_srs_export(bp_ctx* ctx, uint64_t srs_handle, size_t first, size_t n, uint8_t* points96);
/* The same points as G1Projective memory images (z = 1, the identity as (0 : 1 : 0)): G1Projective::from(&G1Affine), g1.rs:176-190. */
int  bp_srs_export_projective144(bp_ctx* ctx, uint64_t srs_handle, size_t first, size_t n, uint8_t* points144);
int  bp_srs_free(bp_ctx* ctx, uint64_t srs_handle);
/* Fixed-base window tables for an SRS that serves many commitments (Setup lives as long as the prover,
 * src/setup.rs:7-10): T[w][i] = 2^(window_bits * w) * P_i for every window w, affine, resident in HBM
 * (windows x srs_len x 128 bytes -- a 112-byte point per 128-byte cache line: 1.7 GB at 2^20 points, 28 GB at 2^24).  With tables every window of an
 * MSM feeds one shared bucket set, so the bucket reduction and the Horner epilogue shrink from `windows`
 * passes to one; results are the same group element.  From 21-bit windows the rows are R^w * P_i for a digit radix R that need not
 * be a power of two (12 windows of 22 bits cover 264 bits for 255-bit scalars: R = 0x288000 fills 1.33 M buckets instead of 2.1 M;
 * csrc/msm_digits.hpp) -- invisible at this boundary, same bytes out.  window_bits: 0 = chosen from srs_len (16 below 2^20 points, 20 -- thirteen
 * windows -- from 2^20 points, 22 from 2^24), BP_SRS_TABLES_OFF = drop the tables, else 4..24 (above 16 the bucket sort is partitioned).  MSMs shorter than 2^window_bits / 8 scalars keep
 * using the table-free path.  The reference has no counterpart (its MSM recomputes from the raw points).
 * Memory budget (hipMemGetInfo per device; the bytes of the tables this SRS holds now count as free): window_bits = 0 never fails
 * for lack of memory -- a width that does not fit beside what else lives on the device falls back to wider windows (fewer rows; only
 * widths an MSM over this SRS would use, 8 * srs_len >= 2^width) and then to no tables at all (bp_srs_table_info reports what was
 * built; results are the same bytes either way); an explicit width that does not fit returns BP_ERR_TOO_LARGE with the sizes in
 * bp_last_error() before anything is released or allocated: the SRS keeps the tables it had.
 * window_bits = 256 + w (w = 6..22: tables of EVERY bit position, T[p][i] = 2^p * P_i for p < 256, used with the scalars' width-w
 * non-adjacent form) exists ONLY in the experiment build libbp_msm_ntt_exp.so (measured slower twice, docs/EXPERIMENTS.md C); the
 * shipped library rejects it with BP_ERR_INVALID_ARG. */
#define BP_SRS_TABLES_OFF 1u
int  bp_srs_precompute(bp_ctx* ctx, uint64_t srs_handle, uint32_t window_bits);
/* window_bits / windows / bytes of the tables of an SRS (all 0 without tables). */
int  bp_srs_table_info(bp_ctx* ctx, uint64_t srs_handle, uint32_t* window_bits, uint32_t* windows, uint64_t* bytes);

/* ---- MSM: BucketMSM::bucket_msm(points, scalars, b, c) (src/msm.rs:76-118) ----------------------- */
/* sum_{i < min(n_scalars, srs_len - first)} s_i * P_{first+i}   (zip truncation of msm.rs:29).
 * The window parameters (b, c) of the reference are not taken: for b = 256 and c dividing 256 -- the only configuration
 * the reference calls it with (setup.rs:36: b = 256, c = 4) -- they do not change the group element.  (For other values
 * msm.rs:83,119-139 drops the low 256 - c*floor(b/c) bits of every scalar; that is not reproduced, and the mirrors of
 * BucketMSM::bucket_msm reject such (b, c).)
 * out96: affine result in the 96-byte encoding (identity = 0x40 then zeros). */
int  bp_msm_g1(bp_ctx* ctx, uint64_t srs_handle, const void* scalars, size_t n_scalars, int scalar_fmt,
               uint8_t out96[96]);
/* Same sum over the SRS slice starting at `first`, result as a 144-byte projective partial (multi-GPU:
 * each rank owns a point range; partials are all-gathered and added).  scalars_on_device != 0: `scalars`
 * points to HBM. */
int  bp_msm_g1_partial(bp_ctx* ctx, uint64_t srs_handle, size_t first, const void* scalars, size_t n_scalars,
                       int scalar_fmt, int scalars_on_device, uint8_t out144[144]);
/* One process per GPU: the same sum as bp_msm_g1_partial, left in HBM as an opaque BP_MSM_BLOB_BYTES record at d_blob (the
 * per-window / per-bit-plane partial sums of this rank, not yet combined).  The caller all-gathers the records of all
 * ranks on the device (RCCL) and hands the gathered host copy to bp_msm_blobs_combine: one collective and one
 * device-to-host copy per MSM, no host hop before the exchange.  The record is complete when the call returns (it is
 * written on the context's own stream, which the call waits for); work the caller still has pending on d_blob on another
 * stream -- e.g. the zero fill of a freshly allocated torch tensor -- must have finished before the call. */
#define BP_MSM_BLOB_BYTES 22592u      /* 64-byte header + 128 accumulator slots of 176 bytes */
int  bp_msm_g1_blob_device(bp_ctx* ctx, uint64_t srs_handle, size_t first, const void* scalars, size_t n_scalars, int scalar_fmt,
                           int scalars_on_device, void* d_blob);
/* The same, STREAM-ORDERED: enqueued on the context's stream and not waited for.  With bp_set_stream(the caller's stream) the
 * record is written in that stream's order: behind whatever the caller enqueued on d_blob or the scalars before (a zero fill,
 * a producer kernel), in front of whatever it enqueues next (the RCCL all-gather of dist.ShardedMsm) -- no host wait, and
 * ordering is a property of the call, not of prose.  Errors that need the result (a canonical-bytes scalar >= q) travel in
 * the record's header and surface in bp_msm_blobs_combine.  bp_msm_last_stats reads the timing once the stream has passed it.
 * Host scalars (scalars_on_device == 0) are copied by a stream-ordered hipMemcpyAsync: from pageable memory the runtime has
 * staged them when the call returns, from PINNED memory (hipHostMalloc, a torch pinned tensor) the DMA reads them later --
 * the buffer must stay valid and unchanged until the stream has passed the call (bp_synchronize, or an event of the caller's). */
int  bp_msm_g1_blob_device_async(bp_ctx* ctx, uint64_t srs_handle, size_t first, const void* scalars, size_t n_scalars, int scalar_fmt,
                                 int scalars_on_device, void* d_blob);
/* Optional device-side pre-sum of the gathered records (n_blobs of them in HBM, BP_MSM_BLOB_BYTES apart): records of equal
 * window layout -- the normal case -- are added slot by slot on the GPU into ONE record at d_out_blob, so only 22 KB cross to
 * the host and the host does one Horner pass.  If the layouts differ the output record is marked invalid (bp_msm_blobs_combine
 * rejects it with BP_ERR_INVALID_ARG) and the caller combines the gathered records on the host instead. */
int  bp_msm_blobs_sum_device(bp_ctx* ctx, const void* d_blobs, size_t n_blobs, void* d_out_blob);
/* The same, stream-ordered (enqueued on the context's stream, not waited for). */
int  bp_msm_blobs_sum_device_async(bp_ctx* ctx, const void* d_blobs, size_t n_blobs, void* d_out_blob);
/* Host-side: combine n_blobs records (host memory, BP_MSM_BLOB_BYTES apart) into the affine result.  Records with the
 * same window layout are added slot by slot before the one Horner pass (msm.rs:107-115). */
int  bp_msm_blobs_combine(const void* blobs, size_t n_blobs, uint8_t out96[96]);
/* Host-side, no GPU: the scalars that BucketMSM::bucket_msm(points, scalars, b, c) effectively multiplies the points by, for ANY
 * (b, c).  The reference walks k = floor(b / c) windows of c bits over the 256-bit big-endian image of the scalar, most
 * significant first (msm.rs:83, 119-139), so it uses the top k*c bits only: result = sum_i (K_i >> (256 - k*c)) * P_i.  For its
 * one call site (b = 256, c = 4, setup.rs:36) and whenever c divides 256 = b the shift is zero.  out_le32 receives n 32-byte
 * little-endian canonical values (feed them to bp_msm_g1 with BP_FR_BYTES_LE).  BP_ERR_INVALID_ARG where the reference panics:
 * c = 0 (division by zero), k = 0 (t_points[0], msm.rs:105), k*c > 256 (bit slice out of range, msm.rs:132), c > 63
 * (the 1 << c bucket vector, msm.rs:24). */
int  bp_msm_window_scalars(const void* scalars, size_t n, int scalar_fmt, size_t b, size_t c, void* out_le32);
/* Host-side: add n projective partials (complete addition, g1.rs:670-712) and normalise (g1.rs:49-63). */
int  bp_g1_sum_partials(const uint8_t* partials144, size_t n, uint8_t out96[96]);
/* Host-side conversions of single points (for the Rust shim's G1Projective <-> bytes plumbing). */
int  bp_g1_partial_to_bytes96(const uint8_t in144[144], uint8_t out96[96]);
int  bp_g1_bytes96_to_partial(const uint8_t in96[96], uint8_t out144[144]);
/* G1Affine::to_compressed (g1.rs:221-244): the 48-byte encoding the transcript absorbs (transcript.rs:66-69) and
 * bp_prove emits; host-side, no GPU. */
int  bp_g1_bytes96_to_compressed48(const uint8_t in96[96], uint8_t out48[48]);
/* HIP-event duration of the bucket-accumulation kernel of the last MSM on this ctx, and its adds. */
int  bp_msm_last_stats(bp_ctx* ctx, float* accumulate_ms, float* total_device_ms, uint64_t* mixed_adds,
                       uint32_t* window_bits);
/* The same per member of a bp_init_multi context (member 0 .. bp_ctx_devices() - 1; a plain context has member 0 only), plus
 * the time of the member's scalar upload when the last MSM took host scalars (0 otherwise): the split of one
 * Setup::commit(&Polynomial) (setup.rs:32-37) over the GPUs into PCIe time and compute time.  Any pointer may be null. */
int  bp_msm_last_member_stats(bp_ctx* ctx, int member, float* upload_ms, float* accumulate_ms, float* total_device_ms,
                              uint64_t* mixed_adds);
/* 1 when the last MSM on this ctx went through fixed-base tables, 0 when not, negative on error. */
int  bp_msm_last_used_tables(bp_ctx* ctx);

/* ---- one process per GPU: the collective under the boundary (SURVEY.md 8b "bp_ctx owns ... the RCCL comm", 8e) ------------------
 * The reference is one thread on one CPU (prover.rs has no notion of ranks); a Rust host that runs one process per GPU calls these
 * instead of bringing its own RCCL binding.  The library is linked against librccl; every collective is enqueued on the context's
 * stream between the library's own kernels.  Communicators belong to plain bp_init contexts (a bp_init_multi context combines its
 * shards itself).  Exercised on the build pool's one-GPU boxes with worlds of ONE rank only (tests/test_gpu_dist.py); no multi-GPU
 * run exists yet, and no number for a world > 1 is claimed.
 * bp_comm_unique_id: rank 0 makes the 128-byte id (ncclGetUniqueId); the host carries it to the other ranks by whatever means it
 * has (a file, a socket, MPI).  bp_comm_init_rank: every rank, same id (ncclCommInitRank; collective -- returns when all `world`
 * ranks have called it; it allocates every buffer the later collectives need and ends with a 32-byte agreement all-gather in
 * which the ranks compare world and record size).  bp_comm_info: rank / world of the context's communicator (world = 0 without one).
 * Failure semantics -- the reference panics, it never blocks (src/setup.rs:34, src/utils.rs:65,108), and neither does a collective here:
 *   - a rank that fails locally in front of a collective still JOINS it and says so inside it (a poisoned MSM record; the agreement
 *     word of bp_ntt_columns_allgather); every rank of the communicator then returns that rank's error code;
 *   - every host wait behind a collective polls the stream and ncclCommGetAsyncError and gives up after the bound of
 *     bp_comm_set_timeout_ms (milliseconds; default 120 000; 0 = wait for ever): the communicator is aborted (ncclCommAbort),
 *     bp_comm_info reports world 0, the call returns BP_ERR_COMM; a new communicator may be created afterwards;
 *   - ncclCommInitRank runs on a helper thread under the same bound; when it expires (a rank never arrived) the call returns
 *     BP_ERR_COMM, the helper stays parked inside RCCL until the process ends and this context refuses further communicators.
 * bp_comm_stats: collectives this context has enqueued so far (agreement all-gathers included) and the bound in force. */
#define BP_COMM_ID_BYTES 128
int  bp_comm_unique_id(uint8_t id[128]);
int  bp_comm_init_rank(bp_ctx* ctx, const uint8_t id[128], int rank, int world);
int  bp_comm_info(bp_ctx* ctx, int* rank, int* world);
int  bp_comm_set_timeout_ms(bp_ctx* ctx, uint32_t ms);
int  bp_comm_stats(bp_ctx* ctx, uint64_t* collectives, uint32_t* timeout_ms);
int  bp_comm_destroy(bp_ctx* ctx);
/* Setup::commit / BucketMSM::bucket_msm over ALL ranks (src/setup.rs:32-37 -> src/msm.rs:76-118): every rank passes the scalars of
 * ITS point range (srs_handle = the rank's shard, `first` inside it, zip truncation as bp_msm_g1_partial) and every rank receives
 * the same 96 bytes: sum over all ranks' (point, scalar) pairs.  Inside: the rank's bit planes stay in HBM as a BP_MSM_BLOB_BYTES
 * record -> ONE ncclAllGather of the records over xGMI -> slot-wise sum of the gathered records on the device -> ONE 22-KB
 * device-to-host copy -> host Horner (msm.rs:107-115) + normalisation.  One host wait per call.  Collective: every rank of the
 * communicator must call it (an empty range is fine: n_scalars = 0).  A rank whose own part fails (unknown handle, `first` beyond
 * its shard, out of memory, a scalar >= q) still takes part: its record carries the error and EVERY rank returns that code. */
int  bp_msm_g1_allgather(bp_ctx* ctx, uint64_t srs_handle, size_t first, const void* scalars, size_t n_scalars, int scalar_fmt,
                         int scalars_on_device, uint8_t out96[96]);
/* GPU time of the last bp_msm_g1_allgather from "this rank's record complete" to "gathered, summed and copied" (HIP events). */
int  bp_comm_last_exchange_ms(bp_ctx* ctx, float* ms);
/* NTT by independent columns (Polynomial::ntt / i_ntt on separate polynomials, src/polynomial.rs:47-55): d_columns holds
 * world x columns_per_rank columns of 2^log_n Montgomery elements in HBM, rank r's finished columns in block r (this rank's block
 * filled by its own bp_ntt_fr_device calls on this context); ONE in-place ncclAllGather leaves every column on every rank.  A 32-byte
 * agreement all-gather goes first (status, log_n, columns_per_rank of every rank): a rank with bad arguments, or ranks that disagree
 * on the shape, make every rank return the same error before any column moves.  log_n <= 28. */
int  bp_ntt_columns_allgather(bp_ctx* ctx, void* d_columns, uint32_t log_n, size_t columns_per_rank);

/* ---- DFT: ntt_381 / i_ntt_381 (src/utils.rs:63-81, 106-129) --------------------------------------- */
/* In-place natural-order transform of length 2^log_n on `batch` vectors, vector b at data + b*stride
 * elements (32 bytes each).  inverse != 0 includes the 1/N scaling (utils.rs:126). */
int  bp_ntt_fr(bp_ctx* ctx, void* data, uint32_t log_n, int inverse, int scalar_fmt, size_t batch, size_t stride);
/* Same on HBM-resident Montgomery data. */
int  bp_ntt_fr_device(bp_ctx* ctx, void* d_data, uint32_t log_n, int inverse, size_t batch, size_t stride);
/* Same, enqueued only: returns once the kernels are on the context's stream (the one given to bp_set_stream, or the context's own);
 * bp_synchronize or any blocking entry point of this context waits for them.  d_data must stay valid until then. */
int  bp_ntt_fr_device_async(bp_ctx* ctx, void* d_data, uint32_t log_n, int inverse, size_t batch, size_t stride);
/* HIP-event duration of all kernels of the last bp_ntt_fr_device / _async call (0 while an enqueued one is still running). */
int  bp_ntt_last_stats(bp_ctx* ctx, float* device_ms, uint32_t* passes);
/* Group contexts (bp_init_multi) and host data: batch > 1 deals the columns round-robin to the members (option i);
 * ONE transform of 2^22 elements or more (BP_NTT_GROUP_SPLIT_FROM) is cut over the members (option ii): each uploads a slice
 * of the columns of the first pass over its own PCIe link, the members swap blocks of the intermediate buffer (peer copies),
 * run the remaining passes on their share and download their outputs.  A member count that is not 2, 4 or 8, or a shape
 * that does not divide, runs on the first device.  Number of members that took part in the last bp_ntt_fr call: */
int  bp_ntt_last_members(bp_ctx* ctx);
/* root_of_unity(n) (utils.rs:39-43) and roots_of_unity(n) (utils.rs:45-52), output in scalar_fmt. */
int  bp_root_of_unity(uint64_t group_order, int scalar_fmt, uint8_t out32[32]);
int  bp_roots_of_unity(bp_ctx* ctx, uint64_t group_order, int scalar_fmt, void* out);

/* Host-side format conversion of n scalars between BP_FR_BYTES_LE and BP_FR_MONT (Scalar::from_bytes / to_bytes,
 * scalar.rs:264-304); returns BP_ERR_BAD_SCALAR for a canonical input >= q.  in == out is allowed. */
int  bp_fr_convert(const void* in, size_t n, int from_fmt, int to_fmt, void* out);

/* Synthetic benchmark scalars written straight into HBM (Montgomery limbs): element i = Scalar::from_bytes_wide
 * (scalar.rs:308-339) of 64 bytes of a SplitMix64 stream. Not in the reference. */
int  bp_fr_synthetic_device(bp_ctx* ctx, void* d_out, size_t n, uint64_t seed);

/* ---- Polynomial (src/polynomial.rs:14-380); values are n x 32-byte scalars in scalar_fmt ---------- */
/* coeffs_evaluate (polynomial.rs:34-45); asserts Monomial basis. */
int  bp_poly_evaluate(bp_ctx* ctx, const void* coeffs, size_t n, int basis, const void* x32, int scalar_fmt,
                      void* out32);
/* impl Add/Sub for Polynomial (polynomial.rs:76-117,134-174). out needs max(na,nb) slots; *n_out = length. */
int  bp_poly_add(bp_ctx* ctx, const void* a, size_t na, const void* b, size_t nb, int basis, int scalar_fmt,
                 void* out, size_t* n_out);
int  bp_poly_sub(bp_ctx* ctx, const void* a, size_t na, const void* b, size_t nb, int basis, int scalar_fmt,
                 void* out, size_t* n_out);
/* impl Add<Scalar>/Sub<Scalar>/Mul<Scalar> (polynomial.rs:57-74,119-132,176-187), including the reference's
 * Lagrange-basis Sub<Scalar> quirk (it adds, :126-128). op: 0 add, 1 sub, 2 mul. */
int  bp_poly_scalar_op(bp_ctx* ctx, const void* a, size_t n, int basis, const void* s32, int op, int scalar_fmt,
                       void* out);
/* impl Mul for Polynomial, Monomial basis (polynomial.rs:240-273): out has na+nb-1 coefficients. */
int  bp_poly_mul(bp_ctx* ctx, const void* a, size_t na, const void* b, size_t nb, int basis, int scalar_fmt,
                 void* out, size_t* n_out);
/* impl Div for Polynomial (polynomial.rs:314-380): quotient only, exactly as the reference produces it
 * (zero quotient coefficients are squeezed out).  out needs na slots. */
int  bp_poly_div(bp_ctx* ctx, const void* a, size_t na, const void* b, size_t nb, int basis, int scalar_fmt,
                 void* out, size_t* n_out);
/* Round-2 permutation grand product (prover.rs:279-319), "next" row f2 of SURVEY.md section 8.  Six Lagrange-basis
 * columns of n values (witness a, b, c and sigma s1, s2, s3), challenges beta, gamma and coset shifts k1, k2
 * (prover.rs:99-100: 2 and 3) as 32-byte scalars in scalar_fmt.  z_out receives z_0 .. z_{n-1} (z_0 = 1).
 * BP_ERR_DIV_ZERO where a denominator is zero (invert().unwrap()), BP_ERR_ASSERT where z_n != 1 (:319).
 * One batched inversion for all 3n denominators: z_i = prefix(num)_i * suffix(den)_i / prod(den). */
int  bp_grand_product(bp_ctx* ctx, const void* a, const void* b, const void* c, const void* s1, const void* s2, const void* s3,
                      size_t n, const void* beta32, const void* gamma32, const void* k1_32, const void* k2_32, int scalar_fmt,
                      void* z_out);

/* ---- the same operators on HBM-resident data (Montgomery limbs) -------------------
 * d_* are device pointers (hipMalloc / torch CUDA tensors); scalars and results of O(1) size stay on the host.
 * Semantics, length rules, quirks and error codes are those of the host-pointer entry points above. */
int  bp_poly_add_device(bp_ctx* ctx, const void* d_a, size_t na, const void* d_b, size_t nb, int basis, void* d_out, size_t* n_out);
int  bp_poly_sub_device(bp_ctx* ctx, const void* d_a, size_t na, const void* d_b, size_t nb, int basis, void* d_out, size_t* n_out);
int  bp_poly_scalar_op_device(bp_ctx* ctx, const void* d_a, size_t n, int basis, const void* s32_mont, int op, void* d_out);
int  bp_poly_mul_device(bp_ctx* ctx, const void* d_a, size_t na, const void* d_b, size_t nb, int basis, void* d_out, size_t* n_out);
int  bp_poly_div_device(bp_ctx* ctx, const void* d_a, size_t na, const void* d_b, size_t nb, int basis, void* d_out, size_t* n_out);
int  bp_poly_evaluate_device(bp_ctx* ctx, const void* d_coeffs, size_t n, int basis, const void* x32_mont, void* out32_mont);
/* out[i] = a[i] * w^i, i.e. p(x) -> p(w x)  (prover.rs:661-674 monomial_z_to_z_omega) */
int  bp_poly_scale_powers_device(bp_ctx* ctx, const void* d_a, size_t n, const void* w32_mont, void* d_out);
int  bp_roots_of_unity_device(bp_ctx* ctx, uint64_t group_order, void* d_out);
int  bp_grand_product_device(bp_ctx* ctx, const void* d_a, const void* d_b, const void* d_c, const void* d_s1, const void* d_s2,
                             const void* d_s3, size_t n, const void* beta32, const void* gamma32, const void* k1_32,
                             const void* k2_32, void* d_z);
int  bp_commit_device(bp_ctx* ctx, uint64_t srs_handle, const void* d_coeffs, size_t n, int basis, uint8_t out96[96]);
/* `count` commitments against one SRS (prover.rs:249-251, 483-485, 640-641 commit two or three polynomials in a row): polynomial i
 * = n[i] Montgomery coefficients in HBM at d_coeffs[i]; out96 receives count x 96 bytes.  The pipelines run together. */
int  bp_commit_many_device(bp_ctx* ctx, uint64_t srs_handle, const void* const* d_coeffs, const size_t* n, size_t count, int basis,
                           uint8_t* out96);

/* Setup::commit (setup.rs:32-37): asserts Monomial basis, MSM of the coefficients against the SRS. */
int  bp_commit(bp_ctx* ctx, uint64_t srs_handle, const void* coeffs, size_t n, int basis, int scalar_fmt,
               uint8_t out96[96]);

/* ---- Prover::prove (src/prover.rs:64-175; rounds :177-647) -- SURVEY.md 8f "next" rows 1-3 ------------------ */
/* Preprocessed circuit = CommonPreprocessedInput (src/program.rs:34-50): the eight Lagrange columns
 * QL QR QM QO QC S1 S2 S3 (in this order) of a 2^log_n-row circuit, log_n >= 3.  Uploaded once and kept in HBM
 * with the forms the reference re-derives inside every proof (coefficient forms, prover.rs:379-386).
 * columns_on_device != 0: the eight pointers are HBM addresses of Montgomery data. */
int  bp_circuit_load(bp_ctx* ctx, uint32_t log_n, const void* const columns[8], int scalar_fmt, int columns_on_device,
                     uint64_t* circuit_handle);
int  bp_circuit_free(bp_ctx* ctx, uint64_t circuit_handle);
/* The verifier's preprocessing (src/verifier.rs:61-68: i_ntt + Setup::commit of each column): commitments to
 * QL QR QM QO QC S1 S2 S3 in that order, 8 x 96 bytes (uncompressed affine), from the coefficient forms already in HBM. */
int  bp_circuit_commitments(bp_ctx* ctx, uint64_t srs_handle, uint64_t circuit_handle, uint8_t out768[768]);
/* Program::make_s_polynomials (src/program.rs:76-147) for circuits of any size: the reference labels every cell with
 * roots_of_unity(group_order)[row] recomputed per cell (utils.rs:29-36), i.e. O(n^2) field work; this is the same map in
 * O(n log n) on the host (no GPU needed).  wire_ids: 3 * 2^log_n variable ids, row-major, columns L R O inside a row;
 * 0 = empty wire (the reference's None), equal ids = the same variable name.  Empty rows after the constraints are rows
 * of zeros.  Cells of one variable form a cycle in row-major order and the NEXT cell of the cycle receives THIS cell's
 * label column * w^row, columns numbered 1 2 3 (program.rs:126-137).  s1 s2 s3: 2^log_n Montgomery scalars each, host. */
int  bp_make_s_polynomials(uint32_t log_n, const uint32_t* wire_ids, void* s1, void* s2, void* s3);
/* One proof.  a, b, c: the witness as the three Lagrange wire columns the reference builds at prover.rs:186-227;
 * public_input: the Lagrange column of prover.rs:114-127 (-x_i in the first rows, zero elsewhere), NULL = all zero;
 * 2^log_n scalars each.  blinders: b1..b11 of prover.rs:110 as 11 x 32 canonical little-endian bytes -- an input here
 * (prove_with_blinding), because the reference draws them from thread_rng and is therefore not reproducible.
 * The SRS should hold 2^log_n + 6 points (verify_proof_test.rs:16); a shorter one truncates the commitments the way
 * Setup::commit's zip does (msm.rs:29), as in the reference.  Challenges come from the reference's transcript
 * (src/transcript.rs:4-86; merlin 3.0.0 restated on the host).
 * proof: 9 x 48-byte compressed G1 points in Proof field order (verifier.rs:23-40: a_1 b_1 c_1 z_1 t_lo_1 t_mid_1
 * t_hi_1 w_zeta_1 w_zeta_omega_1), then a_bar b_bar c_bar s1_bar s2_bar z_omega_bar as 32-byte little-endian.
 * Errors mirror the reference's panics: BP_ERR_ASSERT for z_n != 1 (prover.rs:319) and r(zeta) != 0 (:615), i.e. a
 * witness that does not satisfy the circuit; BP_ERR_DIV_ZERO for a zero permutation denominator. */
int  bp_prove(bp_ctx* ctx, uint64_t srs_handle, uint64_t circuit_handle, const void* a, const void* b, const void* c,
              const void* public_input, int scalar_fmt, int witness_on_device, const uint8_t blinders[352],
              uint8_t proof[624]);
/* Host wall-clock milliseconds of rounds 1..5 and of the whole last bp_prove on this ctx. */
int  bp_prove_last_stats(bp_ctx* ctx, float round_ms[5], float* total_ms);
/* merlin conformance vector: Transcript::new("test protocol"), append_message("some label", "some data"),
 * challenge_bytes("challenge", 32) -- lets a binding check the host transcript without a GPU. */
int  bp_transcript_test_vector(uint8_t out32[32]);

#ifdef __cplusplus
}
#endif
#endif
