// ctx.hpp -- per-GPU context shared by the translation units of libbp_msm_ntt.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <string>
#include <vector>

#include "../../include/bp_msm_ntt.h"
#include "g1.hpp"
#include "g1_28.hpp"

namespace bp {

// Experiment knobs.  The shipped library reads NO environment variable: every default is the measured best and nothing in the
// embedding process's environment can change a kernel shape.  A build with -DBP_EXPERIMENT (`make exp` -> libbp_msm_ntt_exp.so,
// loaded by the tests under tests/ that are marked `experiment`, and by the A/B tools) reads the BP_* variables of DESIGN.md
// section 12, compiles the alternative builds they select (the one-histogram counting sort, every-position NAF tables) and
// honours the test hook BP_FORCE_PEER_COPIES.
#ifdef BP_EXPERIMENT
inline const char* knob(const char* name) { return getenv(name); }
constexpr bool EXPERIMENT_BUILD = true;
#else
inline const char* knob(const char*) { return nullptr; }
constexpr bool EXPERIMENT_BUILD = false;
#endif
// numeric knob: taken only inside [lo, hi]; absent, malformed or out of range leaves the default
inline uint32_t knob_u32(const char* name, uint32_t dflt, uint32_t lo = 0, uint32_t hi = 0xffffffffu) {
  const char* v = knob(name);
  if (!v || !*v) return dflt;
  char* end = nullptr;
  const unsigned long x = strtoul(v, &end, 10);
  return (end == v || *end || x < lo || x > hi) ? dflt : (uint32_t)x;
}

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

// SrsEntry::table_c = MSM_NAF_FLAG | w: tables of every bit position (MSM_NAF_ROWS rows) for width-w NAF digits (msm_kernels.hpp MsmPlan::naf)
constexpr uint32_t MSM_NAF_FLAG = 0x100u, MSM_NAF_ROWS = 256;

struct SrsEntry {
  g1_affine* d_points = nullptr;        // 96 B/point, reference Montgomery limbs (export, generic kernels)
  g1_affine28* d_points28 = nullptr;    // 128-B slot per point (112 B used), 14 x 28-bit limbs, R' = 2^392 (bucket accumulation)
  g1_affine28* d_table = nullptr;       // optional fixed-base tables: row w = 2^(table_c w) * SRS, table_W rows of n points
  uint32_t table_c = 0, table_W = 0;
  size_t n = 0;                         // points resident on this device
  // multi-GPU group (bp_init_multi): the leader's entry describes the whole SRS, every member holds one contiguous point range
  size_t first = 0;                     // global index of this device's first point
  size_t n_global = 0;                  // length of the whole SRS (== n on a single-device context)
  std::vector<uint64_t> member_handle;  // leader only: handle of the shard on member r (index 0 = the leader's own entry)
};

// preprocessed circuit of the native prover (prover.hip): CommonPreprocessedInput (program.rs:34-50) resident in HBM,
// with the derived forms the reference recomputes in every proof.  Column order: ql qr qm qo qc s1 s2 s3.
// Round 3 of a proof split by COSET over the members of a group context: the quotient coset g <w_4n> is the
// union of the four cosets s_j <w_n>, s_j = g w_4n^j; a member evaluates a, b, c, z, PI on its cosets from the five coefficient
// vectors (size-n transforms), runs the quotient kernel there against its own quarter(s) of the circuit's coset tables and sends
// n quotient coefficients per coset back.  One share per participating member (the first 2 or 4), resident from bp_circuit_load.
struct CosetShare {
  bp_ctx* member = nullptr;
  bp_ctx* work = nullptr;              // the member's side context (own stream, workspace, transform tables): where the share's round-3 work runs
  uint32_t first = 0, count = 0;       // cosets [first, first + count) of the four
  fr_t* pre = nullptr;                 // count x 9 x n: the eight circuit columns + L1 on each coset (coset j = entries 4 i + j of CircuitEntry::coset)
  fr_t* xs = nullptr;                  // count x n: the coset's points s_j w_n^i
  fr_t* spow = nullptr;                // count x n: s_j^i
  fr_t* sinv = nullptr;                // count x n: s_j^-i
  hipEvent_t done = nullptr;           // recorded on the work stream behind the copy of its quotient coefficients to the leader
};

struct CircuitEntry {
  uint32_t log_n = 0;
  fr_t* lag = nullptr;        // 8 x n Lagrange columns as loaded
  fr_t* coef = nullptr;       // 8 x n coefficient forms (i_ntt, prover.rs:379-386)
  fr_t* coset = nullptr;      // 9 x 4n evaluations on the quotient coset g <w_4n> (the eight columns + L1)
  fr_t* coset_x = nullptr;    // 4n coset points g w_4n^i
  fr_t* roots = nullptr;      // w_n^i, i < n (roots_of_unity(group_order), prover.rs:282)
  fr_t* g_pow = nullptr;      // g^i, i < n + 8
  fr_t* ginv_pow = nullptr;   // g^-i, i < 4n
  fr_t zh_inv[4];             // 1 / (X^n - 1) on the coset (period 4)
  std::vector<CosetShare> split;       // group contexts only: round 3 by coset over the members (empty otherwise)
};

struct MemberWorker;     // persistent host thread of one member of a group context (capi_ctx.hip)

struct tw29_t;
struct NttTables {       // per (log_n, inverse); entries are 48-byte 29-bit-limb twiddle records (fr29.hpp)
  tw29_t* lo = nullptr;        // w_N^j, j < 2^h
  tw29_t* hi = nullptr;        // w_N^(j << h), j < 2^(k-h)
  tw29_t* hi_scaled = nullptr; // hi * N^-1 (inverse transforms, first pass)
  tw29_t* n_inv_tw = nullptr;  // N^-1 as a twiddle record (single-pass inverses)
  fr_t* n_inv = nullptr;       // N^-1, Montgomery (table builder input)
  uint32_t h = 0;
  // inter-pass twiddles as full tables (one per strided pass): full[i][(e << s_i) + r] = w_M^(e r), M = 2^(l_i + s_i), N^-1
  // folded into pass 0 of an inverse -- one product per element instead of lookup product + application.  Built on first
  // use for N <= 2^24 (48 B per entry: 50 MB at 2^20, 805 MB at 2^24); nullptr = use lo/hi.
  tw29_t* full[3] = {nullptr, nullptr, nullptr};
  uint32_t full_ls[3] = {0, 0, 0};     // (l << 8) | s the table was built for (the experiment knobs can change the split)
};

}  // namespace bp

struct bp_ctx {
  int device = 0;
  // multi-GPU group (bp_init_multi): members[0] == this for the leader, empty for a plain bp_init context.
  // Every member is a full single-device context (own stream, workspaces, NTT tables) driven by the leader's host thread.
  std::vector<bp_ctx*> members;
  bp_ctx* leader = nullptr;                        // set on members[1..]
  // a group whose device list names a GPU more than once ({0, 0}: what a one-GPU box can rehearse; production lists are distinct):
  // its members beyond the leader take the GPU-to-GPU branches -- hipMemcpyPeerAsync behind the leader's event -- exactly as
  // members on other cards do, so the lines a multi-GPU node executes run in the default test suite, with no knob
  bool rehearsal = false;
  // leader only: workers[r - 1] is the host thread that drives member r whenever the members work from or to host memory, or
  // wait for their streams: created once in bp_init_multi, reused by every call (no thread is spawned per MSM or per transform)
  std::vector<bp::MemberWorker*> workers;
  // Extra single-device contexts on THIS device (own stream + workspaces, created on first use): independent commitments of
  // one caller -- the three of prover rounds 1 and 3, the two of round 5, the verifier's eight -- run on them concurrently,
  // so one MSM's latency-bound tail (bucket tree, fix-up, scans) overlaps another's bulk kernel.
  std::vector<bp_ctx*> lanes;
  // One more single-device context on this device for prover work that does not wait for a Fiat-Shamir challenge (the coset
  // evaluations of a, b, c and PI, prover.rs:386-450, are known after round 1): enqueued beside rounds 1-2, it fills the GPU time
  // the commitments' latency-bound tails and the host's transcript steps leave idle.  side_ev: [0] inputs ready, [1] side work done.
  bp_ctx* side = nullptr;
  hipEvent_t side_ev[2] = {nullptr, nullptr};
  hipEvent_t seam_ev[4] = {nullptr, nullptr, nullptr, nullptr};      // bp_msm_g1_projective144: piece k uploaded and normalised (recorded on the side stream)
  hipStream_t stream = nullptr;
  bool own_stream = true;
  // experiment (BP_ACC_LOW_PRIORITY=1): msm_accumulate runs on a lowest-priority stream of its own, chained to `stream` by
  // events, while `stream` is created at the highest priority -- so that sort / fix-up / tree kernels of OTHER pipelines take the
  // workgroup slots a retiring accumulation generation frees.  Measured: profiles/r05_accumulate_generations_ab.txt.
  hipStream_t acc_stream = nullptr;
  hipEvent_t acc_ev[2] = {nullptr, nullptr};
  std::string last_error;
  std::map<std::string, bp::DevBuf> ws;            // grow-only named device workspaces
  std::map<uint64_t, bp::SrsEntry> srs;
  std::map<uint64_t, bp::CircuitEntry> circuits;
  uint64_t next_handle = 1;
  bp::tw29_t* small_tw[2] = {nullptr, nullptr};    // w_1024^j, j < 512: forward / inverse
  std::map<uint32_t, bp::NttTables> ntt_tables;    // key = log_n * 2 + inverse
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // [0..3] MSM / NTT timing, [4] cross-stream ordering (groups)
  // stats of the last calls
  float msm_accumulate_ms = 0, msm_total_ms = 0;
  float msm_upload_ms = 0;         // host scalars: the H2D copy in front of the last MSM (ev[4] .. ev[0]); else 0
  float shard_accumulate_ms = 0, shard_total_ms = 0;    // this member's own share of the last MSM (the fields above hold the group's
  uint64_t shard_adds = 0;                               // figures on the leader: slowest shard, all additions)
  uint64_t msm_adds = 0;
  uint32_t msm_c = 0;
  bool msm_tables = false;
  float prove_ms[6] = {0, 0, 0, 0, 0, 0};            // host wall clock of rounds 1..5 and of the whole bp_prove
  bool msm_async_pending = false;  // bp_msm_g1_blob_device_async enqueued an MSM whose events have not been read yet
  float ntt_ms = 0;
  bool ntt_async_pending = false;  // bp_ntt_fr_device_async enqueued a transform whose events have not been read yet
  uint32_t ntt_passes = 0;
  uint32_t ntt_members = 1;       // members of a group context that took part in the last host transform
  // one process per GPU (capi_comm.hip): the RCCL communicator of this context (ncclComm_t), created by bp_comm_init_rank
  void* comm = nullptr;
  int comm_rank = 0, comm_world = 0;
  hipEvent_t comm_ev[2] = {nullptr, nullptr};      // record complete / gathered, summed and copied
  // every buffer a collective needs exists from bp_comm_init_rank on (nothing between "called" and "joined the collective" can run
  // out of memory): comm_dev = mine | summed | gathered[world] records, then the agreement words (mine | all[world]); comm_host = the
  // pinned mirror (world records + the agreement words)
  uint8_t* comm_dev = nullptr;
  void* comm_host = nullptr;
  size_t comm_host_cap = 0;
  float comm_exchange_ms = 0;
  uint32_t comm_timeout_ms = 120000;               // bound of every wait on a collective and of ncclCommInitRank (bp_comm_set_timeout_ms; 0 = none)
  uint64_t comm_collectives = 0;                   // collectives this context has enqueued on its communicator(s) (bp_comm_stats)
  bool comm_init_stuck = false;                    // an ncclCommInitRank of this context ran into the bound: its helper thread is still inside RCCL
  uint64_t free_bytes_probe = 0;                   // tests only (bpx_set_free_bytes_probe): the free-memory reading bp_srs_precompute decides on; 0 = hipMemGetInfo
  bool gen_table_ready = false;                    // srs_generate_run: the generator's multiples were built into gen_table_ptr
  const void* gen_table_ptr = nullptr;
  void* pinned = nullptr;                          // small pinned staging buffer (window sums etc.)
  size_t pinned_cap = 0;
};

namespace bp {

int fail(bp_ctx* ctx, int code, const char* what, hipError_t e, const char* file, int line);

// Every entry point that takes a ctx runs under one of these: the calling thread's current device is restored on return
// (torch shares this HIP runtime; a library call must not leave the thread on another GPU).
// The OUTERMOST guard of a thread -- the one an extern "C" entry point (or a member's worker task) opens -- also drops the thread's
// pending HIP error first: hipGetLastError() after a launch reports the LAST error of the calling thread, whoever caused it -- the
// embedding process (torch, RCCL, the Rust host's own HIP calls) or a previous context's teardown (gpurun_out/r3_t37.log: "invalid
// device ordinal" surfacing in the next bp_init) -- and an innocent bp_* call must not return it as BP_ERR_HIP.  Guards nested inside a
// call (helpers, teardown loops, the memory-budget probe) have no such side effect: a launch error raised earlier in the SAME call is
// still there for the post-launch hipGetLastError() that follows it.
inline int& device_guard_depth() {
  static thread_local int depth = 0;
  return depth;
}
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int device) {
    if (device_guard_depth()++ == 0) (void)hipGetLastError();
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) (void)hipSetDevice(device);
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    device_guard_depth()--;
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
inline bool is_group(const bp_ctx* ctx) { return ctx->members.size() > 1; }
// does a copy between the leader's memory and member m's cross devices?  (m on another GPU, or a rehearsal group's member)
inline bool peer_path(const bp_ctx* m, int other_device) {
  if (m->device != other_device) return true;
  if (m->leader && m->leader->rehearsal) return true;
  const char* v = knob("BP_FORCE_PEER_COPIES");
  return v && *v && *v != '0';
}

#define BP_HIP(ctx, call)                                                              \
  do {                                                                                 \
    hipError_t e__ = (call);                                                           \
    if (e__ != hipSuccess) return ::bp::fail(ctx, BP_ERR_HIP, #call, e__, __FILE__, __LINE__); \
  } while (0)
#define BP_TRY(expr)              \
  do {                            \
    int rc__ = (expr);            \
    if (rc__ != BP_OK) return rc__; \
  } while (0)

// Wait for a stream.  A thread blocked in hipStreamSynchronize wakes up tens of microseconds after the GPU has finished; the
// calls of this library are a few milliseconds long and wait several times each (nine commitments and ~30 small results per
// proof), so the wait polls hipStreamQuery for the first few milliseconds and only then blocks (experiment build: BP_WAIT_BLOCK=1 blocks at once).
hipError_t stream_wait(hipStream_t st);

// device workspace `name` of at least `bytes` (contents undefined after growth)
int ws_get(bp_ctx* ctx, const char* name, size_t bytes, void** out);
int pinned_get(bp_ctx* ctx, size_t bytes, void** out);

// ---- launchers implemented in msm.hip / ntt.hip / poly.hip / srs.hip --------------------------------
// One MSM = msm_launch (everything enqueued on ctx->stream, nothing waited for) + msm_finish (wait, status, host epilogue).
// The split lets one host thread keep several devices (or several MSMs of one stream) in flight.  `slot` selects one of
// MSM_SLOTS result areas in the pinned staging buffer; launches on one stream reuse the device workspaces in stream order.
constexpr int MSM_SLOTS = 4;
struct MsmPending {
  bool empty = true, blob = false;
  uint32_t tables = 0;                  // 0: per-window buckets; 1: fixed-base window tables; 2: every-position tables (odd NAF digits)
  uint32_t c = 0, Wr = 0, n_planes = 0;
  uint32_t quads = 1;                   // table-free: values per window the device hands over (msm_planes_window_quads); 1: whole window sums
  uint32_t J = 1;                       // scalar vectors in the pipeline (msm_launch_many): msm_finish writes J results
  uint64_t adds = 0;
  const void* h_windows = nullptr;      // pinned: n_planes accumulator slots + the status word
};
// d_blob != nullptr: the result stays in HBM as a BP_MSM_BLOB_BYTES record (msm_kernels.hpp) instead of the pinned slot
int msm_launch(bp_ctx* ctx, const g1_affine28* d_points28, size_t n, const fr_t* d_scalars, int fmt, uint32_t table_c, size_t table_stride,
               int slot, void* d_blob, MsmPending* out);
int msm_launch_many(bp_ctx* ctx, const g1_affine28* d_points28, uint32_t J, const fr_t* const* d_scalars_each, const size_t* n_each, int fmt,
                    uint32_t table_c, size_t table_stride, int slot, void* d_blob, MsmPending* out);
int msm_blobs_combine(const uint8_t* blobs, size_t n_blobs, g1_proj* out);
int msm_blobs_sum_device_run(bp_ctx* ctx, const void* d_blobs, size_t n_blobs, void* d_out, bool wait = true);
int msm_blob_poisoned(const uint8_t* blobs, size_t n_blobs, uint32_t* rank);
int msm_blob_poison_run(bp_ctx* ctx, void* d_blob, int err, uint32_t rank);
int msm_finish(bp_ctx* ctx, const MsmPending& pend, g1_proj* host_out);
int msm_run(bp_ctx* ctx, const g1_affine28* d_points28, size_t n, const fr_t* d_scalars, int fmt, uint32_t table_c, size_t table_stride,
            g1_proj* host_out);
int msm_init_device(bp_ctx* ctx);
// k independent commitments sum_i coeffs_j[i] * SRS[i] (Setup::commit, setup.rs:32-37) of HBM-resident Montgomery coefficient
// vectors, all in flight together; out[j] in the order given
int commit_many(bp_ctx* ctx, uint64_t srs_handle, const fr_t* const* d_coeffs, const size_t* n, int k, g1_proj* out);
int srs_tables_run(bp_ctx* ctx, const g1_affine* d_points, const g1_affine28* d_points28, size_t n, uint32_t c, g1_affine28** d_table,
                   uint32_t* windows);
uint32_t srs_table_rows(uint32_t c);                 // rows of the fixed-base tables of window width c (msm.hip)
int srs_to28_run(bp_ctx* ctx, const g1_affine* d_in, size_t n, g1_affine28** d_out);
int srs_to28_into(bp_ctx* ctx, const g1_affine* d_in, size_t n, g1_affine28* d_out);      // the same into a buffer of the caller's, on ctx->stream
int ntt_init_tables(bp_ctx* ctx);
int ntt_run(bp_ctx* ctx, fr_t* d_data, uint32_t log_n, int inverse, size_t batch, size_t stride);
// one transform in two phases over the members of a group context (ntt.hip)
bool ntt_split_ok(uint32_t log_n, uint32_t parts);
void ntt_split_shape(uint32_t log_n, uint32_t* l1);
int ntt_tmp_buffer(bp_ctx* ctx, uint32_t log_n, fr_t** out);
int ntt_run_part(bp_ctx* ctx, fr_t* d_data, uint32_t log_n, int inverse, size_t batch, size_t stride, int phase, uint32_t part, uint32_t parts);
int fr_convert_run(bp_ctx* ctx, fr_t* d, size_t n, int dir);
int fr_binary_run(bp_ctx* ctx, const fr_t* a, size_t na, const fr_t* b, size_t nb, fr_t* out, size_t n, int op);
int fr_scalar_run(bp_ctx* ctx, const fr_t* a, const fr_t& s, fr_t* out, size_t n, int op);
int poly_eval_run(bp_ctx* ctx, const fr_t* d_coeffs, size_t n, const fr_t& x, fr_t* host_out);
int poly_eval_many_run(bp_ctx* ctx, int k, const fr_t* const* d_coeffs, const size_t* n, const fr_t* x, fr_t* host_out);
int poly_div_run(bp_ctx* ctx, fr_t* d_a, size_t na, const fr_t* d_b, size_t nb, const fr_t& b0, const fr_t& b_lead, bool binomial,
                 fr_t* d_q, size_t nq);
int fr_nonzero_stats_run(bp_ctx* ctx, const fr_t* d_a, size_t n, size_t lo, size_t hi, size_t* eff_len, size_t* nonzero_in_range);
int fr_compact_nonzero_run(bp_ctx* ctx, fr_t* d_q, size_t* n);
int fr_scale_powers_run(bp_ctx* ctx, const fr_t* d_a, size_t n, const fr_t& w, fr_t* d_out);
int fr_synthetic_run(bp_ctx* ctx, fr_t* d_out, size_t n, uint64_t seed);
int fr_scan_mul_run(bp_ctx* ctx, const fr_t* d_in, size_t n, int reverse, int inclusive, fr_t* d_out, fr_t* d_total);
int grand_product_run(bp_ctx* ctx, const fr_t* a, const fr_t* b, const fr_t* c, const fr_t* s1, const fr_t* s2, const fr_t* s3, size_t n,
                      const fr_t& beta, const fr_t& gamma, const fr_t& k1, const fr_t& k2, const fr_t& root, fr_t* d_z,
                      const fr_t* d_roots = nullptr);
int roots_run(bp_ctx* ctx, const fr_t& w, size_t n, fr_t* d_out);
int srs_decode_run(bp_ctx* ctx, const uint8_t* d_bytes, size_t n, g1_affine* d_out);
int srs_encode_run(bp_ctx* ctx, const g1_affine* d_in, size_t n, uint8_t* d_bytes);
int srs_from_projective_run(bp_ctx* ctx, const g1_proj* d_in, size_t n, g1_affine* d_out);
int srs_generate_run(bp_ctx* ctx, const fr_t& a, const fr_t& d, int mode, size_t first, size_t n, g1_affine* d_out);

void comm_release(bp_ctx* ctx);                   // capi_comm.hip: destroys the context's communicator, if any
int side_ctx_get(bp_ctx* ctx, bp_ctx** out);       // creates ctx->side and its events on first use (capi_ctx.hip)
int circuit_build(bp_ctx* ctx, uint32_t log_n, fr_t* d_lag, CircuitEntry* out);
int circuit_split_build(bp_ctx* ctx, CircuitEntry& e);      // leader of a group: the members' coset shares (prover.hip)
void circuit_release(CircuitEntry& e);
// Witness columns still in host memory (the Rust caller's Vec<Scalar>s): round 1 uploads them itself on the side context's stream, one
// column at a time, and starts each column's commitment as soon as its coefficients exist -- the pageable copies of b and c (and the
// host thread blocked in them) run beside the commitment of a.  cols: a, b, c, PI (PI may be null); fmt: BP_FR_MONT / BP_FR_BYTES_LE.
struct ProveStaged {
  const void* cols[4];
  int fmt;
};
// one commitment on lane j (< 3) of a single-device context, enqueued behind `ready`; waits for nothing.  commit_lane_finish waits.
int commit_lane_launch(bp_ctx* ctx, int j, uint64_t srs_handle, const fr_t* d_coeffs, size_t n, hipEvent_t ready, MsmPending* pend);
int commit_lane_finish(bp_ctx* ctx, int j, const MsmPending& pend, g1_proj* out);
// pi_zero: the caller passed no public inputs, so the PI column of d_wit is all zero (PI(X) = 0: its transforms are skipped, not computed)
int prove_run(bp_ctx* ctx, uint64_t srs, const CircuitEntry& cir, const fr_t* d_wit, const fr_t blind[11], uint8_t proof[624], bool pi_zero = false,
              const ProveStaged* staged = nullptr);
void transcript_test_vector(uint8_t out32[32]);

// ---- host-side helpers (capi_ctx.hip) ---------------------------------------------------
void host_horner(g1_proj& out, const g1_proj* window_sums, uint32_t W, uint32_t c);
void host_quad_horner(g1_proj& out, const g1_proj* quads, uint32_t W, uint32_t c, uint32_t nq);
void host_plane_horner(g1_proj& out, const g1_proj* planes, uint32_t W, uint32_t c, bool odd_digits = false);
void host_encode96(uint8_t out96[96], const g1_proj& p);
bool host_decode96(g1_proj& out, const uint8_t in96[96]);
void host_compress48(uint8_t out48[48], const g1_proj& p);

}  // namespace bp
