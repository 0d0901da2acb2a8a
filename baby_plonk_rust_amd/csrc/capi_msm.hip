// capi_msm.hip -- C ABI, part 2: the SRS (Setup.powers_of_x resident in HBM, fixed-base tables) and the G1 MSM entry points
// (BucketMSM::bucket_msm, Setup::commit over one GPU or the shards of a group context, records for the one-process-per-GPU path).
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "ctx.hpp"

#include "capi_common.hpp"

using namespace bp;
// ---------------------------------------------------------------------------------------------- SRS
// takes ownership of d (freed on failure); the entry covers global points [first, first + n) of an SRS of n_global points
static int srs_register(bp_ctx* ctx, g1_affine* d, size_t n, size_t first, size_t n_global, uint64_t* handle) {
  SrsEntry e;
  e.d_points = d;
  e.n = n;
  e.first = first;
  e.n_global = n_global;
  int rc = srs_to28_run(ctx, d, n, &e.d_points28);
  if (rc == BP_OK && stream_wait(ctx->stream) != hipSuccess) rc = fail(ctx, BP_ERR_HIP, "srs_to28", hipGetLastError(), __FILE__, __LINE__);
  if (rc != BP_OK) {
    (void)hipFree(d);
    if (e.d_points28) (void)hipFree(e.d_points28);
    return rc;
  }
  *handle = ctx->next_handle++;
  ctx->srs[*handle] = e;
  return BP_OK;
}
int srs_find(bp_ctx* ctx, uint64_t handle, SrsEntry** out) {
  auto it = ctx->srs.find(handle);
  if (it == ctx->srs.end()) return fail(ctx, BP_ERR_INVALID_ARG, "unknown SRS handle", hipSuccess, __FILE__, __LINE__);
  *out = &it->second;
  return BP_OK;
}
// handle of the shard held by member r (the leader's entry lists them; a plain context has only its own)
static uint64_t member_handle(const SrsEntry& lead, uint64_t own, size_t r) { return lead.member_handle.empty() ? own : lead.member_handle[r]; }

static int srs_free_one(bp_ctx* ctx, uint64_t handle) {
  SrsEntry* e;
  BP_TRY(srs_find(ctx, handle, &e));
  DeviceGuard guard(ctx->device);
  BP_HIP(ctx, stream_wait(ctx->stream));
  BP_HIP(ctx, hipFree(e->d_points));
  BP_HIP(ctx, hipFree(e->d_points28));
  if (e->d_table) BP_HIP(ctx, hipFree(e->d_table));
  ctx->srs.erase(handle);
  return BP_OK;
}

// One shard of an SRS on one device.  kind 0: decode 96-byte encodings, 1: normalise 144-byte projective images,
// 2: generate tau^i G, 3: generate (a + i d) G.  src: this shard's slice of the host input (kinds 0, 1).
static int srs_make_one(bp_ctx* ctx, int kind, const uint8_t* src, const fr_t& a, const fr_t& d, size_t first, size_t n, size_t n_global,
                        uint64_t* handle) {
  DeviceGuard guard(ctx->device);
  g1_affine* d_pts = nullptr;
  BP_HIP(ctx, hipMalloc((void**)&d_pts, std::max<size_t>(n, 1) * sizeof(g1_affine)));
  int rc = BP_OK;
  if (kind == 0 || kind == 1) {
    const size_t rec = kind == 0 ? 96 : 144;
    uint8_t* d_bytes = nullptr;
    rc = ws_get(ctx, "io.bytes", n * rec, (void**)&d_bytes);
    if (rc == BP_OK && n) {
      hipError_t e = hipMemcpyAsync(d_bytes, src, n * rec, hipMemcpyHostToDevice, ctx->stream);
      if (e != hipSuccess) rc = fail(ctx, BP_ERR_HIP, "SRS upload", e, __FILE__, __LINE__);
    }
    if (rc == BP_OK) rc = kind == 0 ? srs_decode_run(ctx, d_bytes, n, d_pts) : srs_from_projective_run(ctx, (const g1_proj*)d_bytes, n, d_pts);
  } else {
    rc = srs_generate_run(ctx, a, d, kind == 2 ? 0 : 1, first, n, d_pts);
  }
  if (rc == BP_OK) {
    hipError_t e = stream_wait(ctx->stream);
    if (e != hipSuccess) rc = fail(ctx, BP_ERR_HIP, "SRS build", e, __FILE__, __LINE__);
  }
  if (rc != BP_OK) {
    (void)hipFree(d_pts);
    return rc;
  }
  return srs_register(ctx, d_pts, n, first, n_global, handle);
}

// the whole SRS: one shard per member (contiguous point ranges), the leader's entry lists the members' handles
static int srs_make(bp_ctx* ctx, int kind, const uint8_t* src, const fr_t& a, const fr_t& d, size_t n, uint64_t* handle) {
  const std::vector<bp_ctx*> sh = shards_of(ctx);
  const size_t rec = kind == 0 ? 96 : 144;
  std::vector<uint64_t> hs;
  for (size_t r = 0; r < sh.size(); r++) {
    size_t lo, hi;
    shard_range(n, r, sh.size(), &lo, &hi);
    uint64_t h = 0;
    int rc = lift(ctx, sh[r], srs_make_one(sh[r], kind, src ? src + lo * rec : nullptr, a, d, lo, hi - lo, n, &h));
    if (rc != BP_OK) {
      for (size_t k = 0; k < hs.size(); k++) (void)srs_free_one(sh[k], hs[k]);
      return rc;
    }
    hs.push_back(h);
  }
  if (sh.size() > 1) ctx->srs[hs[0]].member_handle = hs;
  *handle = hs[0];
  return BP_OK;
}

int bp_srs_load(bp_ctx* ctx, const uint8_t* points96, size_t n, uint64_t* srs_handle) {
  if (!ctx || !srs_handle || (n && !points96)) return BP_ERR_INVALID_ARG;
  return srs_make(ctx, 0, points96, Fr::zero(), Fr::zero(), n, srs_handle);
}
int bp_srs_load_projective144(bp_ctx* ctx, const uint8_t* points144, size_t n, uint64_t* srs_handle) {
  if (!ctx || !srs_handle || (n && !points144)) return BP_ERR_INVALID_ARG;
  return srs_make(ctx, 1, points144, Fr::zero(), Fr::zero(), n, srs_handle);
}

static int srs_generate_common(bp_ctx* ctx, size_t n, const uint8_t a32[32], const uint8_t d32[32], int mode, uint64_t* handle) {
  if (!ctx || !handle || !a32 || (mode == 1 && !d32)) return BP_ERR_INVALID_ARG;
  fr_t a, d = Fr::zero();
  if (!fr_bytes_to_mont(a, a32, BP_FR_BYTES_LE)) return fail(ctx, BP_ERR_BAD_SCALAR, "scalar >= q", hipSuccess, __FILE__, __LINE__);
  if (mode == 1 && !fr_bytes_to_mont(d, d32, BP_FR_BYTES_LE)) return fail(ctx, BP_ERR_BAD_SCALAR, "scalar >= q", hipSuccess, __FILE__, __LINE__);
  return srs_make(ctx, mode == 0 ? 2 : 3, nullptr, a, d, n, handle);
}
int bp_srs_generate(bp_ctx* ctx, size_t powers, const uint8_t tau32[32], uint64_t* srs_handle) {
  return srs_generate_common(ctx, powers, tau32, nullptr, 0, srs_handle);
}
int bp_srs_generate_progression(bp_ctx* ctx, size_t n, const uint8_t a32[32], const uint8_t d32[32], uint64_t* srs_handle) {
  return srs_generate_common(ctx, n, a32, d32, 1, srs_handle);
}

int bp_srs_len(bp_ctx* ctx, uint64_t srs_handle, size_t* n) {
  if (!ctx || !n) return BP_ERR_INVALID_ARG;
  SrsEntry* e;
  BP_TRY(srs_find(ctx, srs_handle, &e));
  *n = e->n_global;
  return BP_OK;
}

int bp_srs_export(bp_ctx* ctx, uint64_t srs_handle, size_t first, size_t n, uint8_t* points96) {
  if (!ctx || (n && !points96)) return BP_ERR_INVALID_ARG;
  SrsEntry* lead;
  BP_TRY(srs_find(ctx, srs_handle, &lead));
  if (first > lead->n_global || n > lead->n_global - first) return fail(ctx, BP_ERR_INVALID_ARG, "SRS range out of bounds", hipSuccess, __FILE__, __LINE__);
  const std::vector<bp_ctx*> sh = shards_of(ctx);
  for (size_t r = 0; r < sh.size(); r++) {
    bp_ctx* m = sh[r];
    SrsEntry* e;
    BP_TRY(lift(ctx, m, srs_find(m, member_handle(*lead, srs_handle, r), &e)));
    const size_t lo = std::max(first, e->first), hi = std::min(first + n, e->first + e->n);
    if (lo >= hi) continue;
    DeviceGuard guard(m->device);
    uint8_t* d_bytes;
    BP_TRY(lift(ctx, m, ws_get(m, "io.bytes", (hi - lo) * 96, (void**)&d_bytes)));
    BP_TRY(lift(ctx, m, srs_encode_run(m, e->d_points + (lo - e->first), hi - lo, d_bytes)));
    BP_HIP(ctx, hipMemcpyAsync(points96 + (lo - first) * 96, d_bytes, (hi - lo) * 96, hipMemcpyDeviceToHost, m->stream));
    BP_HIP(ctx, stream_wait(m->stream));
  }
  return BP_OK;
}

int bp_srs_export_projective144(bp_ctx* ctx, uint64_t srs_handle, size_t first, size_t n, uint8_t* points144) {
  if (!ctx || (n && !points144)) return BP_ERR_INVALID_ARG;
  SrsEntry* lead;
  BP_TRY(srs_find(ctx, srs_handle, &lead));
  if (first > lead->n_global || n > lead->n_global - first) return fail(ctx, BP_ERR_INVALID_ARG, "SRS range out of bounds", hipSuccess, __FILE__, __LINE__);
  const std::vector<bp_ctx*> sh = shards_of(ctx);
  const fp_t one = Fp::one();
  for (size_t r = 0; r < sh.size(); r++) {
    bp_ctx* m = sh[r];
    SrsEntry* e;
    BP_TRY(lift(ctx, m, srs_find(m, member_handle(*lead, srs_handle, r), &e)));
    const size_t lo = std::max(first, e->first), hi = std::min(first + n, e->first + e->n);
    if (lo >= hi) continue;
    DeviceGuard guard(m->device);
    std::vector<g1_affine> aff(hi - lo);
    BP_HIP(ctx, hipMemcpyAsync(aff.data(), e->d_points + (lo - e->first), (hi - lo) * sizeof(g1_affine), hipMemcpyDeviceToHost, m->stream));
    BP_HIP(ctx, stream_wait(m->stream));
    for (size_t i = 0; i < hi - lo; i++) {                   // G1Projective::from(&G1Affine) (g1.rs:176-190): z = 1, or 0 for the identity
      g1_proj p;
      p.x = aff[i].x;
      p.y = aff[i].y;
      p.z = g1_affine_is_identity(aff[i]) ? Fp::zero() : one;
      if (g1_affine_is_identity(aff[i])) p = g1_identity();
      memcpy(points144 + (lo - first + i) * 144, &p, 144);
    }
  }
  return BP_OK;
}

int bp_srs_free(bp_ctx* ctx, uint64_t srs_handle) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  SrsEntry* lead;
  BP_TRY(srs_find(ctx, srs_handle, &lead));
  const std::vector<bp_ctx*> sh = shards_of(ctx);
  const std::vector<uint64_t> hs = lead->member_handle;
  int rc = BP_OK;
  for (size_t r = sh.size(); r-- > 0;) {                 // the leader's entry (r = 0) goes last: it names the others
    const int rc1 = lift(ctx, sh[r], srs_free_one(sh[r], hs.empty() ? srs_handle : hs[r]));
    if (rc == BP_OK) rc = rc1;
  }
  return rc;
}

static int srs_precompute_one(bp_ctx* ctx, uint64_t handle, uint32_t c) {
  SrsEntry* e;
  BP_TRY(srs_find(ctx, handle, &e));
  DeviceGuard guard(ctx->device);
  BP_HIP(ctx, stream_wait(ctx->stream));
  if (e->d_table) {
    BP_HIP(ctx, hipFree(e->d_table));
    e->d_table = nullptr;
    e->table_c = e->table_W = 0;
  }
  if (c == BP_SRS_TABLES_OFF) return BP_OK;
  BP_TRY(srs_tables_run(ctx, e->d_points, e->d_points28, e->n, c, &e->d_table, &e->table_W));
  e->table_c = c;
  return BP_OK;
}

int bp_srs_precompute(bp_ctx* ctx, uint64_t srs_handle, uint32_t window_bits) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  SrsEntry* lead;
  BP_TRY(srs_find(ctx, srs_handle, &lead));
  uint32_t c = window_bits;
  if (c == 0) {                       // auto, from the length of one shard (every shard of a group gets the same width)
    uint32_t lg = 0;
    const uint64_t n = lead->n;
    while ((2ull << lg) <= n) lg++;                       // floor(log2 n)
    if (lg >= 24) {                   // 12 windows: the 2^21-bucket tree (+0.8 ms) against n fewer additions (-1.8 ms at 2^24; a tie at 2^23)
      c = 22;
    } else if (lg >= 20) {            // 13 windows instead of 16: pays once the sort and the 2^19-bucket tree are small against
      c = 20;                         // 3 x n additions (round 3: -3 % at 2^20, -6 % at 2^21, -12 % at 2^22; profiles/r03_window_width_ab.txt)
    } else if (n >= (1u << 14)) {     // throughput regime: reduction work 2^c stays below the bucket-add work W * n
      c = lg + 2 > 16 ? 16 : lg + 2;
    } else {                          // latency regime (a few thousand points): every kernel is a dependent chain, and the
      c = lg > 8 ? lg - 4 : 4;        // reduction tree has c - 1 levels -- measured optimum 2^10: 6, 2^12: 8
    }
  }
  const bool naf = (c & MSM_NAF_FLAG) != 0;
  if (naf && !EXPERIMENT_BUILD)        // every-position tables with NAF digits: measured slower twice, experiment builds only
    return fail(ctx, BP_ERR_INVALID_ARG, "window_bits must be 0 (auto), 1 (off) or 4..24", hipSuccess, __FILE__, __LINE__);
  if (naf ? ((c & 0xffu) < 6 || (c & 0xffu) > 22 || (c >> 9)) : (c != BP_SRS_TABLES_OFF && (c < 4 || c > 24)))
    return fail(ctx, BP_ERR_INVALID_ARG, "window_bits must be 0 (auto), 1 (off), 4..24, or 256 + w (w = 6..22: every-position tables)", hipSuccess,
                __FILE__, __LINE__);
  const std::vector<bp_ctx*> sh = shards_of(ctx);
  const std::vector<uint64_t> hs = lead->member_handle;
  // Memory budget: the tables (rows x points x 128 B: 25.8 GB per GPU at 2^24 points) must fit beside whatever else lives on the
  // device -- a second prover context, the caller's tensors -- together with the workspaces the first MSM against them allocates.
  // The check comes BEFORE anything is released or allocated, and counts the bytes of the tables this SRS holds now as free (they
  // go once the new width is accepted): a refused explicit width therefore leaves the SRS exactly as it was, old tables included.
  // An automatic width that does not fit falls back to wider windows (fewer rows: 22 -> 12, 24 -> 11) -- but only to a width MSMs over
  // this SRS would use (8 n >= 2^c, the rule of msm_shard_launch: a 2^18-point SRS never builds 24-bit tables no MSM would touch) --
  // and then to NO tables (the MSM runs on the raw points, same bytes out: bp_srs_table_info reports what was built); an explicit
  // width that does not fit is an error that says how much is missing, not an out-of-memory failure halfway through the build.
  if (c == BP_SRS_TABLES_OFF) {
    for (size_t r = 0; r < sh.size(); r++) BP_TRY(lift(ctx, sh[r], srs_precompute_one(sh[r], hs.empty() ? srs_handle : hs[r], BP_SRS_TABLES_OFF)));
    return BP_OK;
  }
  auto fits = [&](uint32_t cc, size_t* need_out, size_t* free_out) -> int {
    std::map<int, size_t> per_device, held;              // shards that share a device (a rehearsal group) share its free memory
    for (size_t r = 0; r < sh.size(); r++) {
      SrsEntry* e;
      BP_TRY(lift(ctx, sh[r], srs_find(sh[r], hs.empty() ? srs_handle : hs[r], &e)));
      DeviceGuard guard(sh[r]->device);
      size_t free_b = 0, total_b = 0;
      BP_HIP(ctx, hipMemGetInfo(&free_b, &total_b));
      if (ctx->free_bytes_probe) free_b = (size_t)ctx->free_bytes_probe;        // tests: the reading to decide on (bpx_set_free_bytes_probe below)
      const size_t rows = srs_table_rows(cc);
      size_t& need = per_device[sh[r]->device];
      size_t& old_bytes = held[sh[r]->device];
      if (e->d_table) old_bytes += (size_t)e->table_W * std::max<size_t>(e->n, 1) * sizeof(g1_affine28);
      need += rows * e->n * (sizeof(g1_affine28) + 24) + ((size_t)256 << 20);      // + sort records, lists, partial slots of one MSM
      if (need > free_b + old_bytes) {
        *need_out = need;
        *free_out = free_b + old_bytes;
        return 1;
      }
    }
    return 0;
  };
  size_t need = 0, free_b = 0;
  int rc = fits(c, &need, &free_b);
  if (rc < 0) return rc;
  if (rc == 1) {
    if (window_bits != 0) {
      char msg[200];
      snprintf(msg, sizeof msg, "fixed-base tables of width %u need %.1f GiB on a device with %.1f GiB free", c & 0xffu, need / 1073741824.0, free_b / 1073741824.0);
      return fail(ctx, BP_ERR_TOO_LARGE, msg, hipSuccess, __FILE__, __LINE__);
    }
    uint32_t pick = BP_SRS_TABLES_OFF;
    for (uint32_t cc : {22u, 24u}) {
      if (cc <= c || 8 * (uint64_t)lead->n < (1ull << cc)) continue;
      rc = fits(cc, &need, &free_b);
      if (rc < 0) return rc;
      if (rc == 0) { pick = cc; break; }
    }
    c = pick;
  }
  // accepted (or nothing fits: the SRS ends up without tables): only now do the old tables go
  for (size_t r = 0; r < sh.size(); r++) BP_TRY(lift(ctx, sh[r], srs_precompute_one(sh[r], hs.empty() ? srs_handle : hs[r], BP_SRS_TABLES_OFF)));
  if (c == BP_SRS_TABLES_OFF) return BP_OK;
  for (size_t r = 0; r < sh.size(); r++) BP_TRY(lift(ctx, sh[r], srs_precompute_one(sh[r], hs.empty() ? srs_handle : hs[r], c)));
  return BP_OK;
}

// INTERNAL, not part of include/bp_msm_ntt.h: the memory-budget decision of bp_srs_precompute made testable without filling a
// 288-GB device (a test that hogs memory until ~400 MiB are left depends on when other processes' frees reach the
// driver).  bytes != 0: bp_srs_precompute on this context decides as if hipMemGetInfo had reported that many free bytes on every
// device (the bytes of the tables the SRS holds now still count as free on top); 0: the real reading again.  Nothing else reads it.
extern "C" int bpx_set_free_bytes_probe(bp_ctx* ctx, uint64_t bytes) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  ctx->free_bytes_probe = bytes;
  return BP_OK;
}

int bp_srs_table_info(bp_ctx* ctx, uint64_t srs_handle, uint32_t* window_bits, uint32_t* windows, uint64_t* bytes) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  SrsEntry* lead;
  BP_TRY(srs_find(ctx, srs_handle, &lead));
  if (window_bits) *window_bits = lead->table_c;
  if (windows) *windows = lead->table_W;
  if (bytes) {
    *bytes = 0;
    const std::vector<bp_ctx*> sh = shards_of(ctx);
    for (size_t r = 0; r < sh.size(); r++) {
      SrsEntry* e;
      BP_TRY(lift(ctx, sh[r], srs_find(sh[r], member_handle(*lead, srs_handle, r), &e)));
      if (e->d_table) *bytes += (uint64_t)e->table_W * e->n * sizeof(g1_affine28);
    }
  }
  return BP_OK;
}

// ---------------------------------------------------------------------------------------------- MSM
// Enqueue the MSM of one shard: scalars[0..n) against the member's points [local_first, local_first + n).
//   where 0: `scalars` is host memory; 1: HBM of this member's device; 2: HBM of device src_device (the leader's): copied
//   GPU to GPU into the member's workspace once the leader's stream has reached `ready`.
static int msm_shard_launch(bp_ctx* m, SrsEntry* e, size_t local_first, const void* scalars, size_t n, int fmt, int where, int src_device,
                            hipEvent_t ready, int slot, void* d_blob, MsmPending* pend) {
  DeviceGuard guard(m->device);
  const fr_t* d_scalars = (const fr_t*)scalars;
  if (where != 1 && n) {
    fr_t* d;
    BP_TRY(ws_get(m, "io.scalars", n * sizeof(fr_t), (void**)&d));
    if (where == 0) {
      BP_HIP(m, hipEventRecord(m->ev[4], m->stream));            // upload = ev[4] .. ev[0] (msm_launch records ev[0] first thing)
      BP_HIP(m, hipMemcpyAsync(d, scalars, n * sizeof(fr_t), hipMemcpyHostToDevice, m->stream));
    } else {
      BP_HIP(m, hipStreamWaitEvent(m->stream, ready, 0));
      if (!peer_path(m, src_device)) BP_HIP(m, hipMemcpyAsync(d, scalars, n * sizeof(fr_t), hipMemcpyDeviceToDevice, m->stream));
      else BP_HIP(m, hipMemcpyPeerAsync(d, m->device, scalars, src_device, n * sizeof(fr_t), m->stream));
    }
    d_scalars = d;
  }
  // fixed-base tables pay once the bucket adds outweigh the fixed 2^table_c reduction
  const bool tables = e->d_table && 8 * (uint64_t)n >= (1ull << ((e->table_c & MSM_NAF_FLAG) ? (e->table_c & 0xffu) - 2 : e->table_c));
  if (tables) return msm_launch(m, e->d_table + local_first, n, d_scalars, fmt, e->table_c, e->n, slot, d_blob, pend);
  return msm_launch(m, e->d_points28 + local_first, n, d_scalars, fmt, 0, 0, slot, d_blob, pend);
}

// sum_{i < n} s_i P_{first + i} over every shard of the SRS, in two steps so that several such sums can be in flight:
// launch enqueues every shard's whole pipeline (result slot `slot` of each member), finish waits and adds the partial sums.
struct ShardedPending {
  std::vector<MsmPending> pend;
  std::vector<bool> used;
  bool host_scalars = false;
};
static int msm_all_shards_launch(bp_ctx* ctx, uint64_t srs_handle, size_t first, const void* scalars, size_t n_scalars, int scalar_fmt,
                                 int scalars_on_device, int slot, ShardedPending* sp) {
  SrsEntry* lead;
  BP_TRY(srs_find(ctx, srs_handle, &lead));
  if (first > lead->n_global) return fail(ctx, BP_ERR_INVALID_ARG, "SRS offset out of bounds", hipSuccess, __FILE__, __LINE__);
  const size_t n = std::min(n_scalars, lead->n_global - first);          // zip() truncation, msm.rs:29
  const std::vector<bp_ctx*> sh = shards_of(ctx);
  const std::vector<uint64_t> hs = lead->member_handle;
  if (sh.size() > 1 && scalars_on_device) {            // the members' copies must see what the leader's stream has produced
    DeviceGuard guard(ctx->device);
    BP_HIP(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
  }
  sp->pend.assign(sh.size(), MsmPending());
  sp->used.assign(sh.size(), false);
  sp->host_scalars = !scalars_on_device;
  // every shard's range, then the launches: scalars already in HBM are enqueued by this thread (asynchronous copies and kernels);
  // host scalars go through the members' own threads, so that the uploads -- staged by the issuing thread when the memory is
  // pageable, as a Rust Vec<Scalar> is -- run on all PCIe links at once instead of one after another
  struct Part { SrsEntry* e; size_t local_first, cnt; const uint8_t* sc; };
  std::vector<Part> part(sh.size(), Part{nullptr, 0, 0, nullptr});
  std::vector<int> rcs(sh.size(), BP_OK);
  std::vector<bool> take(sh.size(), false);
  for (size_t r = 0; r < sh.size(); r++) {
    SrsEntry* e;
    BP_TRY(lift(ctx, sh[r], srs_find(sh[r], hs.empty() ? srs_handle : hs[r], &e)));
    const size_t lo = std::max(first, e->first), hi = std::min(first + n, e->first + e->n);
    if (lo >= hi && !(sh.size() == 1)) continue;
    part[r] = Part{e, lo < hi ? lo - e->first : 0, lo < hi ? hi - lo : 0, (const uint8_t*)scalars + (lo < hi ? (lo - first) * sizeof(fr_t) : 0)};
    take[r] = true;
  }
  auto launch_one = [&](size_t r) {
    const int where = !scalars_on_device ? 0 : (r == 0 ? 1 : 2);
    rcs[r] = msm_shard_launch(sh[r], part[r].e, part[r].local_first, part[r].sc, part[r].cnt, scalar_fmt, where, ctx->device, ctx->ev[4], slot, nullptr,
                              &sp->pend[r]);
  };
  if (!scalars_on_device && sh.size() > 1) {
    over_members(ctx, sh.size(), [&](size_t r) { return (bool)take[r]; }, launch_one);
  } else {
    for (size_t r = 0; r < sh.size(); r++)
      if (take[r]) launch_one(r);
  }
  int rc = BP_OK;
  for (size_t r = 0; r < sh.size(); r++) {
    if (!take[r]) continue;
    sp->used[r] = rcs[r] == BP_OK;             // a shard that failed to launch has nothing to wait for
    const int rc1 = lift(ctx, sh[r], rcs[r]);
    if (rc == BP_OK) rc = rc1;
  }
  return rc;               // the caller still finishes whatever was launched
}
static int msm_all_shards_finish(bp_ctx* ctx, const ShardedPending& sp, int rc, g1_proj* out) {
  const std::vector<bp_ctx*> sh = shards_of(ctx);
  const std::vector<MsmPending>& pend = sp.pend;
  const std::vector<bool>& used = sp.used;
  if (pend.size() != sh.size()) return rc != BP_OK ? rc : BP_ERR_INVALID_ARG;      // nothing was launched (bad handle / offset)
  // every launched shard is waited for, also after a failure elsewhere.  In a group the waits and the host epilogues (window
  // sums -> Horner, ~0.1 ms each) run on the members' own threads: eight in sequence would cost more than the shards' GPU time
  // of a 2^20-point MSM split eight ways.
  std::vector<g1_proj> part(sh.size());
  std::vector<int> rcs(sh.size(), BP_OK);
  over_members(ctx, sh.size(), [&](size_t r) { return (bool)used[r]; }, [&](size_t r) {
    DeviceGuard guard(sh[r]->device);
    rcs[r] = msm_finish(sh[r], pend[r], &part[r]);
    sh[r]->shard_accumulate_ms = sh[r]->msm_accumulate_ms;
    sh[r]->shard_total_ms = sh[r]->msm_total_ms;
    sh[r]->shard_adds = sh[r]->msm_adds;
    sh[r]->msm_upload_ms = 0;
    if (rcs[r] == BP_OK && sp.host_scalars && !pend[r].empty && hipEventElapsedTime(&sh[r]->msm_upload_ms, sh[r]->ev[4], sh[r]->ev[0]) != hipSuccess) {
      (void)hipGetLastError();
      sh[r]->msm_upload_ms = 0;
    }
  });
  g1_proj acc = g1_identity();
  float acc_ms = 0, dev_ms = 0;
  uint64_t adds = 0;
  for (size_t r = 0; r < sh.size(); r++) {
    if (!used[r]) continue;
    const int rc1 = lift(ctx, sh[r], rcs[r]);
    if (rc1 != BP_OK) {
      if (rc == BP_OK) rc = rc1;
      continue;
    }
    if (sh.size() == 1) acc = part[r]; else g1_add(acc, acc, part[r]);
    acc_ms = std::max(acc_ms, sh[r]->msm_accumulate_ms);
    dev_ms = std::max(dev_ms, sh[r]->msm_total_ms);
    adds += sh[r]->msm_adds;
  }
  if (rc != BP_OK) return rc;
  if (sh.size() > 1) {                                   // stats of a group: the slowest shard, all additions
    ctx->msm_accumulate_ms = acc_ms;
    ctx->msm_total_ms = dev_ms;
    ctx->msm_adds = adds;
  }
  *out = acc;
  return BP_OK;
}
static int msm_all_shards(bp_ctx* ctx, uint64_t srs_handle, size_t first, const void* scalars, size_t n_scalars, int scalar_fmt,
                          int scalars_on_device, g1_proj* out) {
  ShardedPending sp;
  const int rc = msm_all_shards_launch(ctx, srs_handle, first, scalars, n_scalars, scalar_fmt, scalars_on_device, 0, &sp);
  return msm_all_shards_finish(ctx, sp, rc, out);
}


namespace bp {
constexpr int MAX_LANES = 3;
constexpr int MSM_BATCH_MAX = 4;          // = MSM_MAX_BATCH of msm_kernels.hpp (scalar vectors in one pipeline)

// k commitments of HBM-resident coefficient vectors (on the leader) over the members of a group: per batch of up to MSM_BATCH_MAX
// vectors every member receives its slice of each (peer copies behind the leader's event), runs ONE pipeline over the slices'
// bucket sets and delivers one partial sum per vector; the host adds the members' partial sums.  BP_ERR_TOO_LARGE before anything
// was launched = a member's batch does not fit one pipeline.
static int commit_many_group_batched(bp_ctx* ctx, uint64_t srs_handle, SrsEntry* lead, const fr_t* const* d_coeffs, const size_t* n, int k, g1_proj* out) {
  const std::vector<bp_ctx*> sh = ctx->members;
  const std::vector<uint64_t> hs = lead->member_handle;
  const size_t R = sh.size();
  std::vector<SrsEntry*> ent(R, nullptr);
  for (size_t r = 0; r < R; r++) {
    uint64_t h = hs.empty() ? srs_handle : hs[r];
    auto it = sh[r]->srs.find(h);
    if (it == sh[r]->srs.end()) return fail(ctx, BP_ERR_INVALID_ARG, "unknown SRS handle", hipSuccess, __FILE__, __LINE__);
    ent[r] = &it->second;
    if (!ent[r]->d_table) return BP_ERR_TOO_LARGE;
  }
  for (int base = 0; base < k; base += MSM_BATCH_MAX) {
    const int cnt = std::min(MSM_BATCH_MAX, k - base);
    {
      DeviceGuard guard(ctx->device);
      BP_HIP(ctx, hipEventRecord(ctx->ev[4], ctx->stream));          // the coefficient vectors were produced on the leader's stream
    }
    std::vector<MsmPending> pend(R);
    std::vector<int> rcs(R, BP_OK);
    std::vector<bool> used(R, false);
    int rc = BP_OK;
    for (size_t r = 0; r < R && rc == BP_OK; r++) {
      bp_ctx* m = sh[r];
      SrsEntry* e = ent[r];
      DeviceGuard guard(m->device);
      const fr_t* ptrs[MSM_BATCH_MAX];
      size_t lens[MSM_BATCH_MAX], total = 0;
      for (int j = 0; j < cnt; j++) {
        const size_t nj = std::min(n[base + j], lead->n_global);        // zip() truncation, msm.rs:29
        lens[j] = nj > e->first ? std::min(nj - e->first, e->n) : 0;
        total += lens[j];
      }
      if (total == 0) continue;
      if (r == 0) {
        for (int j = 0; j < cnt; j++) ptrs[j] = d_coeffs[base + j] + e->first;
      } else {
        fr_t* d;
        rc = lift(ctx, m, ws_get(m, "io.scalars", total * sizeof(fr_t), (void**)&d));
        if (rc != BP_OK) break;
        hipError_t he = hipStreamWaitEvent(m->stream, ctx->ev[4], 0);
        size_t at = 0;
        for (int j = 0; j < cnt && he == hipSuccess; j++) {
          if (lens[j]) {
            const fr_t* src = d_coeffs[base + j] + e->first;
            if (!peer_path(m, ctx->device)) he = hipMemcpyAsync(d + at, src, lens[j] * sizeof(fr_t), hipMemcpyDeviceToDevice, m->stream);
            else he = hipMemcpyPeerAsync(d + at, m->device, src, ctx->device, lens[j] * sizeof(fr_t), m->stream);
          }
          ptrs[j] = d + at;
          at += lens[j];
        }
        if (he != hipSuccess) { rc = fail(ctx, BP_ERR_HIP, "commit batch: scalar slices", he, __FILE__, __LINE__); break; }
      }
      // (the tables are used whatever the slice lengths: skipping them for very short slices is a speed heuristic of the single path)
      rcs[r] = msm_launch_many(m, e->d_table, (uint32_t)cnt, ptrs, lens, BP_FR_MONT, e->table_c, e->n, 0, nullptr, &pend[r]);
      if (rcs[r] == BP_ERR_TOO_LARGE && r == 0 && base == 0) return BP_ERR_TOO_LARGE;
      used[r] = rcs[r] == BP_OK;
      rc = lift(ctx, m, rcs[r]);
    }
    std::vector<g1_proj> part(R * MSM_BATCH_MAX);
    over_members(ctx, R, [&](size_t r) { return (bool)used[r]; }, [&](size_t r) {
      DeviceGuard guard(sh[r]->device);
      rcs[r] = msm_finish(sh[r], pend[r], &part[r * MSM_BATCH_MAX]);
    });
    for (size_t r = 0; r < R; r++)
      if (used[r] && rc == BP_OK) rc = lift(ctx, sh[r], rcs[r]);
    if (rc != BP_OK) return rc;
    for (int j = 0; j < cnt; j++) {
      g1_proj acc = g1_identity();
      for (size_t r = 0; r < R; r++)
        if (used[r]) g1_add(acc, acc, part[r * MSM_BATCH_MAX + j]);
      out[base + j] = acc;
    }
  }
  return BP_OK;
}

int commit_many(bp_ctx* ctx, uint64_t srs_handle, const fr_t* const* d_coeffs, const size_t* n, int k, g1_proj* out) {
  if (k <= 0) return BP_OK;
  SrsEntry* e;
  BP_TRY(srs_find(ctx, srs_handle, &e));
  if (is_group(ctx) && k > 1 && e->d_table) {   // group: ONE pipeline per member over its slices of up to MSM_BATCH_MAX commitments
    const char* v = knob("BP_COMMIT_BATCH");
    if (!(v && *v == '0')) {
      int rc = commit_many_group_batched(ctx, srs_handle, e, d_coeffs, n, k, out);
      if (rc != BP_ERR_TOO_LARGE) return rc;       // too long for one pipeline somewhere: queue the commitments one by one below
    }
  }
  if (is_group(ctx) || k == 1) {            // group: every member queues its shards of up to MSM_SLOTS commitments back to back
    for (int base = 0; base < k; base += MSM_SLOTS) {
      const int cnt = std::min((int)MSM_SLOTS, k - base);
      ShardedPending sp[MSM_SLOTS];
      int rcs[MSM_SLOTS], rc = BP_OK;
      for (int j = 0; j < cnt; j++) rcs[j] = rc == BP_OK ? (rc = msm_all_shards_launch(ctx, srs_handle, 0, d_coeffs[base + j], n[base + j], BP_FR_MONT, 1, j, &sp[j])) : BP_OK;
      for (int j = 0; j < cnt; j++) {
        if (sp[j].pend.empty()) continue;
        const int rc1 = msm_all_shards_finish(ctx, sp[j], rcs[j], &out[base + j]);
        if (rc == BP_OK) rc = rc1;
      }
      if (rc != BP_OK) return rc;
    }
    return BP_OK;
  }
  DeviceGuard guard(ctx->device);
  // One pipeline over the bucket sets of up to MSM_MAX_BATCH commitments (msm_launch_many): one sort keyed (polynomial, bucket),
  // one accumulation, one fix-up, one tree over J x 2^(c-1) buckets -- the latency-bound tail is paid once per round instead of
  // once per commitment.  Needs the SRS's fixed-base tables and a batch short enough for the partition sort.  On ONE device the
  // concurrent lanes below already hide the tails of two commitments under the accumulation of the third, and the batch's
  // three-fold sort is exposed: measured 35.6 ms per 2^20-gate proof against 34.4-35.4 with lanes (profiles/r03_commit_batch_ab.txt),
  // so the batch runs only on request there (BP_COMMIT_BATCH=1).  The members of a group context, whose shards are short and whose
  // pipelines share one stream each, use it by default (above).
  {
    const char* v = knob("BP_COMMIT_BATCH");
    size_t n_max = 0;
    for (int j = 0; j < k; j++) n_max = std::max(n_max, std::min(n[j], e->n));
    const bool tables = e->d_table && 8 * (uint64_t)n_max >= (1ull << ((e->table_c & MSM_NAF_FLAG) ? (e->table_c & 0xffu) - 2 : e->table_c));
    if (tables && v && *v == '1') {
      bool ok = true;
      for (int base = 0; base < k && ok; base += (int)MSM_BATCH_MAX) {
        const int cnt = std::min((int)MSM_BATCH_MAX, k - base);
        const fr_t* ptrs[MSM_BATCH_MAX];
        size_t lens[MSM_BATCH_MAX];
        for (int j = 0; j < cnt; j++) {
          ptrs[j] = d_coeffs[base + j];
          lens[j] = std::min(n[base + j], e->n);                        // zip() truncation, msm.rs:29
        }
        MsmPending pend;
        int rc = msm_launch_many(ctx, e->d_table, (uint32_t)cnt, ptrs, lens, BP_FR_MONT, e->table_c, e->n, 0, nullptr, &pend);
        if (rc == BP_ERR_TOO_LARGE && base == 0) {                      // too long for one pipeline: the lanes below
          ok = false;
          break;
        }
        if (rc != BP_OK) return rc;
        BP_TRY(msm_finish(ctx, pend, &out[base]));
      }
      if (ok) return BP_OK;
    }
  }
  while ((int)ctx->lanes.size() < MAX_LANES - 1 && (int)ctx->lanes.size() < k - 1) {
    bp_ctx* lane = nullptr;
    int rc = ctx_create(&lane, ctx->device);
    if (rc != BP_OK) return fail(ctx, rc, "commit lane", hipSuccess, __FILE__, __LINE__);
    ctx->lanes.push_back(lane);
  }
  for (int base = 0; base < k; base += MAX_LANES) {
    const int cnt = std::min(MAX_LANES, k - base);
    BP_HIP(ctx, hipEventRecord(ctx->ev[4], ctx->stream));          // the coefficient vectors were produced on ctx->stream
    MsmPending pend[MAX_LANES];
    bool used[MAX_LANES] = {false, false, false};
    int rc = BP_OK;
    for (int j = 0; j < cnt && rc == BP_OK; j++) {
      bp_ctx* lane = j == 0 ? ctx : ctx->lanes[j - 1];
      if (lane != ctx) {
        hipError_t he = hipStreamWaitEvent(lane->stream, ctx->ev[4], 0);
        if (he != hipSuccess) { rc = fail(ctx, BP_ERR_HIP, "commit lane wait", he, __FILE__, __LINE__); break; }
      }
      const size_t cnt_j = std::min(n[base + j], e->n);              // zip() truncation, msm.rs:29
      rc = lift(ctx, lane, msm_shard_launch(lane, e, 0, d_coeffs[base + j], cnt_j, BP_FR_MONT, 1, ctx->device, nullptr, 0, nullptr, &pend[j]));
      used[j] = rc == BP_OK;
    }
    for (int j = 0; j < cnt; j++) {                                 // every launched lane is waited for, also after a failure elsewhere
      if (!used[j]) continue;
      bp_ctx* lane = j == 0 ? ctx : ctx->lanes[j - 1];
      const int rc1 = lift(ctx, lane, msm_finish(lane, pend[j], &out[base + j]));
      if (rc == BP_OK) rc = rc1;
      if (lane != ctx && rc1 == BP_OK && lane->msm_accumulate_ms > ctx->msm_accumulate_ms) ctx->msm_accumulate_ms = lane->msm_accumulate_ms;
    }
    if (rc != BP_OK) return rc;
  }
  return BP_OK;
}

int commit_lane_launch(bp_ctx* ctx, int j, uint64_t srs_handle, const fr_t* d_coeffs, size_t n, hipEvent_t ready, MsmPending* pend) {
  *pend = MsmPending();
  if (j < 0 || j >= MAX_LANES || is_group(ctx)) return fail(ctx, BP_ERR_INVALID_ARG, "commit lane", hipSuccess, __FILE__, __LINE__);
  SrsEntry* e;
  BP_TRY(srs_find(ctx, srs_handle, &e));
  DeviceGuard guard(ctx->device);
  while ((int)ctx->lanes.size() < j) {
    bp_ctx* lane = nullptr;
    int rc = ctx_create(&lane, ctx->device);
    if (rc != BP_OK) return fail(ctx, rc, "commit lane", hipSuccess, __FILE__, __LINE__);
    ctx->lanes.push_back(lane);
  }
  bp_ctx* lane = j == 0 ? ctx : ctx->lanes[j - 1];
  BP_HIP(ctx, hipStreamWaitEvent(lane->stream, ready, 0));
  return lift(ctx, lane, msm_shard_launch(lane, e, 0, d_coeffs, std::min(n, e->n), BP_FR_MONT, 1, ctx->device, nullptr, 0, nullptr, pend));     // zip() truncation, msm.rs:29
}
int commit_lane_finish(bp_ctx* ctx, int j, const MsmPending& pend, g1_proj* out) {
  bp_ctx* lane = j == 0 ? ctx : ctx->lanes[j - 1];
  return lift(ctx, lane, msm_finish(lane, pend, out));
}
}  // namespace bp


int bp_msm_g1_partial(bp_ctx* ctx, uint64_t srs_handle, size_t first, const void* scalars, size_t n_scalars, int scalar_fmt,
                      int scalars_on_device, uint8_t out144[144]) {
  if (!ctx || !out144 || !fmt_ok(scalar_fmt) || (n_scalars && !scalars)) return BP_ERR_INVALID_ARG;
  g1_proj r;
  BP_TRY(msm_all_shards(ctx, srs_handle, first, scalars, n_scalars, scalar_fmt, scalars_on_device, &r));
  memcpy(out144, &r, 144);
  return BP_OK;
}

int bp_msm_g1(bp_ctx* ctx, uint64_t srs_handle, const void* scalars, size_t n_scalars, int scalar_fmt, uint8_t out96[96]) {
  if (!out96) return BP_ERR_INVALID_ARG;
  uint8_t part[144];
  BP_TRY(bp_msm_g1_partial(ctx, srs_handle, 0, scalars, n_scalars, scalar_fmt, 0, part));
  g1_proj r;
  memcpy(&r, part, 144);
  host_encode96(out96, r);
  return BP_OK;
}

// BucketMSM::bucket_msm(points: &[G1Projective], scalars: &[Scalar], ..) (src/msm.rs:76-81) with nothing cached.  From 2^17 pairs the
// operands cross PCIe in pieces on the side context's stream (scalars, points, normalisation, 28-bit copy; the host blocks in the
// pageable copies) while the main stream multiplies the piece before, out of workspaces instead of an SRS entry.  Measured at 2^20
// pairs: the three calls 8.9 ms (upload 2.7 + normalise 1.1 + allocations, multiply 4.6 with the scalars' upload, free 0.4); one piece
// 8.6, two pieces 8.3, three 10.2, four 11.6 -- a multiplication without tables pays ~0.8 ms of sort, running-sum reduction and
// host epilogue per piece whatever its size, so two pieces (BP_SEAM_PIECES: 1..4) are where the overlap still wins.
constexpr int SEAM_PIECES = 4;
static_assert(SEAM_PIECES <= MSM_SLOTS, "one pinned result slot per piece");
int bp_msm_g1_projective144(bp_ctx* ctx, const uint8_t* points144, size_t n_points, const void* scalars, size_t n_scalars, int scalar_fmt,
                            uint8_t out96[96]) {
  if (!ctx || !out96 || !fmt_ok(scalar_fmt) || (n_points && !points144) || (n_scalars && !scalars)) return BP_ERR_INVALID_ARG;
  const size_t n = std::min(n_points, n_scalars);
  if (is_group(ctx) || n < ((size_t)1 << 17)) {
    uint64_t h = 0;
    BP_TRY(bp_srs_load_projective144(ctx, points144, n, &h));
    int rc = bp_msm_g1(ctx, h, scalars, n, scalar_fmt, out96);
    const std::string msg = ctx->last_error;
    (void)bp_srs_free(ctx, h);
    if (rc != BP_OK) ctx->last_error = msg;
    return rc;
  }
  DeviceGuard guard(ctx->device);
  bp_ctx* side;
  BP_TRY(side_ctx_get(ctx, &side));
  for (auto& e : ctx->seam_ev)
    if (!e) BP_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  int pieces = 2;
  {
    const char* v = knob("BP_SEAM_PIECES");
    if (v && *v >= '1' && *v <= '0' + SEAM_PIECES && !v[1]) pieces = *v - '0';
  }
  const size_t piece = (n + pieces - 1) / pieces;
  uint8_t* d_proj;
  g1_affine* d_aff;
  g1_affine28* d_p28;
  fr_t* d_scal;
  BP_TRY(ws_get(ctx, "seam.proj", piece * 144, (void**)&d_proj));
  BP_TRY(ws_get(ctx, "seam.affine", n * sizeof(g1_affine), (void**)&d_aff));
  BP_TRY(ws_get(ctx, "seam.p28", n * sizeof(g1_affine28), (void**)&d_p28));
  BP_TRY(ws_get(ctx, "seam.scalars", n * sizeof(fr_t), (void**)&d_scal));
  BP_HIP(ctx, stream_wait(ctx->stream));                 // the workspaces may still be read by work the caller enqueued before
  MsmPending pend[SEAM_PIECES];
  int launched = 0, rc = BP_OK;
  for (int k = 0; k < pieces && rc == BP_OK; k++) {
    const size_t lo = (size_t)k * piece, cnt = lo < n ? std::min(piece, n - lo) : 0;
    if (!cnt) break;
    hipError_t he = hipMemcpyAsync(d_scal + lo, (const uint8_t*)scalars + lo * 32, cnt * 32, hipMemcpyHostToDevice, side->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(d_proj, points144 + lo * 144, cnt * 144, hipMemcpyHostToDevice, side->stream);
    if (he != hipSuccess) {
      rc = fail(ctx, BP_ERR_HIP, "bucket_msm operands upload", he, __FILE__, __LINE__);
      break;
    }
    rc = srs_from_projective_run(side, (const g1_proj*)d_proj, cnt, d_aff + lo);
    if (rc == BP_OK) rc = srs_to28_into(side, d_aff + lo, cnt, d_p28 + lo);
    if (rc != BP_OK) {
      ctx->last_error = side->last_error;
      break;
    }
    he = hipEventRecord(ctx->seam_ev[k], side->stream);
    if (he == hipSuccess) he = hipStreamWaitEvent(ctx->stream, ctx->seam_ev[k], 0);
    if (he != hipSuccess) {
      rc = fail(ctx, BP_ERR_HIP, "bucket_msm piece order", he, __FILE__, __LINE__);
      break;
    }
    rc = msm_launch(ctx, d_p28 + lo, cnt, d_scal + lo, scalar_fmt, 0, 0, k, nullptr, &pend[k]);
    if (rc == BP_OK) launched = k + 1;
  }
  // every launched piece is finished (waited for) even after an error: the pinned slots and workspaces must be quiet on return
  g1_proj acc = g1_identity();
  uint64_t adds = 0;
  for (int k = 0; k < launched; k++) {
    g1_proj part;
    const int r2 = msm_finish(ctx, pend[k], &part);
    if (r2 != BP_OK && rc == BP_OK) rc = r2;
    if (r2 == BP_OK) {
      g1_add(acc, acc, part);
      adds += pend[k].adds;
    }
  }
  if (rc != BP_OK) {
    (void)stream_wait(side->stream);
    return rc;
  }
  ctx->msm_adds = adds;
  host_encode96(out96, acc);
  return BP_OK;
}

int bp_msm_g1_blob_device(bp_ctx* ctx, uint64_t srs_handle, size_t first, const void* scalars, size_t n_scalars, int scalar_fmt,
                          int scalars_on_device, void* d_blob) {
  if (!ctx || !d_blob || !fmt_ok(scalar_fmt) || (n_scalars && !scalars)) return BP_ERR_INVALID_ARG;
  if (is_group(ctx)) return fail(ctx, BP_ERR_INVALID_ARG, "blob records are the one-process-per-GPU exchange; a bp_init_multi context combines its shards itself", hipSuccess, __FILE__, __LINE__);
  SrsEntry* e;
  BP_TRY(srs_find(ctx, srs_handle, &e));
  if (first > e->n) return fail(ctx, BP_ERR_INVALID_ARG, "SRS offset out of bounds", hipSuccess, __FILE__, __LINE__);
  const size_t n = std::min(n_scalars, e->n - first);
  DeviceGuard guard(ctx->device);
  MsmPending pend;
  BP_TRY(msm_shard_launch(ctx, e, first, scalars, n, scalar_fmt, scalars_on_device ? 1 : 0, ctx->device, nullptr, 0, d_blob, &pend));
  return msm_finish(ctx, pend, nullptr);
}

int bp_msm_g1_blob_device_async(bp_ctx* ctx, uint64_t srs_handle, size_t first, const void* scalars, size_t n_scalars, int scalar_fmt,
                                int scalars_on_device, void* d_blob) {
  if (!ctx || !d_blob || !fmt_ok(scalar_fmt) || (n_scalars && !scalars)) return BP_ERR_INVALID_ARG;
  if (is_group(ctx)) return fail(ctx, BP_ERR_INVALID_ARG, "blob records are the one-process-per-GPU exchange; a bp_init_multi context combines its shards itself", hipSuccess, __FILE__, __LINE__);
  SrsEntry* e;
  BP_TRY(srs_find(ctx, srs_handle, &e));
  if (first > e->n) return fail(ctx, BP_ERR_INVALID_ARG, "SRS offset out of bounds", hipSuccess, __FILE__, __LINE__);
  const size_t n = std::min(n_scalars, e->n - first);
  DeviceGuard guard(ctx->device);
  MsmPending pend;
  BP_TRY(msm_shard_launch(ctx, e, first, scalars, n, scalar_fmt, scalars_on_device ? 1 : 0, ctx->device, nullptr, 0, d_blob, &pend));
  // nothing is waited for: the stats of this MSM are read from its events by bp_msm_last_stats once the stream has passed them
  ctx->msm_async_pending = !pend.empty;
  ctx->msm_c = pend.tables == 2 ? (MSM_NAF_FLAG | (pend.c + 1)) : pend.c;
  ctx->msm_tables = pend.tables != 0;
  ctx->msm_adds = pend.adds;
  if (pend.empty) ctx->msm_accumulate_ms = ctx->msm_total_ms = 0;
  return BP_OK;
}

int bp_msm_blobs_sum_device(bp_ctx* ctx, const void* d_blobs, size_t n_blobs, void* d_out_blob) {
  if (!ctx || !d_blobs || !d_out_blob || n_blobs == 0 || n_blobs > 4096) return BP_ERR_INVALID_ARG;
  DeviceGuard guard(ctx->device);
  return msm_blobs_sum_device_run(ctx, d_blobs, n_blobs, d_out_blob);
}
int bp_msm_blobs_sum_device_async(bp_ctx* ctx, const void* d_blobs, size_t n_blobs, void* d_out_blob) {
  if (!ctx || !d_blobs || !d_out_blob || n_blobs == 0 || n_blobs > 4096) return BP_ERR_INVALID_ARG;
  DeviceGuard guard(ctx->device);
  return msm_blobs_sum_device_run(ctx, d_blobs, n_blobs, d_out_blob, false);
}

int bp_msm_blobs_combine(const void* blobs, size_t n_blobs, uint8_t out96[96]) {
  if (!out96 || (n_blobs && !blobs)) return BP_ERR_INVALID_ARG;
  g1_proj r;
  BP_TRY(msm_blobs_combine((const uint8_t*)blobs, n_blobs, &r));
  host_encode96(out96, r);
  return BP_OK;
}

int bp_msm_window_scalars(const void* scalars, size_t n, int scalar_fmt, size_t b, size_t c, void* out_le32) {
  if (!fmt_ok(scalar_fmt) || (n && (!scalars || !out_le32)) || c == 0 || c > 63) return BP_ERR_INVALID_ARG;
  const size_t k = b / c;
  if (k == 0 || k * c > 256) return BP_ERR_INVALID_ARG;
  const size_t shift = 256 - k * c, words = shift / 32, bits = shift % 32;
  const uint8_t* in = (const uint8_t*)scalars;
  uint8_t* out = (uint8_t*)out_le32;
  for (size_t i = 0; i < n; i++) {
    fr_t v, r;
    memcpy(&v, in + 32 * i, 32);
    if (scalar_fmt == BP_FR_MONT) {
      Fr::from_mont(v, v);                                // Scalar::to_bytes (scalar.rs:292-304)
    } else {
      fr_t t;
      if (!big_sub(t, v, Fr::modulus())) return BP_ERR_BAD_SCALAR;
    }
    for (size_t j = 0; j < 8; j++) {
      const uint64_t lo = j + words < 8 ? v.l[j + words] : 0u, hi = j + words + 1 < 8 ? v.l[j + words + 1] : 0u;
      r.l[j] = (uint32_t)(((hi << 32) | lo) >> bits);
    }
    memcpy(out + 32 * i, &r, 32);
  }
  return BP_OK;
}

int bp_g1_sum_partials(const uint8_t* partials144, size_t n, uint8_t out96[96]) {
  if (!out96 || (n && !partials144)) return BP_ERR_INVALID_ARG;
  g1_proj acc = g1_identity();
  for (size_t i = 0; i < n; i++) {
    g1_proj p;
    memcpy(&p, partials144 + 144 * i, 144);
    g1_add(acc, acc, p);
  }
  host_encode96(out96, acc);
  return BP_OK;
}
int bp_g1_partial_to_bytes96(const uint8_t in144[144], uint8_t out96[96]) { return bp_g1_sum_partials(in144, 1, out96); }
int bp_g1_bytes96_to_partial(const uint8_t in96[96], uint8_t out144[144]) {
  if (!in96 || !out144) return BP_ERR_INVALID_ARG;
  g1_proj p;
  if (!host_decode96(p, in96)) return BP_ERR_BAD_POINT;
  memcpy(out144, &p, 144);
  return BP_OK;
}

int bp_g1_bytes96_to_compressed48(const uint8_t in96[96], uint8_t out48[48]) {
  if (!in96 || !out48) return BP_ERR_INVALID_ARG;
  g1_proj p;
  if (!host_decode96(p, in96)) return BP_ERR_BAD_POINT;
  host_compress48(out48, p);
  return BP_OK;
}
int bp_msm_last_member_stats(bp_ctx* ctx, int member, float* upload_ms, float* accumulate_ms, float* total_device_ms, uint64_t* mixed_adds) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  const std::vector<bp_ctx*> sh = shards_of(ctx);
  if (member < 0 || (size_t)member >= sh.size()) return BP_ERR_INVALID_ARG;
  const bp_ctx* m = sh[member];
  if (upload_ms) *upload_ms = m->msm_upload_ms;
  const bool group = sh.size() > 1;
  if (accumulate_ms) *accumulate_ms = group ? m->shard_accumulate_ms : m->msm_accumulate_ms;
  if (total_device_ms) *total_device_ms = group ? m->shard_total_ms : m->msm_total_ms;
  if (mixed_adds) *mixed_adds = group ? m->shard_adds : m->msm_adds;
  return BP_OK;
}
int bp_msm_last_used_tables(bp_ctx* ctx) { return ctx ? (ctx->msm_tables ? 1 : 0) : BP_ERR_INVALID_ARG; }
int bp_msm_last_stats(bp_ctx* ctx, float* accumulate_ms, float* total_device_ms, uint64_t* mixed_adds, uint32_t* window_bits) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  if (ctx->msm_async_pending) {        // the last MSM was only enqueued: its events are read now (0 while it is still running)
    DeviceGuard guard(ctx->device);
    float a = 0, t = 0;
    if (hipEventElapsedTime(&a, ctx->ev[1], ctx->ev[2]) == hipSuccess && hipEventElapsedTime(&t, ctx->ev[0], ctx->ev[3]) == hipSuccess) {
      ctx->msm_accumulate_ms = a;
      ctx->msm_total_ms = t;
      ctx->msm_async_pending = false;
    } else {
      (void)hipGetLastError();
      ctx->msm_accumulate_ms = ctx->msm_total_ms = 0;
    }
  }
  if (accumulate_ms) *accumulate_ms = ctx->msm_accumulate_ms;
  if (total_device_ms) *total_device_ms = ctx->msm_total_ms;
  if (mixed_adds) *mixed_adds = ctx->msm_adds;
  if (window_bits) *window_bits = ctx->msm_c;
  return BP_OK;
}
