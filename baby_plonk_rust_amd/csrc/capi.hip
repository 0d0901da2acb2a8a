// capi.hip -- the C ABI of include/bp_msm_ntt.h: context management, host<->HBM staging, the O(1)/O(W)
// host epilogues (Horner over window sums, affine normalisation, wire encodings).  All O(N) work runs in
// the HIP kernels of msm.hip / ntt.hip / poly.hip / srs.hip; there is no CPU fallback for it.
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

using namespace bp;

namespace bp {

int fail(bp_ctx* ctx, int code, const char* what, hipError_t e, const char* file, int line) {
  if (ctx) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s%s%s (%s:%d)", what, e != hipSuccess ? ": " : "", e != hipSuccess ? hipGetErrorString(e) : "", file,
             line);
    ctx->last_error = buf;
  }
  return code;
}

static thread_local bool tl_member_worker = false;     // set by the persistent member threads of a group context

hipError_t stream_wait(hipStream_t st) {
  static const bool block = [] { const char* v = knob("BP_WAIT_BLOCK"); return v && *v == '1'; }();
  if (!block) {
    // the calling thread polls for up to 6 ms (a blocked hipStreamSynchronize wakes up ~20 us late, and a call waits several times);
    // a member's worker thread polls for 50 us only and then blocks: N members must not spin N host cores through the GPU phase
    const auto limit = tl_member_worker ? std::chrono::microseconds(50) : std::chrono::microseconds(6000);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; spins++) {
      const hipError_t e = hipStreamQuery(st);
      if (e != hipErrorNotReady) return e;
      __builtin_ia32_pause();
      if ((spins & 15) == 15 && std::chrono::steady_clock::now() - t0 > limit) break;
    }
  }
  return hipStreamSynchronize(st);
}

int ws_get(bp_ctx* ctx, const char* name, size_t bytes, void** out) {
  DevBuf& b = ctx->ws[name];
  if (bytes == 0) bytes = 16;
  if (b.cap < bytes) {
    if (b.p) {
      BP_HIP(ctx, stream_wait(ctx->stream));
      BP_HIP(ctx, hipFree(b.p));
      b.p = nullptr;
      b.cap = 0;
    }
    size_t cap = (bytes + 255) & ~(size_t)255;
    BP_HIP(ctx, hipMalloc(&b.p, cap));
    b.cap = cap;
  }
  *out = b.p;
  return BP_OK;
}

int pinned_get(bp_ctx* ctx, size_t bytes, void** out) {
  if (ctx->pinned_cap < bytes) {
    if (ctx->pinned) BP_HIP(ctx, hipHostFree(ctx->pinned));
    ctx->pinned = nullptr;
    ctx->pinned_cap = 0;
    size_t cap = std::max<size_t>(bytes, 64 * 1024);
    BP_HIP(ctx, hipHostMalloc(&ctx->pinned, cap, hipHostMallocDefault));
    ctx->pinned_cap = cap;
  }
  *out = ctx->pinned;
  return BP_OK;
}

// result = sum_w 2^(c*w) * T_w, most significant window first (the reference's combine, msm.rs:107-115)
void host_horner(g1_proj& out, const g1_proj* window_sums, uint32_t W, uint32_t c) {
  g1_proj acc = window_sums[W - 1];
  for (uint32_t w = W - 1; w-- > 0;) {
    for (uint32_t d = 0; d < c; d++) g1_double(acc, acc);
    g1_add(acc, acc, window_sums[w]);
  }
  out = acc;
}

// planes[w * c + 0] = A_w, planes[w * c + 1 + j] = T_{w,j} (j < c - 1):
//   out = sum_w 2^(c w) (A_w + sum_j 2^j T_{w,j}),  one pass from the top bit position down (bucket b holds digit b + 1);
//   odd_digits (NAF tables, one window): bucket b holds digit 2b + 1, out = A + 2 sum_j 2^j T_j
void host_plane_horner(g1_proj& out, const g1_proj* planes, uint32_t W, uint32_t c, bool odd_digits) {
  g1_proj acc = g1_identity();
  for (uint32_t w = W; w-- > 0;) {
    const g1_proj* p = planes + (size_t)w * c;
    for (uint32_t j = c; j-- > 0;) {
      g1_double(acc, acc);
      if (j + 1 < c) g1_add(acc, acc, p[1 + j]);           // bit position c - 1 of the window carries no plane
    }
    if (odd_digits) g1_double(acc, acc);
    g1_add(acc, acc, p[0]);
  }
  out = acc;
}

static void fp_to_be48_host(uint8_t* b, const fp_t& a) {
  for (int i = 0; i < 12; i++) {
    uint8_t* p = b + 4 * (11 - i);
    p[0] = (uint8_t)(a.l[i] >> 24); p[1] = (uint8_t)(a.l[i] >> 16); p[2] = (uint8_t)(a.l[i] >> 8); p[3] = (uint8_t)a.l[i];
  }
}
static fp_t fp_from_be48_host(const uint8_t* b) {
  fp_t r;
  for (int i = 0; i < 12; i++) {
    const uint8_t* p = b + 4 * (11 - i);
    r.l[i] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
  }
  return r;
}
// G1Affine::from(p).to_uncompressed()  (g1.rs:49-63, 246-260)
void host_encode96(uint8_t out96[96], const g1_proj& p) {
  memset(out96, 0, 96);
  if (g1_is_identity(p)) {
    out96[0] = 0x40;
    return;
  }
  g1_affine a = g1_to_affine(p);
  fp_t x, y;
  Fp::from_mont(x, a.x);
  Fp::from_mont(y, a.y);
  fp_to_be48_host(out96, x);
  fp_to_be48_host(out96 + 48, y);
}
// G1Affine::from_uncompressed_unchecked (g1.rs:273-322) without the curve check
bool host_decode96(g1_proj& out, const uint8_t in96[96]) {
  uint8_t buf[96];
  memcpy(buf, in96, 96);
  const uint32_t flags = buf[0] >> 5;
  buf[0] &= 0x1f;
  fp_t x = fp_from_be48_host(buf), y = fp_from_be48_host(buf + 48), t;
  if (!big_sub(t, x, Fp::modulus()) || !big_sub(t, y, Fp::modulus())) return false;
  if (flags & 0b101) return false;
  if (flags & 0b010) {
    if (!big_is_zero(x) || !big_is_zero(y)) return false;
    out = g1_identity();
    return true;
  }
  Fp::to_mont(out.x, x);
  Fp::to_mont(out.y, y);
  out.z = Fp::one();
  return true;
}

}  // namespace bp

// ------------------------------------------------------------------------------------------------------
static bool fr_bytes_to_mont(fr_t& out, const uint8_t* b32, int fmt) {
  fr_t v;
  memcpy(&v, b32, 32);
  if (fmt == BP_FR_MONT) {
    out = v;
    return true;
  }
  fr_t t;
  if (!big_sub(t, v, Fr::modulus())) return false;       // >= q: Scalar::from_bytes rejects (scalar.rs:264-288)
  Fr::to_mont(out, v);
  return true;
}
static void fr_mont_to_bytes(uint8_t* b32, const fr_t& v, int fmt) {
  fr_t t = v;
  if (fmt == BP_FR_BYTES_LE) Fr::from_mont(t, v);
  memcpy(b32, &t, 32);
}
static bool fmt_ok(int fmt) { return fmt == BP_FR_BYTES_LE || fmt == BP_FR_MONT; }
static bool basis_ok(int b) { return b == BP_BASIS_LAGRANGE || b == BP_BASIS_MONOMIAL; }

// upload n scalars to workspace `name`, converting to Montgomery form on the device if needed
static int upload_fr(bp_ctx* ctx, const char* name, const void* host, size_t n, size_t cap_elems, int fmt, fr_t** out) {
  fr_t* d;
  BP_TRY(ws_get(ctx, name, std::max(cap_elems, n) * sizeof(fr_t), (void**)&d));
  if (n) BP_HIP(ctx, hipMemcpyAsync(d, host, n * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
  if (fmt == BP_FR_BYTES_LE) BP_TRY(fr_convert_run(ctx, d, n, 0));
  *out = d;
  return BP_OK;
}
static int download_fr(bp_ctx* ctx, fr_t* d, void* host, size_t n, int fmt) {
  if (n == 0) return BP_OK;
  if (fmt == BP_FR_BYTES_LE) BP_TRY(fr_convert_run(ctx, d, n, 1));
  BP_HIP(ctx, hipMemcpyAsync(host, d, n * sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
  BP_HIP(ctx, stream_wait(ctx->stream));
  return BP_OK;
}

// ---- group contexts: one persistent host thread per member ----------------------------------------------------------
// Copies from and to pageable host memory are staged by the thread that issues them, and a stream is waited for by the thread
// that calls hipStreamSynchronize: a single-threaded caller (the reference's Setup::commit, setup.rs:32-37) would serialise the
// members' PCIe transfers and host epilogues.  Each member beyond the first therefore owns a worker thread, parked on a
// condition variable between calls.
namespace bp {
struct MemberWorker {
  std::mutex mu;
  std::condition_variable cv;
  std::function<void()> job;
  bool has_job = false, done = true, stop = false;
  std::thread th;
  MemberWorker() : th([this] { loop(); }) {}
  ~MemberWorker() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    th.join();
  }
  void loop() {
    tl_member_worker = true;
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [&] { return has_job || stop; });
      if (stop) return;
      std::function<void()> j = std::move(job);
      has_job = false;
      lk.unlock();
      j();
      lk.lock();
      done = true;
      cv.notify_all();
    }
  }
  void submit(std::function<void()> j) {
    {
      std::lock_guard<std::mutex> lk(mu);
      job = std::move(j);
      has_job = true;
      done = false;
    }
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done; });
  }
};
}  // namespace bp

// work(r) for every member r for which use(r) holds: member 0 on the calling thread, the others on their own threads, all at once;
// returns when every one has finished.  A plain context (or a call that concerns one member) runs inline.
static void over_members(bp_ctx* ctx, size_t R, const std::function<bool(size_t)>& use, const std::function<void(size_t)>& work) {
  bp_ctx* lead = ctx->leader ? ctx->leader : ctx;
  std::vector<size_t> sent;
  for (size_t r = 1; r < R; r++) {
    if (!use(r)) continue;
    if (r - 1 < lead->workers.size()) {
      lead->workers[r - 1]->submit([&work, r] { work(r); });
      sent.push_back(r);
    } else {
      work(r);
    }
  }
  if (R > 0 && use(0)) work(0);
  for (size_t r : sent) lead->workers[r - 1]->wait();
}
// contiguous point range [lo, hi) of shard r of R over n points; the first n % R shards get one extra point
static void shard_range(size_t n, size_t r, size_t R, size_t* lo, size_t* hi) {
  const size_t base = n / R, extra = n % R;
  *lo = r * base + std::min(r, extra);
  *hi = *lo + base + (r < extra ? 1 : 0);
}
// the single-device contexts an entry point has to visit: the members of a group, or the context itself
static std::vector<bp_ctx*> shards_of(bp_ctx* ctx) {
  if (is_group(ctx)) return ctx->members;
  return std::vector<bp_ctx*>(1, ctx);
}
// a member's failure is reported on the context the caller holds
static int lift(bp_ctx* ctx, bp_ctx* member, int rc) {
  if (rc != BP_OK && member != ctx) ctx->last_error = member->last_error;
  return rc;
}

static int ctx_create(bp_ctx** out, int device_id) {
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device_id < 0 || device_id >= count) return BP_ERR_NO_DEVICE;
  DeviceGuard guard(device_id);
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != device_id) return BP_ERR_NO_DEVICE;
  bp_ctx* ctx = new bp_ctx();
  ctx->device = device_id;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return BP_ERR_NO_DEVICE;
  }
  for (auto& e : ctx->ev)
    if (hipEventCreateWithFlags(&e, hipEventDefault) != hipSuccess) {
      bp_destroy(ctx);
      return BP_ERR_NO_DEVICE;
    }
  int rc = ntt_init_tables(ctx);
  if (rc == BP_OK) rc = msm_init_device(ctx);
  if (rc != BP_OK) {
    fprintf(stderr, "bp_init: %s\n", ctx->last_error.c_str());
    bp_destroy(ctx);
    return rc;
  }
  if (stream_wait(ctx->stream) != hipSuccess) {
    bp_destroy(ctx);
    return BP_ERR_HIP;
  }
  *out = ctx;
  return BP_OK;
}

namespace bp {
int side_ctx_get(bp_ctx* ctx, bp_ctx** out) {
  if (!ctx->side) {
    bp_ctx* sd = nullptr;
    int rc = ctx_create(&sd, ctx->device);
    if (rc != BP_OK) return fail(ctx, rc, "side context", hipSuccess, __FILE__, __LINE__);
    ctx->side = sd;
    DeviceGuard guard(ctx->device);
    for (auto& e : ctx->side_ev)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail(ctx, BP_ERR_HIP, "side events", hipGetLastError(), __FILE__, __LINE__);
  }
  *out = ctx->side;
  return BP_OK;
}
}  // namespace bp

extern "C" {

const char* bp_version(void) { return EXPERIMENT_BUILD ? "bp_msm_ntt 0.3 (gfx950) +experiment" : "bp_msm_ntt 0.3 (gfx950)"; }

int bp_init(bp_ctx** out, int device_id) {
  if (!out) return BP_ERR_INVALID_ARG;
  return ctx_create(out, device_id);
}

int bp_init_multi(bp_ctx** out, const int* device_ids, int n_devices) {
  if (!out || !device_ids || n_devices < 1 || n_devices > 64) return BP_ERR_INVALID_ARG;
  *out = nullptr;
  std::vector<bp_ctx*> m;
  for (int r = 0; r < n_devices; r++) {
    bp_ctx* c = nullptr;
    int rc = ctx_create(&c, device_ids[r]);
    if (rc != BP_OK) {
      for (bp_ctx* p : m) bp_destroy(p);
      return rc;
    }
    m.push_back(c);
  }
  if (n_devices > 1) {
    // peer access lets hipMemcpyPeerAsync go GPU to GPU over xGMI; without it the copies are staged through the host
    for (int a = 0; a < n_devices; a++) {
      DeviceGuard guard(device_ids[a]);
      for (int b = 0; b < n_devices; b++) {
        int can = 0;
        if (device_ids[a] == device_ids[b] || hipDeviceCanAccessPeer(&can, device_ids[a], device_ids[b]) != hipSuccess || !can) continue;
        hipError_t e = hipDeviceEnablePeerAccess(device_ids[b], 0);
        if (e != hipSuccess) (void)hipGetLastError();         // hipErrorPeerAccessAlreadyEnabled included
      }
    }
    m[0]->members = m;
    for (int a = 0; a < n_devices; a++)
      for (int b = a + 1; b < n_devices; b++)
        if (device_ids[a] == device_ids[b]) m[0]->rehearsal = true;
    for (int r = 1; r < n_devices; r++) {
      m[r]->leader = m[0];
      m[0]->workers.push_back(new MemberWorker());
    }
  }
  *out = m[0];
  return BP_OK;
}

int bp_ctx_devices(bp_ctx* ctx, int* device_ids, int cap) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  const std::vector<bp_ctx*> sh = shards_of(ctx);
  for (size_t r = 0; r < sh.size() && device_ids && (int)r < cap; r++) device_ids[r] = sh[r]->device;
  return (int)sh.size();
}

void bp_destroy(bp_ctx* ctx) {
  if (!ctx) return;
  {                                          // circuits first: their coset shares live on the members, which go next
    DeviceGuard guard(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->circuits) circuit_release(kv.second);
    ctx->circuits.clear();
  }
  for (MemberWorker* w : ctx->workers) delete w;
  ctx->workers.clear();
  for (size_t r = 1; r < ctx->members.size(); r++) {
    ctx->members[r]->leader = nullptr;
    bp_destroy(ctx->members[r]);
  }
  ctx->members.clear();
  for (bp_ctx* lane : ctx->lanes) bp_destroy(lane);
  ctx->lanes.clear();
  if (ctx->side) {
    bp_destroy(ctx->side);
    ctx->side = nullptr;
  }
  for (auto& e : ctx->side_ev)
    if (e) {
      DeviceGuard guard(ctx->device);
      (void)hipEventDestroy(e);
      e = nullptr;
    }
  for (auto& e : ctx->seam_ev)
    if (e) {
      DeviceGuard guard(ctx->device);
      (void)hipEventDestroy(e);
      e = nullptr;
    }
  DeviceGuard guard(ctx->device);
  if (ctx->stream) (void)stream_wait(ctx->stream);
  for (auto& kv : ctx->ws)
    if (kv.second.p) (void)hipFree(kv.second.p);
  for (auto& kv : ctx->srs) {
    if (kv.second.d_points) (void)hipFree(kv.second.d_points);
    if (kv.second.d_points28) (void)hipFree(kv.second.d_points28);
    if (kv.second.d_table) (void)hipFree(kv.second.d_table);
  }
  for (auto& kv : ctx->circuits) circuit_release(kv.second);
  for (auto& kv : ctx->ntt_tables) {
    (void)hipFree(kv.second.lo);
    (void)hipFree(kv.second.hi);
    if (kv.second.hi_scaled) (void)hipFree(kv.second.hi_scaled);
    for (auto* f : kv.second.full)
      if (f) (void)hipFree(f);
    if (kv.second.n_inv) (void)hipFree(kv.second.n_inv);
    if (kv.second.n_inv_tw) (void)hipFree(kv.second.n_inv_tw);
  }
  for (auto& t : ctx->small_tw)
    if (t) (void)hipFree(t);
  for (auto& e : ctx->ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* bp_last_error(bp_ctx* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

int bp_set_stream(bp_ctx* ctx, void* hip_stream) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  DeviceGuard guard(ctx->device);
  if (hip_stream != nullptr && !ctx->own_stream && ctx->stream == (hipStream_t)hip_stream) return BP_OK;     // already there: no wait (callers re-assert per call)
  BP_HIP(ctx, stream_wait(ctx->stream));
  if (hip_stream == nullptr) {
    if (!ctx->own_stream) {
      BP_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
      ctx->own_stream = true;
    }
    return BP_OK;
  }
  if (ctx->own_stream) BP_HIP(ctx, hipStreamDestroy(ctx->stream));
  ctx->stream = (hipStream_t)hip_stream;
  ctx->own_stream = false;
  return BP_OK;
}

int bp_synchronize(bp_ctx* ctx) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  for (bp_ctx* m : shards_of(ctx)) {
    DeviceGuard guard(m->device);
    BP_HIP(ctx, stream_wait(m->stream));
  }
  return BP_OK;
}

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
static int srs_find(bp_ctx* ctx, uint64_t handle, SrsEntry** out) {
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
  // The old tables go first (their memory counts as free).  An automatic width that does not fit falls back to wider windows
  // (fewer rows: 22 -> 12, 24 -> 11) and then to NO tables (the MSM runs on the raw points, same bytes out: bp_srs_table_info
  // reports what was built); an explicit width that does not fit is an error that says how much is missing, not an
  // out-of-memory failure halfway through the build.
  for (size_t r = 0; r < sh.size(); r++) BP_TRY(lift(ctx, sh[r], srs_precompute_one(sh[r], hs.empty() ? srs_handle : hs[r], BP_SRS_TABLES_OFF)));
  if (c == BP_SRS_TABLES_OFF) return BP_OK;
  auto fits = [&](uint32_t cc, size_t* need_out, size_t* free_out) -> int {
    for (size_t r = 0; r < sh.size(); r++) {
      SrsEntry* e;
      BP_TRY(lift(ctx, sh[r], srs_find(sh[r], hs.empty() ? srs_handle : hs[r], &e)));
      DeviceGuard guard(sh[r]->device);
      size_t free_b = 0, total_b = 0;
      BP_HIP(ctx, hipMemGetInfo(&free_b, &total_b));
      const size_t rows = srs_table_rows(cc);
      const size_t need = rows * e->n * (sizeof(g1_affine28) + 24) + ((size_t)256 << 20);      // + sort records, lists, partial slots of one MSM
      if (need > free_b) {
        *need_out = need;
        *free_out = free_b;
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
      if (cc <= c) continue;
      rc = fits(cc, &need, &free_b);
      if (rc < 0) return rc;
      if (rc == 0) { pick = cc; break; }
    }
    c = pick;
    if (c == BP_SRS_TABLES_OFF) return BP_OK;
  }
  for (size_t r = 0; r < sh.size(); r++) BP_TRY(lift(ctx, sh[r], srs_precompute_one(sh[r], hs.empty() ? srs_handle : hs[r], c)));
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

}  // extern "C"

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

extern "C" {

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

// ---------------------------------------------------------------------------------------------- DFT
int bp_ntt_fr_device(bp_ctx* ctx, void* d_data, uint32_t log_n, int inverse, size_t batch, size_t stride) {
  if (!ctx || (!d_data && batch)) return BP_ERR_INVALID_ARG;
  if (batch == 0) {                   // nothing was enqueued: no events to read
    ctx->ntt_ms = 0;
    return BP_OK;
  }
  DeviceGuard guard(ctx->device);
  BP_TRY(ntt_run(ctx, (fr_t*)d_data, log_n, inverse, batch, stride));
  BP_HIP(ctx, stream_wait(ctx->stream));
  BP_HIP(ctx, hipEventElapsedTime(&ctx->ntt_ms, ctx->ev[0], ctx->ev[1]));
  ctx->ntt_async_pending = false;
  return BP_OK;
}

// Enqueue only: the transform runs on the context's stream behind whatever was enqueued before; bp_synchronize (or any blocking
// entry point on this context) waits for it.  Back-to-back transforms on HBM-resident data then cost their kernels, not a host
// round trip each (16 us of a 0.17 ms call at 2^20).
int bp_ntt_fr_device_async(bp_ctx* ctx, void* d_data, uint32_t log_n, int inverse, size_t batch, size_t stride) {
  if (!ctx || (!d_data && batch)) return BP_ERR_INVALID_ARG;
  if (batch == 0) return BP_OK;
  DeviceGuard guard(ctx->device);
  BP_TRY(ntt_run(ctx, (fr_t*)d_data, log_n, inverse, batch, stride));
  ctx->ntt_async_pending = true;
  ctx->ntt_members = 1;
  return BP_OK;
}

// Group context, batch > 1: the columns are independent transforms (NTT option i), column j goes to member
// j mod R; every member uploads, transforms and downloads its columns on its own stream, nothing is exchanged.
static int ntt_columns_over_members(bp_ctx* ctx, uint8_t* data, uint32_t log_n, int inverse, int scalar_fmt, size_t batch, size_t stride) {
  const std::vector<bp_ctx*> sh = shards_of(ctx);
  const size_t N = (size_t)1 << log_n, R = sh.size();
  std::vector<fr_t*> dbuf(R, nullptr);
  std::vector<size_t> cnt(R, 0);
  // one host thread per member: the column copies from and to pageable memory are staged by the thread that issues them
  std::vector<int> rcs(R, BP_OK);
  for (size_t r = 0; r < R; r++) cnt[r] = batch / R + (r < batch % R ? 1 : 0);
  over_members(ctx, R, [&](size_t r) { return cnt[r] != 0; }, [&](size_t r) {
    bp_ctx* m = sh[r];
    DeviceGuard guard(m->device);
    int rc = ws_get(m, "io.ntt", cnt[r] * N * sizeof(fr_t), (void**)&dbuf[r]);
    for (size_t j = 0; j < cnt[r] && rc == BP_OK; j++) {
      hipError_t e = hipMemcpyAsync(dbuf[r] + j * N, data + (r + j * R) * stride * sizeof(fr_t), N * sizeof(fr_t), hipMemcpyHostToDevice, m->stream);
      if (e != hipSuccess) rc = fail(m, BP_ERR_HIP, "NTT column upload", e, __FILE__, __LINE__);
    }
    if (rc == BP_OK && scalar_fmt == BP_FR_BYTES_LE) rc = fr_convert_run(m, dbuf[r], cnt[r] * N, 0);
    if (rc == BP_OK) rc = ntt_run(m, dbuf[r], log_n, inverse, cnt[r], N);
    if (rc == BP_OK && scalar_fmt == BP_FR_BYTES_LE) rc = fr_convert_run(m, dbuf[r], cnt[r] * N, 1);
    for (size_t j = 0; j < cnt[r] && rc == BP_OK; j++) {
      hipError_t e = hipMemcpyAsync(data + (r + j * R) * stride * sizeof(fr_t), dbuf[r] + j * N, N * sizeof(fr_t), hipMemcpyDeviceToHost, m->stream);
      if (e != hipSuccess) rc = fail(m, BP_ERR_HIP, "NTT column download", e, __FILE__, __LINE__);
    }
    rcs[r] = rc;
  });
  int rc = BP_OK;
  for (size_t r = 0; r < R; r++)
    if (rc == BP_OK) rc = lift(ctx, sh[r], rcs[r]);
  float ms = 0;
  for (size_t r = 0; r < R; r++) {                       // wait for every member, also after a failure elsewhere
    if (cnt[r] == 0) continue;
    DeviceGuard guard(sh[r]->device);
    hipError_t e = stream_wait(sh[r]->stream);
    if (e != hipSuccess && rc == BP_OK) rc = fail(ctx, BP_ERR_HIP, "NTT columns", e, __FILE__, __LINE__);
    float t = 0;
    if (rc == BP_OK && hipEventElapsedTime(&t, sh[r]->ev[0], sh[r]->ev[1]) == hipSuccess) ms = std::max(ms, t);
  }
  ctx->ntt_ms = ms;
  ctx->ntt_passes = sh[0]->ntt_passes;
  ctx->ntt_members = (uint32_t)std::min(R, batch);
  return rc;
}

// Group context, ONE large transform from host memory (NTT option ii): N = 2^(l_1 + s).  Member g uploads the
// columns r in its slice of [0, 2^s) -- every member over its own PCIe link -- and runs pass 1 on them; the members then swap
// blocks of the intermediate buffer (member g' collects the rows e_1 of its slice: R - 1 peer copies of N / R^2 elements each,
// over xGMI), run the remaining passes on their e_1 and download their outputs (index = e_1 mod 2^(l_1): runs of 2^(l_1) / R
// elements).  Buffers keep the full N-element layout on every member, so the kernels address exactly as on one GPU.
// Returns 1 (not an error code of the ABI) when the shape does not split; only then does the caller run the transform on the
// leader.  Any other failure is returned as it is: the download phase writes `data` from every member at once, so after a
// failed copy the buffer may be part input, part output, and must not be transformed again.
static int ntt_one_over_members(bp_ctx* ctx, uint8_t* data, uint32_t log_n, int inverse, int scalar_fmt) {
  const std::vector<bp_ctx*> sh = shards_of(ctx);
  const uint32_t R = (uint32_t)sh.size();
  if (!ntt_split_ok(log_n, R)) return 1;
  uint32_t l1 = 0;
  ntt_split_shape(log_n, &l1);
  const size_t N = (size_t)1 << log_n, S = (size_t)1 << (log_n - l1), L1 = (size_t)1 << l1;     // rows of pass 1 x row length
  const size_t cols = S / R, rows = L1 / R, esz = sizeof(fr_t);
  std::vector<fr_t*> dbuf(R, nullptr), tbuf(R, nullptr);
  std::vector<int> rcs(R, BP_OK);
  for (uint32_t g = 0; g < R; g++) {
    DeviceGuard guard(sh[g]->device);
    BP_TRY(lift(ctx, sh[g], ws_get(sh[g], "io.ntt", N * esz, (void**)&dbuf[g])));
    BP_TRY(lift(ctx, sh[g], ntt_tmp_buffer(sh[g], log_n, &tbuf[g])));
  }
  // copies from and to pageable host memory are staged by the calling thread: one host thread per member keeps every PCIe link busy
  auto on_members = [&](const std::function<int(uint32_t)>& work) {
    over_members(ctx, R, [](size_t) { return true; }, [&](size_t g) { rcs[g] = work((uint32_t)g); });
    for (uint32_t g = 0; g < R; g++)
      if (rcs[g] != BP_OK) return lift(ctx, sh[g], rcs[g]);
    return (int)BP_OK;
  };
  int rc = on_members([&](uint32_t g) -> int {                 // column slice up, pass 1
    bp_ctx* m = sh[g];
    DeviceGuard guard(m->device);
    hipError_t e = hipMemcpy2DAsync(dbuf[g] + g * cols, S * esz, data + g * cols * esz, S * esz, cols * esz, L1, hipMemcpyHostToDevice, m->stream);
    if (e != hipSuccess) return fail(m, BP_ERR_HIP, "NTT column-slice upload", e, __FILE__, __LINE__);
    if (scalar_fmt == BP_FR_BYTES_LE) BP_TRY(fr_convert_run(m, dbuf[g], N, 0));      // other columns: unused memory, converted and ignored
    BP_TRY(ntt_run_part(m, dbuf[g], log_n, inverse, 1, N, 0, g, R));
    BP_HIP(m, hipEventRecord(m->ev[4], m->stream));               // pass 1 of this member is behind this event: the exchange waits for it
    BP_HIP(m, stream_wait(m->stream));
    return BP_OK;
  });
  if (rc != BP_OK) return rc;
  for (uint32_t to = 0; to < R && rc == BP_OK; to++) {         // member `to` collects rows [to * rows, (to + 1) * rows) of everybody's columns,
    bp_ctx* m = sh[to];                                        // then runs the remaining passes on its e_1
    DeviceGuard guard(m->device);
    for (uint32_t from = 0; from < R && rc == BP_OK; from++) {
      if (from == to) continue;
      const size_t off = (size_t)to * rows * S + (size_t)from * cols;
      // stream order, not the host wait above, is what ties the copy to `from`'s pass 1 (the event was recorded on from's stream)
      hipError_t e = hipStreamWaitEvent(m->stream, sh[from]->ev[4], 0);
      if (e == hipSuccess) e = hipMemcpy2DAsync(tbuf[to] + off, S * esz, tbuf[from] + off, S * esz, cols * esz, rows, hipMemcpyDefault, m->stream);   // the runtime finds the two devices
      if (e != hipSuccess) rc = fail(ctx, BP_ERR_HIP, "NTT block exchange", e, __FILE__, __LINE__);
    }
    if (rc == BP_OK) rc = lift(ctx, m, ntt_run_part(m, dbuf[to], log_n, inverse, 1, N, 1, to, R));
    if (rc == BP_OK && scalar_fmt == BP_FR_BYTES_LE) rc = lift(ctx, m, fr_convert_run(m, dbuf[to], N, 1));
  }
  for (uint32_t g = 0; g < R; g++) {                           // every member, also after a failure elsewhere
    DeviceGuard guard(sh[g]->device);
    hipError_t e = stream_wait(sh[g]->stream);
    if (e != hipSuccess && rc == BP_OK) rc = fail(ctx, BP_ERR_HIP, "NTT over the members", e, __FILE__, __LINE__);
  }
  if (rc != BP_OK) return rc;
  rc = on_members([&](uint32_t g) -> int {                     // outputs e_1 + 2^(l_1) m, e_1 in the member's slice
    bp_ctx* m = sh[g];
    DeviceGuard guard(m->device);
    hipError_t e = hipMemcpy2DAsync(data + g * rows * esz, L1 * esz, dbuf[g] + g * rows, L1 * esz, rows * esz, S, hipMemcpyDeviceToHost, m->stream);
    if (e != hipSuccess) return fail(m, BP_ERR_HIP, "NTT output download", e, __FILE__, __LINE__);
    BP_HIP(m, stream_wait(m->stream));
    return BP_OK;
  });
  if (rc != BP_OK) return rc;
  float ms0 = 0, ms1 = 0;
  for (uint32_t g = 0; g < R; g++) {
    DeviceGuard guard(sh[g]->device);
    float a = 0, b2 = 0;
    if (hipEventElapsedTime(&a, sh[g]->ev[0], sh[g]->ev[1]) == hipSuccess) ms0 = std::max(ms0, a);
    if (hipEventElapsedTime(&b2, sh[g]->ev[2], sh[g]->ev[3]) == hipSuccess) ms1 = std::max(ms1, b2);
  }
  ctx->ntt_ms = ms0 + ms1;                                      // kernels only (slowest member of each phase); the exchange is not in it
  ctx->ntt_passes = sh[0]->ntt_passes;
  ctx->ntt_members = R;
  return BP_OK;
}

int bp_ntt_fr(bp_ctx* ctx, void* data, uint32_t log_n, int inverse, int scalar_fmt, size_t batch, size_t stride) {
  if (!ctx || !fmt_ok(scalar_fmt) || (!data && batch)) return BP_ERR_INVALID_ARG;
  if (log_n > 28) return fail(ctx, BP_ERR_TOO_LARGE, "NTT length > 2^28", hipSuccess, __FILE__, __LINE__);
  if (batch == 0) return BP_OK;
  const size_t N = (size_t)1 << log_n;
  if (batch > 1 && stride < N) return fail(ctx, BP_ERR_INVALID_ARG, "NTT stride < N", hipSuccess, __FILE__, __LINE__);
  if (batch > 65535) return fail(ctx, BP_ERR_TOO_LARGE, "NTT batch > 65535", hipSuccess, __FILE__, __LINE__);
  if (is_group(ctx) && batch > 1) return ntt_columns_over_members(ctx, (uint8_t*)data, log_n, inverse, scalar_fmt, batch, stride);
  if (is_group(ctx) && log_n >= knob_u32("BP_NTT_GROUP_SPLIT_FROM", 22, 11, 29)) {     // one large transform: every member's PCIe link and a share of the work
    const int rc = ntt_one_over_members(ctx, (uint8_t*)data, log_n, inverse, scalar_fmt);
    if (rc != 1) return rc;                 // done, or a real failure (reported, never papered over: host data may be partly written);
  }                                         // 1 = the shape does not split over this many members: on the leader
  DeviceGuard guard(ctx->device);
  ctx->ntt_members = 1;
  const size_t span = (batch - 1) * stride + N;
  fr_t* d;
  BP_TRY(upload_fr(ctx, "io.ntt", data, span, span, scalar_fmt, &d));
  BP_TRY(ntt_run(ctx, d, log_n, inverse, batch, stride));
  BP_TRY(download_fr(ctx, d, data, span, scalar_fmt));
  BP_HIP(ctx, hipEventElapsedTime(&ctx->ntt_ms, ctx->ev[0], ctx->ev[1]));
  return BP_OK;
}

int bp_ntt_last_stats(bp_ctx* ctx, float* device_ms, uint32_t* passes) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  if (ctx->ntt_async_pending) {         // the last transform was only enqueued: its events are read now (0 while it is still running)
    DeviceGuard guard(ctx->device);
    float ms = 0;
    if (hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]) == hipSuccess) {
      ctx->ntt_ms = ms;
      ctx->ntt_async_pending = false;
    } else {
      (void)hipGetLastError();
      ctx->ntt_ms = 0;
    }
  }
  if (device_ms) *device_ms = ctx->ntt_ms;
  if (passes) *passes = ctx->ntt_passes;
  return BP_OK;
}
int bp_ntt_last_members(bp_ctx* ctx) { return ctx ? (int)ctx->ntt_members : BP_ERR_INVALID_ARG; }

// utils.rs:39-43: ROOT_OF_UNITY.pow([2^32 / group_order, 0, 0, 0]) -- integer division, as written
static bool host_root_of_unity(fr_t& out, uint64_t group_order) {
  if (group_order == 0) return false;                       // division by zero panics in the reference
  const uint64_t e = ((uint64_t)1 << 32) / group_order;
  uint32_t e32[2] = {(uint32_t)e, (uint32_t)(e >> 32)};
  Fr::pow(out, fr_root_of_unity(false), e32, 2);
  return true;
}
int bp_root_of_unity(uint64_t group_order, int scalar_fmt, uint8_t out32[32]) {
  if (!out32 || !fmt_ok(scalar_fmt)) return BP_ERR_INVALID_ARG;
  fr_t w;
  if (!host_root_of_unity(w, group_order)) return BP_ERR_INVALID_ARG;
  fr_mont_to_bytes(out32, w, scalar_fmt);
  return BP_OK;
}
int bp_roots_of_unity(bp_ctx* ctx, uint64_t group_order, int scalar_fmt, void* out) {
  if (!ctx || !out || !fmt_ok(scalar_fmt)) return BP_ERR_INVALID_ARG;
  fr_t w;
  if (!host_root_of_unity(w, group_order)) return fail(ctx, BP_ERR_INVALID_ARG, "group_order == 0", hipSuccess, __FILE__, __LINE__);
  if (group_order > ((uint64_t)1 << 28)) return fail(ctx, BP_ERR_TOO_LARGE, "group_order > 2^28", hipSuccess, __FILE__, __LINE__);
  DeviceGuard guard(ctx->device);
  fr_t* d;
  BP_TRY(ws_get(ctx, "io.roots", group_order * sizeof(fr_t), (void**)&d));
  BP_TRY(roots_run(ctx, w, group_order, d));
  return download_fr(ctx, d, out, group_order, scalar_fmt);
}

int bp_fr_convert(const void* in, size_t n, int from_fmt, int to_fmt, void* out) {
  if (!fmt_ok(from_fmt) || !fmt_ok(to_fmt) || (n && (!in || !out))) return BP_ERR_INVALID_ARG;
  for (size_t i = 0; i < n; i++) {
    fr_t v;
    if (!fr_bytes_to_mont(v, (const uint8_t*)in + 32 * i, from_fmt)) return BP_ERR_BAD_SCALAR;
    fr_mont_to_bytes((uint8_t*)out + 32 * i, v, to_fmt);
  }
  return BP_OK;
}

int bp_fr_synthetic_device(bp_ctx* ctx, void* d_out, size_t n, uint64_t seed) {
  if (!ctx || (n && !d_out)) return BP_ERR_INVALID_ARG;
  DeviceGuard guard(ctx->device);
  BP_TRY(fr_synthetic_run(ctx, (fr_t*)d_out, n, seed));
  BP_HIP(ctx, stream_wait(ctx->stream));
  return BP_OK;
}

// ---------------------------------------------------------------------------------------------- Polynomial
int bp_poly_evaluate(bp_ctx* ctx, const void* coeffs, size_t n, int basis, const void* x32, int scalar_fmt, void* out32) {
  if (!ctx || !x32 || !out32 || !fmt_ok(scalar_fmt) || !basis_ok(basis) || (n && !coeffs)) return BP_ERR_INVALID_ARG;
  if (basis != BP_BASIS_MONOMIAL) return fail(ctx, BP_ERR_BASIS, "coeffs_evaluate needs the Monomial basis", hipSuccess, __FILE__, __LINE__);
  fr_t x, r;
  if (!fr_bytes_to_mont(x, (const uint8_t*)x32, scalar_fmt)) return fail(ctx, BP_ERR_BAD_SCALAR, "scalar >= q", hipSuccess, __FILE__, __LINE__);
  DeviceGuard guard(ctx->device);
  fr_t* d;
  BP_TRY(upload_fr(ctx, "io.poly_a", coeffs, n, n, scalar_fmt, &d));
  BP_TRY(poly_eval_run(ctx, d, n, x, &r));
  fr_mont_to_bytes((uint8_t*)out32, r, scalar_fmt);
  return BP_OK;
}

static int poly_addsub(bp_ctx* ctx, const void* a, size_t na, const void* b, size_t nb, int basis, int fmt, void* out, size_t* n_out,
                       int op) {
  if (!ctx || !n_out || !fmt_ok(fmt) || !basis_ok(basis) || (na && !a) || (nb && !b)) return BP_ERR_INVALID_ARG;
  if (basis == BP_BASIS_LAGRANGE && na != nb)
    return fail(ctx, BP_ERR_LENGTH, "Polynomials must have the same length", hipSuccess, __FILE__, __LINE__);
  const size_t n = std::max(na, nb);
  *n_out = n;
  if (n == 0) return BP_OK;
  if (!out) return BP_ERR_INVALID_ARG;
  DeviceGuard guard(ctx->device);
  fr_t *da, *db, *dout;
  // add/sub commute with the Montgomery map, so canonical inputs need no conversion at all
  BP_TRY(upload_fr(ctx, "io.poly_a", a, na, na, BP_FR_MONT, &da));
  BP_TRY(upload_fr(ctx, "io.poly_b", b, nb, nb, BP_FR_MONT, &db));
  BP_TRY(ws_get(ctx, "io.poly_out", n * sizeof(fr_t), (void**)&dout));
  BP_TRY(fr_binary_run(ctx, da, na, db, nb, dout, n, op));
  return download_fr(ctx, dout, out, n, BP_FR_MONT);
}
int bp_poly_add(bp_ctx* ctx, const void* a, size_t na, const void* b, size_t nb, int basis, int scalar_fmt, void* out, size_t* n_out) {
  return poly_addsub(ctx, a, na, b, nb, basis, scalar_fmt, out, n_out, 0);
}
int bp_poly_sub(bp_ctx* ctx, const void* a, size_t na, const void* b, size_t nb, int basis, int scalar_fmt, void* out, size_t* n_out) {
  return poly_addsub(ctx, a, na, b, nb, basis, scalar_fmt, out, n_out, 1);
}

int bp_poly_scalar_op(bp_ctx* ctx, const void* a, size_t n, int basis, const void* s32, int op, int scalar_fmt, void* out) {
  if (!ctx || !s32 || !fmt_ok(scalar_fmt) || !basis_ok(basis) || op < 0 || op > 2 || (n && (!a || !out))) return BP_ERR_INVALID_ARG;
  // Monomial Add/Sub<Scalar> index values[0] (polynomial.rs:62,123): an empty polynomial panics there
  if (n == 0) return op == 2 || basis == BP_BASIS_LAGRANGE ? BP_OK
                                                           : fail(ctx, BP_ERR_INVALID_ARG, "empty polynomial", hipSuccess, __FILE__, __LINE__);
  fr_t s;
  if (!fr_bytes_to_mont(s, (const uint8_t*)s32, scalar_fmt)) return fail(ctx, BP_ERR_BAD_SCALAR, "scalar >= q", hipSuccess, __FILE__, __LINE__);
  DeviceGuard guard(ctx->device);
  fr_t *da, *dout;
  BP_TRY(upload_fr(ctx, "io.poly_a", a, n, n, scalar_fmt, &da));
  BP_TRY(ws_get(ctx, "io.poly_out", n * sizeof(fr_t), (void**)&dout));
  if (op == 2) {
    BP_TRY(fr_scalar_run(ctx, da, s, dout, n, 2));
  } else if (basis == BP_BASIS_MONOMIAL) {
    BP_HIP(ctx, hipMemcpyAsync(dout, da, n * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
    BP_TRY(fr_scalar_run(ctx, da, s, dout, 1, op));          // values[0] += / -= rhs
  } else {
    BP_TRY(fr_scalar_run(ctx, da, s, dout, n, 0));           // Lagrange: += rhs for Add AND Sub (polynomial.rs:126-128)
  }
  return download_fr(ctx, dout, out, n, scalar_fmt);
}

int bp_poly_mul(bp_ctx* ctx, const void* a, size_t na, const void* b, size_t nb, int basis, int scalar_fmt, void* out, size_t* n_out) {
  if (!ctx || !n_out || !fmt_ok(scalar_fmt) || !basis_ok(basis) || !a || !b || !out) return BP_ERR_INVALID_ARG;
  if (basis != BP_BASIS_MONOMIAL) return fail(ctx, BP_ERR_BASIS, "Polynomial * Polynomial: Lagrange basis is todo!() in the reference", hipSuccess, __FILE__, __LINE__);
  if (na == 0 || nb == 0) return fail(ctx, BP_ERR_INVALID_ARG, "empty polynomial (len - 1 underflows, polynomial.rs:248-249)", hipSuccess, __FILE__, __LINE__);
  // find_next_power_of_two(n, m) with n = na-1, m = nb-1: smallest power of two >= n + m + 1 (utils.rs:54-61)
  const size_t target = na + nb - 1;
  uint32_t k = 0;
  while (((size_t)1 << k) < target) k++;
  if (k > 28) return fail(ctx, BP_ERR_TOO_LARGE, "product too long", hipSuccess, __FILE__, __LINE__);
  const size_t N = (size_t)1 << k;
  DeviceGuard guard(ctx->device);
  fr_t* d;
  BP_TRY(ws_get(ctx, "io.poly_mul", 2 * N * sizeof(fr_t), (void**)&d));
  BP_HIP(ctx, hipMemsetAsync(d, 0, 2 * N * sizeof(fr_t), ctx->stream));
  BP_HIP(ctx, hipMemcpyAsync(d, a, na * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
  BP_HIP(ctx, hipMemcpyAsync(d + N, b, nb * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
  if (scalar_fmt == BP_FR_BYTES_LE) BP_TRY(fr_convert_run(ctx, d, 2 * N, 0));
  BP_TRY(ntt_run(ctx, d, k, 0, 2, N));                        // evaluate both at the N roots (polynomial.rs:255-260)
  BP_TRY(fr_binary_run(ctx, d, N, d + N, N, d, N, 2));        // pointwise product (:262-266)
  BP_TRY(ntt_run(ctx, d, k, 1, 1, N));                        // i_ntt_381 (:270)
  *n_out = target;                                            // [0 ..= n+m] (:272)
  return download_fr(ctx, d, out, target, scalar_fmt);
}

int bp_poly_div(bp_ctx* ctx, const void* a, size_t na, const void* b, size_t nb, int basis, int scalar_fmt, void* out, size_t* n_out) {
  if (!ctx || !n_out || !fmt_ok(scalar_fmt) || !basis_ok(basis) || (na && !a) || (nb && !b)) return BP_ERR_INVALID_ARG;
  if (basis != BP_BASIS_MONOMIAL) return fail(ctx, BP_ERR_BASIS, "Div needs the Monomial basis (polynomial.rs:319)", hipSuccess, __FILE__, __LINE__);
  const fr_t* ha = (const fr_t*)a;
  const fr_t* hb = (const fr_t*)b;
  while (na > 0 && big_is_zero(ha[na - 1])) na--;             // polynomial.rs:325-339 (zero is all-zero in both formats)
  while (nb > 0 && big_is_zero(hb[nb - 1])) nb--;
  if (nb == 0) return fail(ctx, BP_ERR_DIV_ZERO, "division by the zero polynomial", hipSuccess, __FILE__, __LINE__);
  *n_out = 0;
  if (na < nb) return BP_OK;
  if (!out) return BP_ERR_INVALID_ARG;
  const size_t nq = na - nb + 1;
  DeviceGuard guard(ctx->device);
  fr_t *da, *db, *dq;
  BP_TRY(upload_fr(ctx, "io.poly_a", a, na, na, scalar_fmt, &da));
  BP_TRY(upload_fr(ctx, "io.poly_b", b, nb, nb, scalar_fmt, &db));
  BP_TRY(ws_get(ctx, "io.poly_out", nq * sizeof(fr_t), (void**)&dq));
  fr_t b0, b_lead;
  if (!fr_bytes_to_mont(b0, (const uint8_t*)&hb[0], scalar_fmt) || !fr_bytes_to_mont(b_lead, (const uint8_t*)&hb[nb - 1], scalar_fmt))
    return fail(ctx, BP_ERR_BAD_SCALAR, "scalar >= q", hipSuccess, __FILE__, __LINE__);
  bool binomial = nb >= 2;
  for (size_t i = 1; i + 1 < nb && binomial; i++) binomial = big_is_zero(hb[i]);
  BP_TRY(poly_div_run(ctx, da, na, db, nb, b0, b_lead, binomial, dq, nq));
  std::vector<fr_t> q(nq);
  BP_TRY(download_fr(ctx, dq, q.data(), nq, scalar_fmt));
  // The reference inserts one quotient coefficient per loop turn and pops every newly zero leading remainder
  // term (polynomial.rs:371-376): its result is the true quotient with the zero coefficients squeezed out.
  fr_t* o = (fr_t*)out;
  size_t m = 0;
  for (size_t i = 0; i < nq; i++)
    if (!big_is_zero(q[i])) o[m++] = q[i];
  *n_out = m;
  return BP_OK;
}

// ---------------------------------------------------------------------------------------------- device-resident
// The same operators on HBM-resident Montgomery data (the quotient / linearisation
// pipeline without PCIe round trips).  Length rules and quirks are those of the host-pointer entry points.
static int poly_addsub_device(bp_ctx* ctx, const void* a, size_t na, const void* b, size_t nb, int basis, void* out, size_t* n_out, int op) {
  if (!ctx || !n_out || !basis_ok(basis) || (na && !a) || (nb && !b)) return BP_ERR_INVALID_ARG;
  if (basis == BP_BASIS_LAGRANGE && na != nb) return fail(ctx, BP_ERR_LENGTH, "Polynomials must have the same length", hipSuccess, __FILE__, __LINE__);
  const size_t n = std::max(na, nb);
  *n_out = n;
  if (n == 0) return BP_OK;
  if (!out) return BP_ERR_INVALID_ARG;
  DeviceGuard guard(ctx->device);
  BP_TRY(fr_binary_run(ctx, (const fr_t*)a, na, (const fr_t*)b, nb, (fr_t*)out, n, op));
  BP_HIP(ctx, stream_wait(ctx->stream));
  return BP_OK;
}
int bp_poly_add_device(bp_ctx* ctx, const void* d_a, size_t na, const void* d_b, size_t nb, int basis, void* d_out, size_t* n_out) {
  return poly_addsub_device(ctx, d_a, na, d_b, nb, basis, d_out, n_out, 0);
}
int bp_poly_sub_device(bp_ctx* ctx, const void* d_a, size_t na, const void* d_b, size_t nb, int basis, void* d_out, size_t* n_out) {
  return poly_addsub_device(ctx, d_a, na, d_b, nb, basis, d_out, n_out, 1);
}
int bp_poly_scalar_op_device(bp_ctx* ctx, const void* d_a, size_t n, int basis, const void* s32_mont, int op, void* d_out) {
  if (!ctx || !s32_mont || !basis_ok(basis) || op < 0 || op > 2 || (n && (!d_a || !d_out))) return BP_ERR_INVALID_ARG;
  if (n == 0) return op == 2 || basis == BP_BASIS_LAGRANGE ? BP_OK : fail(ctx, BP_ERR_INVALID_ARG, "empty polynomial", hipSuccess, __FILE__, __LINE__);
  fr_t s;
  memcpy(&s, s32_mont, 32);
  DeviceGuard guard(ctx->device);
  const fr_t* a = (const fr_t*)d_a;
  fr_t* out = (fr_t*)d_out;
  if (op == 2) {
    BP_TRY(fr_scalar_run(ctx, a, s, out, n, 2));
  } else if (basis == BP_BASIS_MONOMIAL) {
    if (out != a) BP_HIP(ctx, hipMemcpyAsync(out, a, n * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
    BP_TRY(fr_scalar_run(ctx, a, s, out, 1, op));
  } else {
    BP_TRY(fr_scalar_run(ctx, a, s, out, n, 0));
  }
  BP_HIP(ctx, stream_wait(ctx->stream));
  return BP_OK;
}
int bp_poly_mul_device(bp_ctx* ctx, const void* d_a, size_t na, const void* d_b, size_t nb, int basis, void* d_out, size_t* n_out) {
  if (!ctx || !n_out || !basis_ok(basis) || !d_a || !d_b || !d_out) return BP_ERR_INVALID_ARG;
  if (basis != BP_BASIS_MONOMIAL) return fail(ctx, BP_ERR_BASIS, "Polynomial * Polynomial: Lagrange basis is todo!() in the reference", hipSuccess, __FILE__, __LINE__);
  if (na == 0 || nb == 0) return fail(ctx, BP_ERR_INVALID_ARG, "empty polynomial", hipSuccess, __FILE__, __LINE__);
  const size_t target = na + nb - 1;
  uint32_t k = 0;
  while (((size_t)1 << k) < target) k++;
  if (k > 28) return fail(ctx, BP_ERR_TOO_LARGE, "product too long", hipSuccess, __FILE__, __LINE__);
  const size_t N = (size_t)1 << k;
  DeviceGuard guard(ctx->device);
  fr_t* d;
  BP_TRY(ws_get(ctx, "io.poly_mul", 2 * N * sizeof(fr_t), (void**)&d));
  BP_HIP(ctx, hipMemsetAsync(d, 0, 2 * N * sizeof(fr_t), ctx->stream));
  BP_HIP(ctx, hipMemcpyAsync(d, d_a, na * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
  BP_HIP(ctx, hipMemcpyAsync(d + N, d_b, nb * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
  BP_TRY(ntt_run(ctx, d, k, 0, 2, N));
  BP_TRY(fr_binary_run(ctx, d, N, d + N, N, d, N, 2));
  BP_TRY(ntt_run(ctx, d, k, 1, 1, N));
  BP_HIP(ctx, hipMemcpyAsync(d_out, d, target * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
  BP_HIP(ctx, stream_wait(ctx->stream));
  *n_out = target;
  return BP_OK;
}
int bp_poly_div_device(bp_ctx* ctx, const void* d_a, size_t na, const void* d_b, size_t nb, int basis, void* d_out, size_t* n_out) {
  if (!ctx || !n_out || !basis_ok(basis) || (na && !d_a) || (nb && !d_b)) return BP_ERR_INVALID_ARG;
  if (basis != BP_BASIS_MONOMIAL) return fail(ctx, BP_ERR_BASIS, "Div needs the Monomial basis (polynomial.rs:319)", hipSuccess, __FILE__, __LINE__);
  DeviceGuard guard(ctx->device);
  size_t na_eff, nb_eff, dummy, mid_nonzero = 0;
  BP_TRY(fr_nonzero_stats_run(ctx, (const fr_t*)d_a, na, 0, 0, &na_eff, &dummy));           // trailing zeros trimmed (:325-339)
  BP_TRY(fr_nonzero_stats_run(ctx, (const fr_t*)d_b, nb, 0, 0, &nb_eff, &dummy));
  if (nb_eff == 0) return fail(ctx, BP_ERR_DIV_ZERO, "division by the zero polynomial", hipSuccess, __FILE__, __LINE__);
  *n_out = 0;
  if (na_eff < nb_eff) return BP_OK;
  if (!d_out) return BP_ERR_INVALID_ARG;
  if (nb_eff > 2) BP_TRY(fr_nonzero_stats_run(ctx, (const fr_t*)d_b, nb_eff, 1, nb_eff - 1, &dummy, &mid_nonzero));
  const bool binomial = nb_eff >= 2 && mid_nonzero == 0;
  fr_t ends[2];
  BP_HIP(ctx, hipMemcpyAsync(&ends[0], d_b, sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
  BP_HIP(ctx, hipMemcpyAsync(&ends[1], (const fr_t*)d_b + (nb_eff - 1), sizeof(fr_t), hipMemcpyDeviceToHost, ctx->stream));
  BP_HIP(ctx, stream_wait(ctx->stream));
  const size_t nq = na_eff - nb_eff + 1;
  fr_t *work, *q;
  BP_TRY(ws_get(ctx, "io.poly_div_work", na_eff * sizeof(fr_t), (void**)&work));       // the general path clobbers its dividend
  BP_TRY(ws_get(ctx, "io.poly_out", nq * sizeof(fr_t), (void**)&q));
  BP_HIP(ctx, hipMemcpyAsync(work, d_a, na_eff * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
  BP_TRY(poly_div_run(ctx, work, na_eff, (const fr_t*)d_b, nb_eff, ends[0], ends[1], binomial, q, nq));
  // the reference squeezes zero quotient coefficients out (polynomial.rs:371-376); they are rare, so count first
  size_t q_eff, q_nonzero;
  BP_TRY(fr_nonzero_stats_run(ctx, q, nq, 0, nq, &q_eff, &q_nonzero));
  if (q_nonzero == nq) {
    BP_HIP(ctx, hipMemcpyAsync(d_out, q, nq * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
    BP_HIP(ctx, stream_wait(ctx->stream));
    *n_out = nq;
    return BP_OK;
  }
  size_t m = nq;
  BP_TRY(fr_compact_nonzero_run(ctx, q, &m));
  if (m) BP_HIP(ctx, hipMemcpyAsync(d_out, q, m * sizeof(fr_t), hipMemcpyDeviceToDevice, ctx->stream));
  BP_HIP(ctx, stream_wait(ctx->stream));
  *n_out = m;
  return BP_OK;
}
int bp_poly_evaluate_device(bp_ctx* ctx, const void* d_coeffs, size_t n, int basis, const void* x32_mont, void* out32_mont) {
  if (!ctx || !x32_mont || !out32_mont || !basis_ok(basis) || (n && !d_coeffs)) return BP_ERR_INVALID_ARG;
  if (basis != BP_BASIS_MONOMIAL) return fail(ctx, BP_ERR_BASIS, "coeffs_evaluate needs the Monomial basis", hipSuccess, __FILE__, __LINE__);
  fr_t x, r;
  memcpy(&x, x32_mont, 32);
  DeviceGuard guard(ctx->device);
  BP_TRY(poly_eval_run(ctx, (const fr_t*)d_coeffs, n, x, &r));
  memcpy(out32_mont, &r, 32);
  return BP_OK;
}
int bp_poly_scale_powers_device(bp_ctx* ctx, const void* d_a, size_t n, const void* w32_mont, void* d_out) {
  if (!ctx || !w32_mont || (n && (!d_a || !d_out))) return BP_ERR_INVALID_ARG;
  fr_t w;
  memcpy(&w, w32_mont, 32);
  DeviceGuard guard(ctx->device);
  BP_TRY(fr_scale_powers_run(ctx, (const fr_t*)d_a, n, w, (fr_t*)d_out));
  BP_HIP(ctx, stream_wait(ctx->stream));
  return BP_OK;
}
int bp_roots_of_unity_device(bp_ctx* ctx, uint64_t group_order, void* d_out) {
  if (!ctx || !d_out) return BP_ERR_INVALID_ARG;
  fr_t w;
  if (!host_root_of_unity(w, group_order)) return fail(ctx, BP_ERR_INVALID_ARG, "group_order == 0", hipSuccess, __FILE__, __LINE__);
  if (group_order > ((uint64_t)1 << 28)) return fail(ctx, BP_ERR_TOO_LARGE, "group_order > 2^28", hipSuccess, __FILE__, __LINE__);
  DeviceGuard guard(ctx->device);
  BP_TRY(roots_run(ctx, w, group_order, (fr_t*)d_out));
  BP_HIP(ctx, stream_wait(ctx->stream));
  return BP_OK;
}
int bp_grand_product_device(bp_ctx* ctx, const void* a, const void* b, const void* c, const void* s1, const void* s2, const void* s3, size_t n,
                            const void* beta32, const void* gamma32, const void* k1_32, const void* k2_32, void* d_z) {
  if (!ctx || !beta32 || !gamma32 || !k1_32 || !k2_32 || (n && (!a || !b || !c || !s1 || !s2 || !s3 || !d_z))) return BP_ERR_INVALID_ARG;
  if (n == 0) return BP_OK;
  if (n > ((size_t)1 << 25)) return fail(ctx, BP_ERR_TOO_LARGE, "grand product longer than 2^25", hipSuccess, __FILE__, __LINE__);
  fr_t beta, gamma, k1, k2, root;
  memcpy(&beta, beta32, 32); memcpy(&gamma, gamma32, 32); memcpy(&k1, k1_32, 32); memcpy(&k2, k2_32, 32);
  host_root_of_unity(root, n);
  DeviceGuard guard(ctx->device);
  BP_TRY(grand_product_run(ctx, (const fr_t*)a, (const fr_t*)b, (const fr_t*)c, (const fr_t*)s1, (const fr_t*)s2, (const fr_t*)s3, n, beta, gamma,
                           k1, k2, root, (fr_t*)d_z));
  BP_HIP(ctx, stream_wait(ctx->stream));
  return BP_OK;
}
int bp_commit_device(bp_ctx* ctx, uint64_t srs_handle, const void* d_coeffs, size_t n, int basis, uint8_t out96[96]) {
  if (!ctx || !basis_ok(basis) || !out96) return BP_ERR_INVALID_ARG;
  if (basis != BP_BASIS_MONOMIAL) return fail(ctx, BP_ERR_BASIS, "commit needs the Monomial basis (setup.rs:34)", hipSuccess, __FILE__, __LINE__);
  uint8_t part[144];
  BP_TRY(bp_msm_g1_partial(ctx, srs_handle, 0, d_coeffs, n, BP_FR_MONT, 1, part));
  return bp_g1_partial_to_bytes96(part, out96);
}

// Several commitments against one SRS in one call (the three of prover.rs:249-251, of :483-485, the two of :640-641): their
// pipelines are in flight together (commit_many: per-device lanes, or queued shards on a group context), so one commitment's
// latency-bound tail runs under another's bulk kernel.
int bp_commit_many_device(bp_ctx* ctx, uint64_t srs_handle, const void* const* d_coeffs, const size_t* n, size_t count, int basis,
                          uint8_t* out96) {
  if (!ctx || !basis_ok(basis) || (count && (!d_coeffs || !n || !out96))) return BP_ERR_INVALID_ARG;
  if (basis != BP_BASIS_MONOMIAL) return fail(ctx, BP_ERR_BASIS, "commit needs the Monomial basis (setup.rs:34)", hipSuccess, __FILE__, __LINE__);
  if (count > 64) return fail(ctx, BP_ERR_TOO_LARGE, "more than 64 commitments in one call", hipSuccess, __FILE__, __LINE__);
  for (size_t i = 0; i < count; i++)
    if (n[i] && !d_coeffs[i]) return BP_ERR_INVALID_ARG;
  std::vector<g1_proj> cm(count);
  BP_TRY(commit_many(ctx, srs_handle, reinterpret_cast<const fr_t* const*>(d_coeffs), n, (int)count, cm.data()));
  for (size_t i = 0; i < count; i++) host_encode96(out96 + 96 * i, cm[i]);
  return BP_OK;
}

int bp_grand_product(bp_ctx* ctx, const void* a, const void* b, const void* c, const void* s1, const void* s2, const void* s3, size_t n,
                     const void* beta32, const void* gamma32, const void* k1_32, const void* k2_32, int scalar_fmt, void* z_out) {
  if (!ctx || !fmt_ok(scalar_fmt) || !beta32 || !gamma32 || !k1_32 || !k2_32 || (n && (!a || !b || !c || !s1 || !s2 || !s3 || !z_out)))
    return BP_ERR_INVALID_ARG;
  if (n == 0) return BP_OK;
  if (n > ((size_t)1 << 25)) return fail(ctx, BP_ERR_TOO_LARGE, "grand product longer than 2^25", hipSuccess, __FILE__, __LINE__);
  fr_t beta, gamma, k1, k2, root;
  if (!fr_bytes_to_mont(beta, (const uint8_t*)beta32, scalar_fmt) || !fr_bytes_to_mont(gamma, (const uint8_t*)gamma32, scalar_fmt) ||
      !fr_bytes_to_mont(k1, (const uint8_t*)k1_32, scalar_fmt) || !fr_bytes_to_mont(k2, (const uint8_t*)k2_32, scalar_fmt))
    return fail(ctx, BP_ERR_BAD_SCALAR, "scalar >= q", hipSuccess, __FILE__, __LINE__);
  host_root_of_unity(root, n);                                   // roots_of_unity(group_order), utils.rs:45-52
  DeviceGuard guard(ctx->device);
  fr_t* cols;
  BP_TRY(ws_get(ctx, "io.gp_cols", 7 * n * sizeof(fr_t), (void**)&cols));
  const void* src[6] = {a, b, c, s1, s2, s3};
  for (int j = 0; j < 6; j++) BP_HIP(ctx, hipMemcpyAsync(cols + (size_t)j * n, src[j], n * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
  if (scalar_fmt == BP_FR_BYTES_LE) BP_TRY(fr_convert_run(ctx, cols, 6 * n, 0));
  fr_t* z = cols + 6 * n;
  BP_TRY(grand_product_run(ctx, cols, cols + n, cols + 2 * n, cols + 3 * n, cols + 4 * n, cols + 5 * n, n, beta, gamma, k1, k2, root, z));
  return download_fr(ctx, z, z_out, n, scalar_fmt);
}

int bp_commit(bp_ctx* ctx, uint64_t srs_handle, const void* coeffs, size_t n, int basis, int scalar_fmt, uint8_t out96[96]) {
  if (!ctx || !basis_ok(basis)) return BP_ERR_INVALID_ARG;
  if (basis != BP_BASIS_MONOMIAL) return fail(ctx, BP_ERR_BASIS, "commit needs the Monomial basis (setup.rs:34)", hipSuccess, __FILE__, __LINE__);
  return bp_msm_g1(ctx, srs_handle, coeffs, n, scalar_fmt, out96);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------- prover
int bp_circuit_load(bp_ctx* ctx, uint32_t log_n, const void* const columns[8], int scalar_fmt, int columns_on_device, uint64_t* handle) {
  if (!ctx || !columns || !handle || !fmt_ok(scalar_fmt)) return BP_ERR_INVALID_ARG;
  if (log_n < 3 || log_n > 24) return fail(ctx, BP_ERR_INVALID_ARG, "circuit: log_n must be in 3..24", hipSuccess, __FILE__, __LINE__);
  if (columns_on_device && scalar_fmt != BP_FR_MONT) return fail(ctx, BP_ERR_INVALID_ARG, "device columns must be Montgomery", hipSuccess, __FILE__, __LINE__);
  for (int k = 0; k < 8; k++)
    if (!columns[k]) return BP_ERR_INVALID_ARG;
  DeviceGuard guard(ctx->device);
  const size_t n = (size_t)1 << log_n;
  fr_t* lag = nullptr;
  BP_HIP(ctx, hipMalloc((void**)&lag, 8 * n * sizeof(fr_t)));
  for (int k = 0; k < 8; k++) {
    hipError_t e = hipMemcpyAsync(lag + (size_t)k * n, columns[k], n * sizeof(fr_t), columns_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                  ctx->stream);
    if (e != hipSuccess) {
      (void)hipFree(lag);
      return fail(ctx, BP_ERR_HIP, "circuit upload", e, __FILE__, __LINE__);
    }
  }
  int rc = BP_OK;
  if (scalar_fmt == BP_FR_BYTES_LE) rc = fr_convert_run(ctx, lag, 8 * n, 0);
  CircuitEntry e;
  if (rc == BP_OK) rc = circuit_build(ctx, log_n, lag, &e);
  if (rc != BP_OK) {
    (void)hipFree(lag);
    return rc;
  }
  if (is_group(ctx)) {                       // round 3 by coset over the members: their shares of the coset tables
    rc = circuit_split_build(ctx, e);
    if (rc != BP_OK) {
      circuit_release(e);
      return rc;
    }
  }
  *handle = ctx->next_handle++;
  ctx->circuits[*handle] = e;
  return BP_OK;
}
int bp_circuit_free(bp_ctx* ctx, uint64_t handle) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  auto it = ctx->circuits.find(handle);
  if (it == ctx->circuits.end()) return fail(ctx, BP_ERR_INVALID_ARG, "unknown circuit handle", hipSuccess, __FILE__, __LINE__);
  DeviceGuard guard(ctx->device);
  BP_HIP(ctx, stream_wait(ctx->stream));
  circuit_release(it->second);
  ctx->circuits.erase(it);
  return BP_OK;
}
int bp_make_s_polynomials(uint32_t log_n, const uint32_t* wire_ids, void* s1, void* s2, void* s3) {
  if (!wire_ids || !s1 || !s2 || !s3 || log_n > 26) return BP_ERR_INVALID_ARG;
  const size_t n = (size_t)1 << log_n, cells = 3 * n;
  fr_t w;
  host_root_of_unity(w, n);
  std::vector<fr_t> root(n);                                  // roots_of_unity(group_order), utils.rs:45-52
  root[0] = Fr::one();
  for (size_t i = 1; i < n; i++) Fr::mul(root[i], root[i - 1], w);
  fr_t col[3], two, three;
  col[0] = Fr::one();
  Fr::add(two, col[0], col[0]);
  Fr::add(three, two, col[0]);
  col[1] = two;
  col[2] = three;
  auto label = [&](size_t cell) { fr_t r; Fr::mul(r, root[cell / 3], col[cell % 3]); return r; };      // Cell::label, utils.rs:28-37
  // cells grouped by variable, row-major inside a group (program.rs:80-99)
  std::vector<uint64_t> order(cells);
  for (size_t k = 0; k < cells; k++) order[k] = ((uint64_t)wire_ids[k] << 32) | k;
  std::sort(order.begin(), order.end());
  fr_t* out[3] = {(fr_t*)s1, (fr_t*)s2, (fr_t*)s3};
  for (size_t a = 0; a < cells;) {
    size_t b = a;
    while (b < cells && (order[b] >> 32) == (order[a] >> 32)) b++;
    for (size_t j = a; j < b; j++) {                          // uses[j] -> uses[j + 1 cyclically] receives label(uses[j]), :126-137
      const size_t cell = (size_t)(order[j] & 0xffffffffu), next = (size_t)(order[j + 1 < b ? j + 1 : a] & 0xffffffffu);
      out[next % 3][next / 3] = label(cell);
    }
    a = b;
  }
  return BP_OK;
}
int bp_circuit_commitments(bp_ctx* ctx, uint64_t srs_handle, uint64_t circuit_handle, uint8_t out768[768]) {
  if (!ctx || !out768) return BP_ERR_INVALID_ARG;
  auto it = ctx->circuits.find(circuit_handle);
  if (it == ctx->circuits.end()) return fail(ctx, BP_ERR_INVALID_ARG, "unknown circuit handle", hipSuccess, __FILE__, __LINE__);
  const size_t n = (size_t)1 << it->second.log_n;
  const fr_t* polys[8];
  size_t lens[8];
  g1_proj cm[8];
  for (int k = 0; k < 8; k++) { polys[k] = it->second.coef + (size_t)k * n; lens[k] = n; }
  BP_TRY(commit_many(ctx, srs_handle, polys, lens, 8, cm));
  for (int k = 0; k < 8; k++) host_encode96(out768 + 96 * k, cm[k]);
  return BP_OK;
}
int bp_prove(bp_ctx* ctx, uint64_t srs_handle, uint64_t circuit_handle, const void* a, const void* b, const void* c, const void* public_input,
             int scalar_fmt, int witness_on_device, const uint8_t blinders[352], uint8_t proof[624]) {
  if (!ctx || !a || !b || !c || !blinders || !proof || !fmt_ok(scalar_fmt)) return BP_ERR_INVALID_ARG;
  if (witness_on_device && scalar_fmt != BP_FR_MONT) return fail(ctx, BP_ERR_INVALID_ARG, "device witness must be Montgomery", hipSuccess, __FILE__, __LINE__);
  auto it = ctx->circuits.find(circuit_handle);
  if (it == ctx->circuits.end()) return fail(ctx, BP_ERR_INVALID_ARG, "unknown circuit handle", hipSuccess, __FILE__, __LINE__);
  SrsEntry* srs;
  BP_TRY(srs_find(ctx, srs_handle, &srs));
  const size_t n = (size_t)1 << it->second.log_n;
  (void)srs;      // an SRS shorter than group_order + 6 powers truncates the commitments exactly as Setup::commit's zip does (msm.rs:29)
  fr_t blind[11];
  for (int j = 0; j < 11; j++)
    if (!fr_bytes_to_mont(blind[j], blinders + 32 * j, BP_FR_BYTES_LE)) return fail(ctx, BP_ERR_BAD_SCALAR, "blinder >= q", hipSuccess, __FILE__, __LINE__);
  DeviceGuard guard(ctx->device);
  fr_t* wit;
  BP_TRY(ws_get(ctx, "prove.witness", 4 * n * sizeof(fr_t), (void**)&wit));
  const void* cols[4] = {a, b, c, public_input};
  // host witness of 2^18 gates and more on one device: round 1 stages the columns itself, uploads beside the commitments (BP_PROVE_STAGED=0: off)
  bool staged = !witness_on_device && !is_group(ctx) && it->second.log_n >= 18;
  {
    const char* v = knob("BP_PROVE_STAGED");
    if (v && *v == '0') staged = false;
    if (v && *v == '1') staged = !witness_on_device && !is_group(ctx);
  }
  if (staged) {
    const ProveStaged hw = {{a, b, c, public_input}, scalar_fmt};
    return prove_run(ctx, srs_handle, it->second, wit, blind, proof, public_input == nullptr, &hw);
  }
  const hipMemcpyKind kind = witness_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  for (int k = 0; k < 4; k++) {
    if (cols[k]) BP_HIP(ctx, hipMemcpyAsync(wit + (size_t)k * n, cols[k], n * sizeof(fr_t), kind, ctx->stream));
    else BP_HIP(ctx, hipMemsetAsync(wit + (size_t)k * n, 0, n * sizeof(fr_t), ctx->stream));
  }
  if (scalar_fmt == BP_FR_BYTES_LE) BP_TRY(fr_convert_run(ctx, wit, 4 * n, 0));
  return prove_run(ctx, srs_handle, it->second, wit, blind, proof, public_input == nullptr);
}
int bp_prove_last_stats(bp_ctx* ctx, float round_ms[5], float* total_ms) {
  if (!ctx) return BP_ERR_INVALID_ARG;
  if (round_ms) memcpy(round_ms, ctx->prove_ms, 5 * sizeof(float));
  if (total_ms) *total_ms = ctx->prove_ms[5];
  return BP_OK;
}
int bp_transcript_test_vector(uint8_t out32[32]) {
  if (!out32) return BP_ERR_INVALID_ARG;
  transcript_test_vector(out32);
  return BP_OK;
}
