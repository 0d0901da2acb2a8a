// capi_ctx.hip -- C ABI of include/bp_msm_ntt.h, part 1: contexts (bp_init / bp_init_multi / streams), workspaces, the member
// threads of a group context, and the O(1)/O(W) host epilogues (Horner over window sums, affine normalisation, wire encodings).
// All O(N) work runs in the HIP kernels of msm.hip / ntt.hip / poly.hip / srs.hip / prover.hip; there is no CPU fallback for it.
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

// quads[w * nq + j] = the part of window w's sum that carries the factor 2^(4j) (msm_planes_window_quads):
//   out = sum_w 2^(c w) sum_j 2^(4 j) quads[w][j], one Horner pass from the top position down -- c (W - 1) + 4 (nq - 1) doublings, the same
//   dependent chain the W window sums needed (nq = 1 is host_horner: one value per window)
void host_quad_horner(g1_proj& out, const g1_proj* quads, uint32_t W, uint32_t c, uint32_t nq) {
  g1_proj acc = g1_identity();
  uint32_t prev = 0;
  bool first = true;
  for (uint32_t w = W; w-- > 0;)
    for (uint32_t j = nq; j-- > 0;) {
      const uint32_t pos = c * w + 4 * j;
      if (first) {
        acc = quads[(size_t)w * nq + j];
        first = false;
      } else {
        for (uint32_t d = pos; d < prev; d++) g1_double(acc, acc);
        g1_add(acc, acc, quads[(size_t)w * nq + j]);
      }
      prev = pos;
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
bool fr_bytes_to_mont(fr_t& out, const uint8_t* b32, int fmt) {
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
void fr_mont_to_bytes(uint8_t* b32, const fr_t& v, int fmt) {
  fr_t t = v;
  if (fmt == BP_FR_BYTES_LE) Fr::from_mont(t, v);
  memcpy(b32, &t, 32);
}

// upload n scalars to workspace `name`, converting to Montgomery form on the device if needed
int upload_fr(bp_ctx* ctx, const char* name, const void* host, size_t n, size_t cap_elems, int fmt, fr_t** out) {
  fr_t* d;
  BP_TRY(ws_get(ctx, name, std::max(cap_elems, n) * sizeof(fr_t), (void**)&d));
  if (n) BP_HIP(ctx, hipMemcpyAsync(d, host, n * sizeof(fr_t), hipMemcpyHostToDevice, ctx->stream));
  if (fmt == BP_FR_BYTES_LE) BP_TRY(fr_convert_run(ctx, d, n, 0));
  *out = d;
  return BP_OK;
}
int download_fr(bp_ctx* ctx, fr_t* d, void* host, size_t n, int fmt) {
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
void over_members(bp_ctx* ctx, size_t R, const std::function<bool(size_t)>& use, const std::function<void(size_t)>& work) {
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
void shard_range(size_t n, size_t r, size_t R, size_t* lo, size_t* hi) {
  const size_t base = n / R, extra = n % R;
  *lo = r * base + std::min(r, extra);
  *hi = *lo + base + (r < extra ? 1 : 0);
}
// the single-device contexts an entry point has to visit: the members of a group, or the context itself
std::vector<bp_ctx*> shards_of(bp_ctx* ctx) {
  if (is_group(ctx)) return ctx->members;
  return std::vector<bp_ctx*>(1, ctx);
}
// a member's failure is reported on the context the caller holds
int lift(bp_ctx* ctx, bp_ctx* member, int rc) {
  if (rc != BP_OK && member != ctx) ctx->last_error = member->last_error;
  return rc;
}

int ctx_create(bp_ctx** out, int device_id) {
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device_id < 0 || device_id >= count) return BP_ERR_NO_DEVICE;
  DeviceGuard guard(device_id);
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != device_id) return BP_ERR_NO_DEVICE;
  bp_ctx* ctx = new bp_ctx();
  ctx->device = device_id;
  hipError_t se;
  if (const uint32_t keep = knob_u32("BP_ACC_CU_KEEP", 0, 1, 31)) {
    // experiment: msm_accumulate on a stream whose CU mask leaves one CU in every (keep + 1) free for the other streams' kernels (a
    // RESERVATION, where priorities and generations only reorder what waits): hipExtStreamCreateWithCUMask, thinning pattern repeated
    // over 320 mask bits so that it does not depend on how the runtime numbers the CUs of the 8 XCDs
    uint32_t mask[10];
    for (int w = 0; w < 10; w++) {
      mask[w] = 0;
      for (int b = 0; b < 32; b++)
        if ((uint32_t)(w * 32 + b) % (keep + 1) != keep) mask[w] |= 1u << b;
    }
    se = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (se == hipSuccess) se = hipExtStreamCreateWithCUMask(&ctx->acc_stream, 10, mask);
    for (auto& e : ctx->acc_ev)
      if (se == hipSuccess) se = hipEventCreateWithFlags(&e, hipEventDisableTiming);
  } else if (knob_u32("BP_ACC_LOW_PRIORITY", 0, 0, 1)) {         // experiment: tails on high-priority streams, accumulations on low-priority ones
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);      // lo = least urgent (numerically greatest), hi = most urgent
    se = hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, hi);
    if (se == hipSuccess) se = hipStreamCreateWithPriority(&ctx->acc_stream, hipStreamNonBlocking, lo);
    for (auto& e : ctx->acc_ev)
      if (se == hipSuccess) se = hipEventCreateWithFlags(&e, hipEventDisableTiming);
  } else {
    se = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  }
  if (se != hipSuccess) {
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


const char* bp_version(void) { return EXPERIMENT_BUILD ? "bp_msm_ntt 0.4 (gfx950) +experiment" : "bp_msm_ntt 0.4 (gfx950)"; }

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
  comm_release(ctx);
  if (ctx->comm_host) (void)hipHostFree(ctx->comm_host);
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
  for (auto& e : ctx->acc_ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->acc_stream) (void)hipStreamDestroy(ctx->acc_stream);
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
