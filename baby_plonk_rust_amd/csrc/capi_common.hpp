// capi_common.hpp -- helpers shared by the capi_*.hip translation units (not part of the C ABI, not installed).
#pragma once
#include <functional>
#include <vector>

#include "ctx.hpp"

inline bool fmt_ok(int fmt) { return fmt == BP_FR_BYTES_LE || fmt == BP_FR_MONT; }
inline bool basis_ok(int b) { return b == BP_BASIS_LAGRANGE || b == BP_BASIS_MONOMIAL; }
// capi_ctx.hip
bool fr_bytes_to_mont(bp::fr_t& out, const uint8_t* b32, int fmt);
void fr_mont_to_bytes(uint8_t* b32, const bp::fr_t& v, int fmt);
int upload_fr(bp_ctx* ctx, const char* name, const void* host, size_t n, size_t cap_elems, int fmt, bp::fr_t** out);
int download_fr(bp_ctx* ctx, bp::fr_t* d, void* host, size_t n, int fmt);
// work(r) for every member r for which use(r) holds: member 0 on the calling thread, the others on their own threads, all at once
void over_members(bp_ctx* ctx, size_t R, const std::function<bool(size_t)>& use, const std::function<void(size_t)>& work);
void shard_range(size_t n, size_t r, size_t R, size_t* lo, size_t* hi);
std::vector<bp_ctx*> shards_of(bp_ctx* ctx);
int lift(bp_ctx* ctx, bp_ctx* member, int rc);
int ctx_create(bp_ctx** out, int device_id);
// capi_msm.hip
int srs_find(bp_ctx* ctx, uint64_t handle, bp::SrsEntry** out);
// capi_ntt.hip
bool host_root_of_unity(bp::fr_t& out, uint64_t group_order);
