// fr_io.hpp -- 16-byte-vector global loads/stores of Fr elements (32 B = 2 x dwordx4 per lane).
#pragma once
#include "fields.hpp"

namespace bp {

__device__ __forceinline__ fr_t load_fr(const fr_t* __restrict__ p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  fr_t r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  return r;
}
__device__ __forceinline__ void store_fr(fr_t* __restrict__ p, const fr_t& v) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
}  // namespace bp
