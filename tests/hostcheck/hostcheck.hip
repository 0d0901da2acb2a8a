// hostcheck.hip -- TEST-ONLY host build of the product's __host__ __device__ arithmetic headers
// (baby_plonk_rust_amd/csrc/{bigint,fields,g1}.cuh) so their logic can be checked against the
// oracle on a machine without a GPU.  Never linked into the product library.
#define BP_HOST_USE_DEVICE_ALGO 1   /* run the 32-bit column multiplier (the device algorithm) on the CPU */
#include "../../baby_plonk_rust_amd/csrc/g1.cuh"
#include <string.h>
using namespace bp;

extern "C" {
void hc_fr_mul(uint32_t* r, const uint32_t* a, const uint32_t* b) { fr_t x, y, z; memcpy(&x, a, 32); memcpy(&y, b, 32); Fr::mul(z, x, y); memcpy(r, &z, 32); }
void hc_fr_add(uint32_t* r, const uint32_t* a, const uint32_t* b) { fr_t x, y, z; memcpy(&x, a, 32); memcpy(&y, b, 32); Fr::add(z, x, y); memcpy(r, &z, 32); }
void hc_fr_sub(uint32_t* r, const uint32_t* a, const uint32_t* b) { fr_t x, y, z; memcpy(&x, a, 32); memcpy(&y, b, 32); Fr::sub(z, x, y); memcpy(r, &z, 32); }
void hc_fr_neg(uint32_t* r, const uint32_t* a) { fr_t x, z; memcpy(&x, a, 32); Fr::neg(z, x); memcpy(r, &z, 32); }
void hc_fr_inv(uint32_t* r, const uint32_t* a) { fr_t x, z; memcpy(&x, a, 32); fr_invert(z, x); memcpy(r, &z, 32); }
void hc_fr_from_mont(uint32_t* r, const uint32_t* a) { fr_t x, z; memcpy(&x, a, 32); Fr::from_mont(z, x); memcpy(r, &z, 32); }
void hc_fp_mul(uint32_t* r, const uint32_t* a, const uint32_t* b) { fp_t x, y, z; memcpy(&x, a, 48); memcpy(&y, b, 48); Fp::mul(z, x, y); memcpy(r, &z, 48); }
void hc_fp_add(uint32_t* r, const uint32_t* a, const uint32_t* b) { fp_t x, y, z; memcpy(&x, a, 48); memcpy(&y, b, 48); Fp::add(z, x, y); memcpy(r, &z, 48); }
void hc_fp_sub(uint32_t* r, const uint32_t* a, const uint32_t* b) { fp_t x, y, z; memcpy(&x, a, 48); memcpy(&y, b, 48); Fp::sub(z, x, y); memcpy(r, &z, 48); }
void hc_fp_neg(uint32_t* r, const uint32_t* a) { fp_t x, z; memcpy(&x, a, 48); Fp::neg(z, x); memcpy(r, &z, 48); }
void hc_fp_inv(uint32_t* r, const uint32_t* a) { fp_t x, z; memcpy(&x, a, 48); fp_invert(z, x); memcpy(r, &z, 48); }
void hc_g1_add(uint32_t* r, const uint32_t* a, const uint32_t* b) { g1_proj x, y, z; memcpy(&x, a, 144); memcpy(&y, b, 144); g1_add(z, x, y); memcpy(r, &z, 144); }
void hc_g1_double(uint32_t* r, const uint32_t* a) { g1_proj x, z; memcpy(&x, a, 144); g1_double(z, x); memcpy(r, &z, 144); }
void hc_g1_add_mixed(uint32_t* r, const uint32_t* a, const uint32_t* b) { g1_proj x, z; g1_affine y; memcpy(&x, a, 144); memcpy(&y, b, 96); g1_add_mixed(z, x, y); memcpy(r, &z, 144); }
void hc_g1_mul_scalar(uint32_t* r, const uint32_t* a, const uint32_t* k) { g1_proj x, z; fr_t s; memcpy(&x, a, 144); memcpy(&s, k, 32); g1_mul_scalar(z, x, s); memcpy(r, &z, 144); }
void hc_g1_mul_small(uint32_t* r, const uint32_t* a, uint32_t k, int nbits) { g1_proj x, z; memcpy(&x, a, 144); g1_mul_small(z, x, k, nbits); memcpy(r, &z, 144); }
void hc_g1_to_affine(uint32_t* r, const uint32_t* a) { g1_proj x; memcpy(&x, a, 144); g1_affine z = g1_to_affine(x); memcpy(r, &z, 96); }
}
