// cfhd_gfx950.h -- the gfx950 primitives the kernels are written in: wave-level data movement on the DPP / readlane path, packed 16-bit
// arithmetic, address-space qualified loads.  Included as <cfhd_gfx950.h>: the CPU test build (tests/hipemu) puts its own header of the
// same name first on the include path and gets the same kernels in scalar C; nothing of that lives in this directory.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace cfhd {
namespace dev {

__device__ __forceinline__ uint32_t atomic_or_u32(uint32_t *p, uint32_t v) { return atomicOr(p, v); }

// ---- wave64
#define CFHD_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
__device__ __forceinline__ int wave_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ int wave_lane() { return (int)(threadIdx.x & 63u); }
// Inclusive prefix sum over the 64 lanes on the DPP data path (no LDS crossbar round trips): Hillis-Steele inside the rows of 16 lanes
// (row_shr 1, 2, 4, 8; lanes without a source add 0), then lane 15 of every row into rows 1 and 3 (row_bcast:15, row mask 0xa) and
// lane 31 into rows 2 and 3 (row_bcast:31, row mask 0xc).  All 64 lanes must be active.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x)
{
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
	return x;
}
// lanes below this one whose bit is set in a ballot mask
__device__ __forceinline__ uint32_t wave_mbcnt(unsigned long long mask) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)); }
__device__ __forceinline__ uint32_t wave_get(uint32_t x, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)x, lane); }                       // uniform lane index
__device__ __forceinline__ uint32_t wave_read(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(lane)); }
// A pointer every lane of the wave holds the same value of, moved to scalar registers: the per-lane part of an address is then one
// 32-bit offset (global_load ... v_off, s[base]) instead of a 64-bit pointer per band.
template <typename T> __device__ __forceinline__ T *wave_uniform_ptr(T *p)
{
	const uint64_t v = (uint64_t)p;
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
	return (T *)(((uint64_t)hi << 32) | lo);
}

// ---- packed 2 x int16 arithmetic: v_pk_add_i16 / v_pk_sub_i16 with clamp are exactly SSE2's _mm_adds_epi16 / _mm_subs_epi16 on two
// lanes, so the reference's saturating SIMD bodies map one to one onto CDNA4 packed math (half the VALU issue slots of the
// scalar form, saturation for free).
typedef short cfhd_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_adds(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(cfhd_s2, a), __builtin_bit_cast(cfhd_s2, b))); }
__device__ __forceinline__ uint32_t pk_subs(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(cfhd_s2, a), __builtin_bit_cast(cfhd_s2, b))); }
__device__ __forceinline__ uint32_t pk_sra(uint32_t a, int n) { cfhd_s2 x = __builtin_bit_cast(cfhd_s2, a); x = x >> (short)n; return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t pk_lolo(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }          // (a.lo, b.lo)
__device__ __forceinline__ uint32_t pk_hihi(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }          // (a.hi, b.hi)
// wrapping (non-saturating) packed add / negate and signed max: the quantizer's 16-bit arithmetic (quantize.c:1395)
__device__ __forceinline__ uint32_t pk_addw(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (cfhd_s2)(__builtin_bit_cast(cfhd_s2, a) + __builtin_bit_cast(cfhd_s2, b))); }
// wrapping packed multiply, low halves of the products (v_pk_mul_lo_u16): value x divisor in the reference's 16-bit PIXEL arithmetic, two coefficients at a time
typedef unsigned short cfhd_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_mulw(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (cfhd_us2)(__builtin_bit_cast(cfhd_us2, a) * __builtin_bit_cast(cfhd_us2, b))); }
__device__ __forceinline__ uint32_t pk_negw(uint32_t a) { return __builtin_bit_cast(uint32_t, (cfhd_s2)(-__builtin_bit_cast(cfhd_s2, a))); }
__device__ __forceinline__ uint32_t pk_maxs(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(cfhd_s2, a), __builtin_bit_cast(cfhd_s2, b))); }
// 10 -> 8 bits on two lanes: clamp at zero, halve, add the dither bit, >> shift, saturate (v_pk_max_i16 / v_pk_ashrrev_i16 / v_pk_min_i16)
__device__ __forceinline__ uint32_t pk_to8(uint32_t v, int shift, uint32_t dither)
{
	cfhd_s2 x = __builtin_bit_cast(cfhd_s2, v);
	const cfhd_s2 zero = { 0, 0 }, top = { 255, 255 };
	x = __builtin_elementwise_max(x, zero);
	x = (x >> (short)1) + __builtin_bit_cast(cfhd_s2, dither);
	x = x >> (short)shift;
	x = __builtin_elementwise_min(x, top);
	return __builtin_bit_cast(uint32_t, x);
}
__device__ __forceinline__ uint32_t byte_perm(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
// 10 -> 8 bits of four neighbouring samples: e = (s0, s2), o = (s1, s3) as 16-bit lanes before the last >> 1, d2e / d2o = twice the dither bit of each lane (bit 1, bit 17),
// shift1 = shift + 1 (<= 7).  Equal to pk_to8 on both words, bytes in sample order: ((v >> 1) + d) >> shift = (v + 2 d) >> (shift + 1) for v >= 0, a negative v stays <= 0
// through the saturating add and the arithmetic shift, and v_sat_pk_u8_i16 clamps each lane to 0 .. 255 while packing -- three instructions a word instead of five, one v_perm
// instead of a shift and an or.
__device__ __forceinline__ uint32_t sat_pk_u8(uint32_t v) { uint32_t r; asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(v)); return r; }
__device__ __forceinline__ uint32_t pk_to8_bytes(uint32_t e, uint32_t o, int shift1, uint32_t d2e, uint32_t d2o)
{
	e = pk_sra(pk_adds(e, d2e), shift1); o = pk_sra(pk_adds(o, d2o), shift1);
	return __builtin_amdgcn_perm(sat_pk_u8(o), sat_pk_u8(e), 0x05010400u);
}
__device__ __forceinline__ uint32_t rotr32(uint32_t w, uint32_t n) { return __builtin_amdgcn_alignbit(w, w, n); }   // n < 32
// product of the low 24 bits (v_mul_u32_u24, full rate; the compiler turns a plain 32-bit multiply whose low half alone is used into the
// quarter-rate v_mul_lo_u32): the low 16 bits equal those of the full product, which is all the 16-bit coefficient arithmetic keeps
__device__ __forceinline__ uint32_t mul_u24(uint32_t a, uint32_t b) { uint32_t r; asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// ---- loads through the global address space (global_load_dword): a flat load would tick lgkmcnt as well and every LDS access in
// between would drain the loads in flight
typedef uint32_t cfhd_u4 __attribute__((ext_vector_type(4)));
#define CFHD_LDG32(p) (*(const __attribute__((address_space(1))) uint32_t *)(p))
typedef uint32_t cfhd_u2 __attribute__((ext_vector_type(2)));
#define CFHD_LDG64(p) (*(const __attribute__((address_space(1))) cfhd_u2 *)(p))
#define CFHD_LDG128(p) (*(const __attribute__((address_space(1))) cfhd_u4 *)(p))
// ... and 16-byte stores through it (global_store_dwordx4 with a scalar base when the pointer's base is wave-uniform; a generic pointer gives flat_store)
__device__ __forceinline__ void store_u32x4_global(void *at, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { cfhd_u4 v; v.x = a; v.y = b; v.z = c; v.w = d; *(__attribute__((address_space(1))) cfhd_u4 *)at = v; }

// two consecutive dwords in one store (global_store_dwordx2) at an address that is only dword aligned: gfx950 serves unaligned vector accesses to global memory
typedef uint32_t cfhd_u2_a4 __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ void store_u32x2_dword_aligned(uint32_t *at, uint32_t a, uint32_t b) { cfhd_u2_a4 v; v.x = a; v.y = b; *(__attribute__((address_space(1))) cfhd_u2_a4 *)at = v; }

} // namespace dev
} // namespace cfhd
