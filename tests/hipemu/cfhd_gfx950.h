// tests/hipemu/cfhd_gfx950.h -- TEST INFRASTRUCTURE ONLY: the scalar-C twin of cineform-sdk_amd/csrc/cfhd_gfx950.h.  The emulated test build
// puts this directory first on its include path, so that the kernel headers' #include <cfhd_gfx950.h> finds these definitions and the same
// kernel source runs on the CPU (hip_emu.h: fibers for threads, counting barriers, wave exchanges).
#pragma once
#include <stdint.h>
#include "hip_emu.h"

namespace cfhd {
namespace dev {

inline uint32_t atomic_or_u32(uint32_t *p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }

#define CFHD_WAVE_SYNC() hipemu::wave_sync()
inline int wave_uniform(int x) { return x; }
inline int wave_lane() { return (int)hipemu::lane_id(); }
inline uint32_t wave_incl_scan(uint32_t x)
{
	const int lane = wave_lane();
	for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, (unsigned)d); if (lane >= d) x += y; }
	return x;
}
inline uint32_t wave_mbcnt(unsigned long long mask) { return (uint32_t)__builtin_popcountll(mask & ((1ull << hipemu::lane_id()) - 1ull)); }
inline uint32_t wave_get(uint32_t x, int lane) { return __shfl(x, lane); }
inline uint32_t wave_read(uint32_t v, int lane) { return __shfl(v, lane); }
template <typename T> inline T *wave_uniform_ptr(T *p) { return p; }

namespace emu16 {
inline int sat(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }
inline int lo(uint32_t v) { return (int)(int16_t)(v & 0xffffu); }
inline int hi(uint32_t v) { return (int)(int16_t)(v >> 16); }
inline uint32_t pack(int l, int h) { return ((uint32_t)(uint16_t)l) | ((uint32_t)(uint16_t)h << 16); }
inline uint32_t to8(int v, int shift, int dither) { if (v < 0) v = 0; const int x = ((v >> 1) + dither) >> shift; return (uint32_t)(x > 255 ? 255 : x); }
}
inline uint32_t pk_adds(uint32_t a, uint32_t b) { using namespace emu16; return pack(sat(lo(a) + lo(b)), sat(hi(a) + hi(b))); }
inline uint32_t pk_subs(uint32_t a, uint32_t b) { using namespace emu16; return pack(sat(lo(a) - lo(b)), sat(hi(a) - hi(b))); }
inline uint32_t pk_sra(uint32_t a, int n) { using namespace emu16; return pack(lo(a) >> n, hi(a) >> n); }
inline uint32_t pk_lolo(uint32_t a, uint32_t b) { return (a & 0xffffu) | (b << 16); }
inline uint32_t pk_hihi(uint32_t a, uint32_t b) { return (a >> 16) | (b & 0xffff0000u); }
inline uint32_t pk_addw(uint32_t a, uint32_t b) { using namespace emu16; return pack(lo(a) + lo(b), hi(a) + hi(b)); }
inline uint32_t pk_mulw(uint32_t a, uint32_t b) { return ((a & 0xffffu) * (b & 0xffffu) & 0xffffu) | (((a >> 16) * (b >> 16) & 0xffffu) << 16); }
inline uint32_t pk_negw(uint32_t a) { using namespace emu16; return pack(-lo(a), -hi(a)); }
inline uint32_t pk_maxs(uint32_t a, uint32_t b) { using namespace emu16; return pack(lo(a) > lo(b) ? lo(a) : lo(b), hi(a) > hi(b) ? hi(a) : hi(b)); }
inline uint32_t pk_to8(uint32_t v, int shift, uint32_t dither) { using namespace emu16; return to8(lo(v), shift, (int)(dither & 1u)) | (to8(hi(v), shift, (int)((dither >> 16) & 1u)) << 16); }
inline uint32_t pk_to8_bytes(uint32_t e, uint32_t o, int shift1, uint32_t d2e, uint32_t d2o)
{
	using namespace emu16;
	const int s = shift1 - 1;
	return to8(lo(e), s, (int)((d2e >> 1) & 1u)) | (to8(lo(o), s, (int)((d2o >> 1) & 1u)) << 8) | (to8(hi(e), s, (int)((d2e >> 17) & 1u)) << 16) | (to8(hi(o), s, (int)((d2o >> 17) & 1u)) << 24);
}
inline uint32_t rotr32(uint32_t w, uint32_t n) { n &= 31u; return n ? (w >> n) | (w << (32u - n)) : w; }
inline uint32_t mul_u24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }
inline uint32_t byte_perm(uint32_t s0, uint32_t s1, uint32_t sel)
{
	const uint64_t src = ((uint64_t)s0 << 32) | s1;
	uint32_t r = 0;
	for (int k = 0; k < 4; k++) { const uint32_t b = (sel >> (8 * k)) & 0xffu; r |= (b < 8 ? (uint32_t)((src >> (8 * b)) & 0xffu) : (b == 0x0c ? 0u : 0xffu)) << (8 * k); }
	return r;
}

struct emu_u4 { uint32_t x, y, z, w; };
typedef emu_u4 cfhd_u4;
#define CFHD_LDG32(p) (*(const uint32_t *)(p))
struct emu_u2 { uint32_t x, y; };
typedef emu_u2 cfhd_u2;
#define CFHD_LDG64(p) (*(const cfhd::dev::cfhd_u2 *)(p))
#define CFHD_LDG128(p) (*(const cfhd::dev::cfhd_u4 *)(p))
inline void store_u32x2_dword_aligned(uint32_t *at, uint32_t a, uint32_t b) { at[0] = a; at[1] = b; }
inline void store_u32x4_global(void *at, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { uint32_t *p = (uint32_t *)at; p[0] = a; p[1] = b; p[2] = c; p[3] = d; }

} // namespace dev
} // namespace cfhd
