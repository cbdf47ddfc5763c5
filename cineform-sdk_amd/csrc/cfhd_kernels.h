// cfhd_kernels.h -- hand-written HIP kernels (gfx950) for the CineForm hot path.
//
//   forward:  k_fwd_yuv422  packed 8-bit 4:2:2 -> level-1 bands of Y, V, U (unpack + 2/6 analysis + quantize, fused)
//             k_fwd_plane   int16 plane        -> 4 bands (levels 2/3, and level 1 of planar formats), optional 2-bit prescale
//   inverse:  k_inv_plane   4 bands            -> int16 plane of the next lower level (plain / "descale" variant)
//             k_inv_yuv422  level-1 bands x3   -> packed 8-bit 4:2:2 (2/6 synthesis + 10->8 bit + interleave, fused)
//
// Shape of every kernel: one 256-thread workgroup owns a tile of 64 x 16 output coefficients (per band)
// of one job (= one channel plane of one frame; blockIdx.z walks the job table so a single launch covers
// every channel of every frame in flight).  The packed/planar input tile (with its 2/6 halo) is staged
// into LDS with coalesced dword loads, the horizontal pass runs LDS -> LDS on packed int16 pairs, the
// vertical pass + quantizer runs LDS -> HBM with 4-byte stores.  HBM traffic is therefore input once
// (+12.5 % row halo, +6 % column halo, mostly L2 hits) and output once: the kernels are bandwidth-bound
// stencils; there is no MFMA-shaped work here.
//
// Arithmetic follows the reference's SSE2 routines bit for bit (saturating int16 adds in the same
// association order; border taps in 32-bit):
//   Codec/spatial.c:253 FilterHorizontalRow16s, :3669 FilterHorizontalRow10bit16s, :10026 FilterSpatialQuant16s,
//   :12942 FilterSpatialV210Quant16s, :14726 FilterSpatialYUVQuant16s, Codec/quantize.c:1395 QuantizeRow16sTo16s,
//   Codec/convert.c:4667 UnpackRowYUV16s, Codec/spatial.c:21877 InvertSpatialQuant16s, :22414 InvertSpatialQuantDescale16s,
//   Codec/InvertHorizontalStrip16s.c:459/:1700/:3770/:5025, Codec/spatial.c:31341-31975 InvertSpatial*Row16sToOutput.
#pragma once
#include <stdint.h>
#if defined(CFHD_HIPEMU)
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif

namespace cfhd {
namespace dev {

enum { TW = 64, TH = 16, ROWS = 2 * TH + 4, NTHREADS = 256, NSTAGE = (ROWS * (TW + 4) + NTHREADS - 1) / NTHREADS };

struct QuantParam { int mid; unsigned mult; int divisor; };

struct FwdPlaneJob {
	const int16_t *in; int in_pitch;        // elements
	int width, height;                      // input plane
	int prescale;                           // 0 or 2
	int16_t *out[4]; int out_pitch;         // LL, LH, HL, HH
	QuantParam q[4];
};

struct FwdYuvJob {
	const uint8_t *in; int in_pitch;        // bytes
	int width, height, display_height;      // luma samples; rows >= display_height read as 0x80
	int uyvy, shift;
	int16_t *out[3][4]; int out_pitch[3];   // channel order Y, V, U (reference order)
	QuantParam q[3][4];
};

struct InvPlaneJob {
	const int16_t *band[4]; int band_pitch;
	int width, height;                      // band dimensions
	int descale;                            // 0, or 2 when the encoder prescaled this level
	int16_t *out; int out_pitch;
};

struct InvYuvJob {
	const int16_t *band[3][4]; int band_pitch[3];
	int width, height;                      // luma band dimensions (chroma bands are width/2)
	int display_height;                     // output rows
	int uyvy, shift;
	uint32_t dither_seed;                   // per frame; the kernel xors in its launch-wide seed
	uint8_t *out; int out_pitch;            // bytes
};

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int sat16(int x) { return x < -32768 ? -32768 : (x > 32767 ? 32767 : x); }
__device__ __forceinline__ int adds16(int a, int b) { return sat16(a + b); }
__device__ __forceinline__ int subs16(int a, int b) { return sat16(a - b); }
__device__ __forceinline__ int lo16(uint32_t v) { return (int)(int16_t)(v & 0xffffu); }
__device__ __forceinline__ int hi16(uint32_t v) { return (int)(int16_t)(v >> 16); }
__device__ __forceinline__ uint32_t pack16(int lo, int hi) { return ((uint32_t)(uint16_t)lo) | ((uint32_t)(uint16_t)hi << 16); }

// ---- packed 2 x int16 arithmetic: v_pk_add_i16 / v_pk_sub_i16 with clamp are exactly SSE2's _mm_adds_epi16 / _mm_subs_epi16 on two
// lanes, so the reference's saturating SIMD bodies map one to one onto CDNA4 packed math (half the VALU issue slots of the
// scalar form, saturation for free).  Under tests/hipemu the same operations are spelled out in scalar C.
#if defined(CFHD_HIPEMU)
__device__ __forceinline__ uint32_t pk_adds(uint32_t a, uint32_t b) { return pack16(adds16(lo16(a), lo16(b)), adds16(hi16(a), hi16(b))); }
__device__ __forceinline__ uint32_t pk_subs(uint32_t a, uint32_t b) { return pack16(subs16(lo16(a), lo16(b)), subs16(hi16(a), hi16(b))); }
__device__ __forceinline__ uint32_t pk_sra(uint32_t a, int n) { return pack16(lo16(a) >> n, hi16(a) >> n); }
__device__ __forceinline__ uint32_t pk_lolo(uint32_t a, uint32_t b) { return (a & 0xffffu) | (b << 16); }          // (a.lo, b.lo)
__device__ __forceinline__ uint32_t pk_hihi(uint32_t a, uint32_t b) { return (a >> 16) | (b & 0xffff0000u); }       // (a.hi, b.hi)
#else
typedef short cfhd_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_adds(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(cfhd_s2, a), __builtin_bit_cast(cfhd_s2, b))); }
__device__ __forceinline__ uint32_t pk_subs(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(cfhd_s2, a), __builtin_bit_cast(cfhd_s2, b))); }
__device__ __forceinline__ uint32_t pk_sra(uint32_t a, int n) { cfhd_s2 x = __builtin_bit_cast(cfhd_s2, a); x = x >> (short)n; return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t pk_lolo(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
__device__ __forceinline__ uint32_t pk_hihi(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
#endif
__device__ __forceinline__ uint32_t pk_set(int v) { return pack16(v, v); }

// 2/6 analysis highpass on two lanes at once (SIMD association order, spatial.c:326-397 / :10301-10351)
__device__ __forceinline__ uint32_t pk_hp_mid(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5)
{
	uint32_t s = pk_subs(0u, a0);
	s = pk_subs(s, a1);
	s = pk_adds(s, a4);
	s = pk_adds(s, a5);
	s = pk_adds(s, pk_set(4));
	s = pk_sra(s, 3);
	return pk_adds(s, pk_subs(a2, a3));
}

// quantize.c:1395: sign * (((|x| + mid) * mult) >> 16) with 16-bit wrap of |x|+mid
__device__ __forceinline__ int quantize(int v, const QuantParam &q)
{
	if (q.divisor <= 1) return v;
	int neg = v < 0;
	unsigned a = ((unsigned)(neg ? -v : v) + (unsigned)q.mid) & 0xffffu;
	int r = (int)((a * q.mult) >> 16);
	return (int)(int16_t)(neg ? -r : r);
}

// 2/6 analysis highpass, interior tap (SIMD association order, spatial.c:326-397 / :10301-10351)
__device__ __forceinline__ int hp_mid(int a0, int a1, int a2, int a3, int a4, int a5)
{
	int s = subs16(0, a0);
	s = subs16(s, a1);
	s = adds16(s, a4);
	s = adds16(s, a5);
	s = adds16(s, 4);
	s >>= 3;
	return adds16(s, subs16(a2, a3));
}
__device__ __forceinline__ int hp_first(int a0, int a1, int a2, int a3, int a4, int a5) { return sat16((5 * a0 - 11 * a1 + 4 * a2 + 4 * a3 - a4 - a5 + 4) >> 3); }
__device__ __forceinline__ int hp_last(int a0, int a1, int a2, int a3, int a4, int a5) { return sat16((11 * a4 - 5 * a5 - 4 * a3 - 4 * a2 + a1 + a0 + 4) >> 3); }

// Horizontal analysis of output columns c and c+1.  d[0..3] are the packed sample pairs (x[2k], x[2k+1]) for k = c-1 .. c+2,
// dm2 the pair k = c-2 (only read when column c is the last one).  Interior taps run on packed lanes; the border columns use
// the 6-tap border filters in 32 bits on x[0..5] and x[W-6..W-1] exactly as the reference's scalar code does.
__device__ __forceinline__ void horiz_pair(const uint32_t *d, uint32_t dm2, int prescale, bool first0, bool last0, bool last1, uint32_t &lpk, uint32_t &hpk)
{
	uint32_t p0 = d[0], p1 = d[1], p2 = d[2], p3 = d[3];
	if (prescale) {
		// FilterHorizontalRow10bit16s (spatial.c:3669): every tap sees (x + 3) >> 2, the lowpass is ((x0+3) + (x1+3) - 3) >> 2
		const uint32_t three = pk_set(3);
		p0 = pk_adds(p0, three); p1 = pk_adds(p1, three); p2 = pk_adds(p2, three); p3 = pk_adds(p3, three);
		uint32_t low = pk_adds(pk_lolo(p1, p2), pk_hihi(p1, p2));
		low = pk_subs(low, three);
		lpk = pk_sra(low, 2);
		p0 = pk_sra(p0, 2); p1 = pk_sra(p1, 2); p2 = pk_sra(p2, 2); p3 = pk_sra(p3, 2);
	} else {
		lpk = pk_adds(pk_lolo(p1, p2), pk_hihi(p1, p2));
	}
	const uint32_t em = pk_lolo(p0, p1), om = pk_hihi(p0, p1), e0 = pk_lolo(p1, p2), o0 = pk_hihi(p1, p2), ep = pk_lolo(p2, p3), op = pk_hihi(p2, p3);
	hpk = pk_hp_mid(em, om, e0, o0, ep, op);
	if (first0 || last0 || last1) {
		int h0 = lo16(hpk), h1 = hi16(hpk);
		if (first0) h0 = hp_first(lo16(p1), hi16(p1), lo16(p2), hi16(p2), lo16(p3), hi16(p3));
		if (last0) {
			uint32_t pm = dm2;
			if (prescale) pm = pk_sra(pk_adds(pm, pk_set(3)), 2);
			h0 = hp_last(lo16(pm), hi16(pm), lo16(p0), hi16(p0), lo16(p1), hi16(p1));
		}
		if (last1) h1 = hp_last(lo16(p0), hi16(p0), lo16(p1), hi16(p1), lo16(p2), hi16(p2));
		hpk = pack16(h0, h1);
	}
}

// Vertical analysis + quantizer for two adjacent columns held as packed pairs in LDS.
// sl/sh point at the window's first row for this column pair; stride in dwords. pos: 0 top, 1 middle, 2 bottom.
__device__ __forceinline__ void vert_pair_store(const uint32_t *sl, const uint32_t *sh, int stride, int pos, const QuantParam *q,
                                                int16_t *const *out, int out_pitch, int r, int c, bool valid1)
{
	uint32_t L[6], H[6];
#pragma unroll
	for (int k = 0; k < 6; k++) { L[k] = sl[k * stride]; H[k] = sh[k * stride]; }
	uint32_t ll, hl, lh, hh;
	if (pos == 1) {
		ll = pk_adds(L[2], L[3]); hl = pk_hp_mid(L[0], L[1], L[2], L[3], L[4], L[5]);
		lh = pk_adds(H[2], H[3]); hh = pk_hp_mid(H[0], H[1], H[2], H[3], H[4], H[5]);
	} else {
		int res[4][2];
#pragma unroll
		for (int e = 0; e < 2; e++) {
			int a[6], b[6];
#pragma unroll
			for (int k = 0; k < 6; k++) { a[k] = e ? hi16(L[k]) : lo16(L[k]); b[k] = e ? hi16(H[k]) : lo16(H[k]); }
			if (pos == 0) {
				res[0][e] = sat16(a[0] + a[1]); res[2][e] = hp_first(a[0], a[1], a[2], a[3], a[4], a[5]);
				res[1][e] = sat16(b[0] + b[1]); res[3][e] = hp_first(b[0], b[1], b[2], b[3], b[4], b[5]);
			} else {
				res[0][e] = sat16(a[4] + a[5]); res[2][e] = hp_last(a[0], a[1], a[2], a[3], a[4], a[5]);
				res[1][e] = sat16(b[4] + b[5]); res[3][e] = hp_last(b[0], b[1], b[2], b[3], b[4], b[5]);
			}
		}
		ll = pack16(res[0][0], res[0][1]); lh = pack16(res[1][0], res[1][1]); hl = pack16(res[2][0], res[2][1]); hh = pack16(res[3][0], res[3][1]);
	}
	lh = pack16(quantize(lo16(lh), q[1]), quantize(hi16(lh), q[1]));
	hl = pack16(quantize(lo16(hl), q[2]), quantize(hi16(hl), q[2]));
	hh = pack16(quantize(lo16(hh), q[3]), quantize(hi16(hh), q[3]));
	if (!valid1) { ll &= 0xffffu; lh &= 0xffffu; hl &= 0xffffu; hh &= 0xffffu; }
	const size_t o = (size_t)r * out_pitch + c;
	*(uint32_t *)(out[0] + o) = ll; *(uint32_t *)(out[1] + o) = lh; *(uint32_t *)(out[2] + o) = hl; *(uint32_t *)(out[3] + o) = hh;
}

// The job descriptor is copied into LDS once per workgroup: fields that are indexed with a per-lane channel number would
// otherwise be fetched with dependent vector loads from HBM inside the item loops.
template <typename J> __device__ __forceinline__ void stage_job(J *dst, const J *src)
{
	const uint32_t *s = (const uint32_t *)src; uint32_t *d = (uint32_t *)dst;
	for (int i = threadIdx.x; i < (int)(sizeof(J) / 4); i += NTHREADS) d[i] = s[i];
	__syncthreads();
}

__device__ __forceinline__ int window_first_row(int r, int half_height, int height) { return r == 0 ? 0 : (r == half_height - 1 ? height - 6 : 2 * r - 2); }
__device__ __forceinline__ int tile_first_row(int r0, int height) { int s = 2 * r0 - 2; if (s > height - 6) s = height - 6; return s < 0 ? 0 : s; }

// =============================================================================================
// Forward, int16 plane source
// =============================================================================================
__global__ void __launch_bounds__(NTHREADS) k_fwd_plane(const FwdPlaneJob *jobs)
{
	__shared__ FwdPlaneJob s_job;
	stage_job(&s_job, &jobs[blockIdx.z]);
	const FwdPlaneJob &job = s_job;
	const int W = job.width, H = job.height, HW = W >> 1, HH = H >> 1;
	const int c0 = blockIdx.x * TW, r0 = blockIdx.y * TH;
	__shared__ uint32_t s_in[ROWS][TW + 4];     // dword d of a row holds samples 2(c0-2+d), 2(c0-2+d)+1
	__shared__ uint32_t s_l[ROWS][TW / 2];
	__shared__ uint32_t s_h[ROWS][TW / 2];
	const bool active = (c0 < HW) && (r0 < HH);         // uniform per workgroup
	const int tid = threadIdx.x;
	const int row_start = tile_first_row(r0, H);
	if (active) {
		// all loads of the tile are issued before the first LDS store: one HBM round trip per workgroup instead of one per iteration
		uint32_t va[NSTAGE];
#pragma unroll
		for (int k = 0; k < NSTAGE; k++) {
			const int i = tid + k * NTHREADS;
			const int j = i / (TW + 4), d = i - j * (TW + 4);
			const int y = row_start + j, dw = c0 - 2 + d;    // dword index within the plane row
			va[k] = 0;
			if (i < ROWS * (TW + 4) && y < H && dw >= 0 && dw < HW) va[k] = *(const uint32_t *)(job.in + (size_t)y * job.in_pitch + 2 * dw);
		}
#pragma unroll
		for (int k = 0; k < NSTAGE; k++) {
			const int i = tid + k * NTHREADS;
			if (i < ROWS * (TW + 4)) { const int j = i / (TW + 4); s_in[j][i - j * (TW + 4)] = va[k]; }
		}
	}
	__syncthreads();
	if (active) {
		for (int i = tid; i < ROWS * (TW / 2); i += NTHREADS) {
			int j = i / (TW / 2), p = i - j * (TW / 2);
			int c = c0 + 2 * p;
			int y = row_start + j;
			if (c >= HW || y >= H) continue;
			// samples x[2c-2 .. 2c+5] = dwords (c-1 .. c+2) -> local d = 2p+1 .. 2p+4
			uint32_t d[4];
#pragma unroll
			for (int k = 0; k < 4; k++) d[k] = s_in[j][2 * p + 1 + k];
			uint32_t lpk, hpk;
			horiz_pair(d, s_in[j][2 * p], job.prescale, c == 0, c == HW - 1, c + 1 == HW - 1, lpk, hpk);
			s_l[j][p] = lpk;
			s_h[j][p] = hpk;
		}
	}
	__syncthreads();
	if (active) {
		for (int i = tid; i < TH * (TW / 2); i += NTHREADS) {
			int rl = i / (TW / 2), p = i - rl * (TW / 2);
			int r = r0 + rl, c = c0 + 2 * p;
			if (r >= HH || c >= HW) continue;
			int j = window_first_row(r, HH, H) - row_start;
			int pos = r == 0 ? 0 : (r == HH - 1 ? 2 : 1);
			vert_pair_store(&s_l[j][p], &s_h[j][p], TW / 2, pos, job.q, job.out, job.out_pitch, r, c, c + 1 < HW);
		}
	}
}

// =============================================================================================
// Forward level 1, packed 8-bit 4:2:2 source (all three channels of a tile in one workgroup)
// =============================================================================================
__global__ void __launch_bounds__(NTHREADS) k_fwd_yuv422(const FwdYuvJob *jobs)
{
	__shared__ FwdYuvJob s_job;
	stage_job(&s_job, &jobs[blockIdx.z]);
	const FwdYuvJob &job = s_job;
	const int W = job.width, H = job.height;          // luma samples
	const int DW = W >> 1;                            // dwords per row = luma output columns = chroma samples
	const int HH = H >> 1;
	const int c0 = blockIdx.x * TW, r0 = blockIdx.y * TH;   // luma output tile origin
	__shared__ uint32_t s_in[ROWS][TW + 4];           // dword d = packed pixel pair (c0 - 2 + d)
	__shared__ uint32_t s_l[ROWS][TW];                // [0,32) luma pairs, [32,48) V pairs, [48,64) U pairs
	__shared__ uint32_t s_h[ROWS][TW];
	const bool active = (c0 < DW) && (r0 < HH);
	const int tid = threadIdx.x;
	const int row_start = tile_first_row(r0, H);
	const int shift = job.shift;
	if (active) {
		uint32_t va[NSTAGE];
#pragma unroll
		for (int k = 0; k < NSTAGE; k++) {
			const int i = tid + k * NTHREADS;
			const int j = i / (TW + 4), d = i - j * (TW + 4);
			const int y = row_start + j, dw = c0 - 2 + d;
			va[k] = 0;
			if (i < ROWS * (TW + 4) && y < H && dw >= 0 && dw < DW)
				va[k] = (y < job.display_height) ? *(const uint32_t *)(job.in + (size_t)y * job.in_pitch + 4 * (size_t)dw) : 0x80808080u;
		}
#pragma unroll
		for (int k = 0; k < NSTAGE; k++) {
			const int i = tid + k * NTHREADS;
			if (i < ROWS * (TW + 4)) { const int j = i / (TW + 4); s_in[j][i - j * (TW + 4)] = va[k]; }
		}
	}
	__syncthreads();
	const int ysh0 = job.uyvy ? 8 : 0;                                 // byte lane of Y0 (Y1 is 16 bits further)
	const int ush = job.uyvy ? 0 : 8, vsh = job.uyvy ? 16 : 24;       // byte lanes of U, V
	if (active) {
		// luma: output column c <-> dword c ; needs dwords c-1 .. c+2 for the pair (c, c+1)
		for (int i = tid; i < ROWS * (TW / 2); i += NTHREADS) {
			int j = i / (TW / 2), p = i - j * (TW / 2);
			int c = c0 + 2 * p, y = row_start + j;
			if (c >= DW || y >= H) continue;
			// luma pair of dword v: (Y0, Y1) << shift, one AND + one shift per packed pair
			uint32_t d[4];
#pragma unroll
			for (int k = 0; k < 4; k++) d[k] = ((s_in[j][2 * p + 1 + k] >> ysh0) & 0x00ff00ffu) << shift;
			uint32_t lpk, hpk;
			horiz_pair(d, 0u, 0, c == 0, false, c + 1 == DW - 1, lpk, hpk);     // DW is even: c is never the last column
			s_l[j][p] = lpk;
			s_h[j][p] = hpk;
		}
		// chroma: output column cc <-> chroma samples 2cc, 2cc+1 = dwords 2cc, 2cc+1 ; pair (cc, cc+1) needs dwords 2cc-2 .. 2cc+5
		const int CW = DW >> 1;                       // chroma output columns
		for (int i = tid; i < ROWS * (TW / 4); i += NTHREADS) {
			int j = i / (TW / 4), p = i - j * (TW / 4);
			int cc = (c0 >> 1) + 2 * p, y = row_start + j;
			if (cc >= CW || y >= H) continue;
			// chroma samples k and k+1 sit in dwords 2cc-2+2k.. : pack (sample(2m), sample(2m+1)) from two dwords
			uint32_t du[4], dv[4];
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const uint32_t a0 = s_in[j][4 * p + 2 * k], a1 = s_in[j][4 * p + 2 * k + 1];
				du[k] = (((a0 >> ush) & 0xffu) | (((a1 >> ush) & 0xffu) << 16)) << shift;
				dv[k] = (((a0 >> vsh) & 0xffu) | (((a1 >> vsh) & 0xffu) << 16)) << shift;
			}
			const bool first = (cc == 0), last = (cc + 1 == CW - 1);       // CW is even
			uint32_t lpk, hpk;
			horiz_pair(dv, 0u, 0, first, false, last, lpk, hpk);
			s_l[j][TW / 2 + p] = lpk; s_h[j][TW / 2 + p] = hpk;
			horiz_pair(du, 0u, 0, first, false, last, lpk, hpk);
			s_l[j][TW / 2 + TW / 4 + p] = lpk; s_h[j][TW / 2 + TW / 4 + p] = hpk;
		}
	}
	__syncthreads();
	if (active) {
		const int CW = DW >> 1;
		for (int i = tid; i < TH * TW; i += NTHREADS) {
			int rl = i / TW, p = i - rl * TW;
			int r = r0 + rl;
			if (r >= HH) continue;
			int ch, c, limit;
			if (p < TW / 2) { ch = 0; c = c0 + 2 * p; limit = DW; }
			else if (p < TW / 2 + TW / 4) { ch = 1; c = (c0 >> 1) + 2 * (p - TW / 2); limit = CW; }
			else { ch = 2; c = (c0 >> 1) + 2 * (p - TW / 2 - TW / 4); limit = CW; }
			if (c >= limit) continue;
			int j = window_first_row(r, HH, H) - row_start;
			int pos = r == 0 ? 0 : (r == HH - 1 ? 2 : 1);
			vert_pair_store(&s_l[j][p], &s_h[j][p], TW, pos, job.q[ch], job.out[ch], job.out_pitch[ch], r, c, true);
		}
	}
}

// =============================================================================================
// Inverse
// =============================================================================================
// Vertical synthesis at band row r for one column: low = vertical-lowpass band (LL or LH), hi = HL/HH value.
// top/bottom borders in 32-bit (spatial.c:21975-22030, :22330-22400), interior in SIMD order (:22080-22150).
__device__ __forceinline__ void inv_vert(int lm1, int l0, int lp1, int lfar, int hi, int pos, int &even, int &odd)
{
	// pos 0 (top):    l0 = row 0, lp1 = row 1, lfar = row 2
	// pos 2 (bottom): l0 = row h-1, lm1 = row h-2, lfar = row h-3
	if (pos == 0) {
		even = sat16((((11 * l0 - 4 * lp1 + lfar + 4) >> 3) + hi) >> 1);
		odd = sat16((((5 * l0 + 4 * lp1 - lfar + 4) >> 3) - hi) >> 1);
	} else if (pos == 2) {
		even = sat16((((5 * l0 + 4 * lm1 - lfar + 4) >> 3) + hi) >> 1);
		odd = sat16((((11 * l0 - 4 * lm1 + lfar + 4) >> 3) - hi) >> 1);
	} else {
		int e = subs16(lm1, lp1); e = adds16(e, 4); e >>= 3; e = adds16(e, l0); e = adds16(e, hi); even = e >> 1;
		int o = subs16(0, lm1); o = adds16(o, lp1); o = adds16(o, 4); o >>= 3; o = adds16(o, l0); o = subs16(o, hi); odd = o >> 1;
	}
}

// Horizontal synthesis at column c, before the final >>1 / doubling (InvertHorizontalStrip16s.c:371-402, borders :172-198, :409-438).
__device__ __forceinline__ void inv_horiz(int lm1, int l0, int lp1, int lfar, int hi, int pos, int &even, int &odd)
{
	if (pos == 0) { even = ((11 * l0 - 4 * lp1 + lfar + 4) >> 3) + hi; odd = ((5 * l0 + 4 * lp1 - lfar + 4) >> 3) - hi; }
	else if (pos == 2) { even = ((5 * l0 + 4 * lm1 - lfar + 4) >> 3) + hi; odd = ((11 * l0 - 4 * lm1 + lfar + 4) >> 3) - hi; }
	else {
		int e = subs16(lm1, lp1); e = adds16(e, 4); e >>= 3; e = adds16(e, l0); even = adds16(e, hi);
		int o = subs16(lp1, lm1); o = adds16(o, 4); o >>= 3; o = adds16(o, l0); odd = subs16(o, hi);
	}
}

enum { ITW = 64, ITH = 8, ICOLS = ITW + 4 };    // inverse tile: 64 x 8 band coefficients -> 128 x 16 outputs

// The four taps of one vertical synthesis: l[0..2] = lowpass rows (r-1, r, r+1), or for the border rows (r, r+1, r+2) resp.
// (r-2, r-1, r); l[3] = the highpass value.  Zero outside [0,w) (never used by a valid tap).
struct VTaps { int l0, l1, l2, hi; };
__device__ __forceinline__ VTaps inv_vert_load(const int16_t *vlow, const int16_t *vhigh, int pitch, int w, int h, int r, int c)
{
	// branch-free: clamp the coordinates so that every lane issues its four loads back to back, then mask
	const bool valid = c >= 0 && c < w && r < h;
	const int cc = c < 0 ? 0 : (c >= w ? w - 1 : c), rr = r >= h ? h - 1 : r;
	const int first = rr == 0 ? 0 : (rr == h - 1 ? h - 3 : rr - 1);
	const int16_t *p = vlow + (size_t)first * pitch + cc;
	VTaps t;
	t.l0 = p[0]; t.l1 = p[pitch]; t.l2 = p[2 * (size_t)pitch];
	t.hi = vhigh[(size_t)rr * pitch + cc];
	const int mask = valid ? -1 : 0;                 // arithmetic select: no branch between the loads of consecutive items
	t.l0 &= mask; t.l1 &= mask; t.l2 &= mask; t.hi &= mask;
	return t;
}
__device__ __forceinline__ void inv_vert_apply(const VTaps &t, int r, int h, int &even, int &odd)
{
	const int pos = r == 0 ? 0 : (r == h - 1 ? 2 : 1);
	if (pos == 0) inv_vert(0, t.l0, t.l1, t.l2, t.hi, 0, even, odd);                 // rows 0,1,2
	else if (pos == 2) inv_vert(t.l1, t.l2, 0, t.l0, t.hi, 2, even, odd);           // rows h-3 (far), h-2, h-1
	else inv_vert(t.l0, t.l1, t.l2, 0, t.hi, 1, even, odd);
}

__global__ void __launch_bounds__(NTHREADS) k_inv_plane(const InvPlaneJob *jobs)
{
	__shared__ InvPlaneJob s_job;
	stage_job(&s_job, &jobs[blockIdx.z]);
	const InvPlaneJob &job = s_job;
	const int w = job.width, h = job.height;
	const int c0 = blockIdx.x * ITW, r0 = blockIdx.y * ITH;
	// vertical results for columns c0-2 .. c0+ITW+1 : [row parity][L/H][r][col]
	__shared__ int16_t s_v[2][2][ITH][ICOLS];
	const bool active = (c0 < w) && (r0 < h);
	const int tid = threadIdx.x;
	if (active) {
		enum { NIT = (ITH * ICOLS + NTHREADS - 1) / NTHREADS };
		VTaps tl[NIT], th[NIT];
#pragma unroll
		for (int k = 0; k < NIT; k++) {                  // every load of the tile first ...
			const int i = tid + k * NTHREADS, rl = i / ICOLS, cl = i - rl * ICOLS;
			const int r = i < ITH * ICOLS ? r0 + rl : h, c = c0 - 2 + cl;
			tl[k] = inv_vert_load(job.band[0], job.band[2], job.band_pitch, w, h, r, c);   // (LL, HL) -> horizontal-lowpass rows
			th[k] = inv_vert_load(job.band[1], job.band[3], job.band_pitch, w, h, r, c);   // (LH, HH) -> horizontal-highpass rows
		}
#pragma unroll
		for (int k = 0; k < NIT; k++) {                  // ... then the arithmetic and the LDS stores
			const int i = tid + k * NTHREADS, rl = i / ICOLS, cl = i - rl * ICOLS;
			const int r = r0 + rl;
			if (i >= ITH * ICOLS || r >= h) continue;
			int e, o;
			inv_vert_apply(tl[k], r, h, e, o);
			s_v[0][0][rl][cl] = (int16_t)e; s_v[1][0][rl][cl] = (int16_t)o;
			inv_vert_apply(th[k], r, h, e, o);
			s_v[0][1][rl][cl] = (int16_t)e; s_v[1][1][rl][cl] = (int16_t)o;
		}
	}
	__syncthreads();
	if (active) {
		for (int i = tid; i < 2 * ITH * ITW; i += NTHREADS) {
			int par = i / (ITH * ITW), rem = i - par * (ITH * ITW);
			int rl = rem / ITW, cl = rem - rl * ITW;
			int r = r0 + rl, c = c0 + cl;
			if (r >= h || c >= w) continue;
			const int16_t *L = &s_v[par][0][rl][cl + 2], *Hh = &s_v[par][1][rl][cl + 2];
			const int pos = c == 0 ? 0 : (c == w - 1 ? 2 : 1);
			int even, odd;
			if (pos == 0) inv_horiz(0, L[0], L[1], L[2], Hh[0], 0, even, odd);
			else if (pos == 2) inv_horiz(L[-1], L[0], 0, L[-2], Hh[0], 2, even, odd);
			else inv_horiz(L[-1], L[0], L[1], 0, Hh[0], 1, even, odd);
			if (job.descale) { even = sat16(even * 2); odd = sat16(odd * 2); }
			else { even = sat16(even >> 1); odd = sat16(odd >> 1); }
			*(uint32_t *)(job.out + (size_t)(2 * r + par) * job.out_pitch + 2 * c) = pack16(even, odd);
		}
	}
}

// 10 -> 8 bit reduction of one reconstructed sample v (= lowfilter +/- high, before the >>1):
// negative values clamp to zero first (the +2048 / subs_epu16 pair, InvertHorizontalStrip16s.c:4086-4089),
// dither 0/1 is added before the shift (:3869-3893, rand()&1 per SIMD lane in the reference), result clamps to 8 bits.
__device__ __forceinline__ uint32_t to8(int v, int shift, int dither)
{
	if (v < 0) v = 0;
	int x = ((v >> 1) + dither) >> shift;
	return (uint32_t)(x > 255 ? 255 : x);
}

// Counter-based stand-in for the reference's libc rand() dither: one bit per (frame seed, output row, lane of 16).
__device__ __forceinline__ int dither_bit(uint32_t seed, int row, int lane)
{
	uint32_t x = seed ^ ((uint32_t)row * 0x9E3779B1u) ^ ((uint32_t)lane * 0x85EBCA77u);
	x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
	return (int)(x & 1u);
}

__global__ void __launch_bounds__(NTHREADS) k_inv_yuv422(const InvYuvJob *jobs, uint32_t launch_seed)
{
	__shared__ InvYuvJob s_job;
	stage_job(&s_job, &jobs[blockIdx.z]);
	const InvYuvJob &job = s_job;
	const uint32_t seed = job.dither_seed ^ launch_seed;
	const int w = job.width, h = job.height;          // luma band ; chroma bands are w/2 wide
	const int c0 = blockIdx.x * ITW, r0 = blockIdx.y * ITH;
	__shared__ int16_t s_y[2][2][ITH][ICOLS];
	__shared__ int16_t s_c[2][2][2][ITH][ITW / 2 + 4]; // [V,U][parity][L/H][r][col]
	const bool active = (c0 < w) && (r0 < h);
	const int tid = threadIdx.x;
	const int cw = w >> 1, cc0 = c0 >> 1;
	if (active) {
		enum { NY = (ITH * ICOLS + NTHREADS - 1) / NTHREADS, CCOLS = ITW / 2 + 4, NC = (2 * ITH * CCOLS + NTHREADS - 1) / NTHREADS };
		VTaps yl[NY], yh[NY], cl_[NC], ch_[NC];
#pragma unroll
		for (int k = 0; k < NY; k++) {
			const int i = tid + k * NTHREADS, rl = i / ICOLS, cl = i - rl * ICOLS;
			const int r = i < ITH * ICOLS ? r0 + rl : h, c = c0 - 2 + cl;
			yl[k] = inv_vert_load(job.band[0][0], job.band[0][2], job.band_pitch[0], w, h, r, c);
			yh[k] = inv_vert_load(job.band[0][1], job.band[0][3], job.band_pitch[0], w, h, r, c);
		}
#pragma unroll
		for (int k = 0; k < NC; k++) {
			const int i = tid + k * NTHREADS, q = i / (ITH * CCOLS), rem = i - q * (ITH * CCOLS), rl = rem / CCOLS, cl = rem - rl * CCOLS;
			const int r = i < 2 * ITH * CCOLS ? r0 + rl : h, c = cc0 - 2 + cl, chn = (q & 1) + 1;
			cl_[k] = inv_vert_load(job.band[chn][0], job.band[chn][2], job.band_pitch[chn], cw, h, r, c);
			ch_[k] = inv_vert_load(job.band[chn][1], job.band[chn][3], job.band_pitch[chn], cw, h, r, c);
		}
#pragma unroll
		for (int k = 0; k < NY; k++) {
			const int i = tid + k * NTHREADS, rl = i / ICOLS, cl = i - rl * ICOLS, r = r0 + rl;
			if (i >= ITH * ICOLS || r >= h) continue;
			int e, o;
			inv_vert_apply(yl[k], r, h, e, o);
			s_y[0][0][rl][cl] = (int16_t)e; s_y[1][0][rl][cl] = (int16_t)o;
			inv_vert_apply(yh[k], r, h, e, o);
			s_y[0][1][rl][cl] = (int16_t)e; s_y[1][1][rl][cl] = (int16_t)o;
		}
#pragma unroll
		for (int k = 0; k < NC; k++) {
			const int i = tid + k * NTHREADS, q = i / (ITH * CCOLS), rem = i - q * (ITH * CCOLS), rl = rem / CCOLS, cl = rem - rl * CCOLS, r = r0 + rl;
			if (i >= 2 * ITH * CCOLS || r >= h) continue;
			int e, o;
			inv_vert_apply(cl_[k], r, h, e, o);
			s_c[q][0][0][rl][cl] = (int16_t)e; s_c[q][1][0][rl][cl] = (int16_t)o;
			inv_vert_apply(ch_[k], r, h, e, o);
			s_c[q][0][1][rl][cl] = (int16_t)e; s_c[q][1][1][rl][cl] = (int16_t)o;
		}
	}
	__syncthreads();
	if (active) {
		// one item = one chroma band column = 2 chroma samples, 2 luma band columns = 4 luma samples = 8 output bytes
		for (int i = tid; i < 2 * ITH * (ITW / 2); i += NTHREADS) {
			int par = i / (ITH * (ITW / 2)), rem = i - par * (ITH * (ITW / 2));
			int rl = rem / (ITW / 2), cl = rem - rl * (ITW / 2);
			int r = r0 + rl, cc = cc0 + cl;
			int orow = 2 * r + par;
			if (r >= h || cc >= cw || orow >= job.display_height) continue;
			int yv[4], uv[2], vv[2];
#pragma unroll
			for (int k = 0; k < 2; k++) {
				int c = 2 * cc + k;
				const int16_t *L = &s_y[par][0][rl][2 * cl + k + 2], *Hh = &s_y[par][1][rl][2 * cl + k + 2];
				const int pos = c == 0 ? 0 : (c == w - 1 ? 2 : 1);
				if (pos == 0) inv_horiz(0, L[0], L[1], L[2], Hh[0], 0, yv[2 * k], yv[2 * k + 1]);
				else if (pos == 2) inv_horiz(L[-1], L[0], 0, L[-2], Hh[0], 2, yv[2 * k], yv[2 * k + 1]);
				else inv_horiz(L[-1], L[0], L[1], 0, Hh[0], 1, yv[2 * k], yv[2 * k + 1]);
			}
			{
				const int pos = cc == 0 ? 0 : (cc == cw - 1 ? 2 : 1);
#pragma unroll
				for (int k = 0; k < 2; k++) {
					const int16_t *L = &s_c[k][par][0][rl][cl + 2], *Hh = &s_c[k][par][1][rl][cl + 2];
					int *dst = k == 0 ? vv : uv;
					if (pos == 0) inv_horiz(0, L[0], L[1], L[2], Hh[0], 0, dst[0], dst[1]);
					else if (pos == 2) inv_horiz(L[-1], L[0], 0, L[-2], Hh[0], 2, dst[0], dst[1]);
					else inv_horiz(L[-1], L[0], L[1], 0, Hh[0], 1, dst[0], dst[1]);
				}
			}
			const int sh = job.shift;
			uint32_t px[2];
#pragma unroll
			for (int k = 0; k < 2; k++) {
				// byte lanes within a 16-byte output group decide the dither lane (the reference uses one random bit per SIMD lane)
				int lane = ((4 * cc + 2 * k) & 7) * 2;
				uint32_t y0 = to8(yv[2 * k], sh, sh >= 2 ? dither_bit(seed, orow, lane) : 0);
				uint32_t y1 = to8(yv[2 * k + 1], sh, sh >= 2 ? dither_bit(seed, orow, lane + 1) : 0);
				uint32_t u = to8(uv[k], sh, sh >= 2 ? dither_bit(seed, orow, 16 + ((2 * cc + k) & 7)) : 0);
				uint32_t v = to8(vv[k], sh, sh >= 2 ? dither_bit(seed, orow, 24 + ((2 * cc + k) & 7)) : 0);
				px[k] = job.uyvy ? (u | (y0 << 8) | (v << 16) | (y1 << 24)) : (y0 | (u << 8) | (y1 << 16) | (v << 24));
			}
			uint2 o2; o2.x = px[0]; o2.y = px[1];
			*(uint2 *)(job.out + (size_t)orow * job.out_pitch + 8 * (size_t)cc) = o2;
		}
	}
}

} // namespace dev
} // namespace cfhd
