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
#include <cfhd_gfx950.h>

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
	// k_fwd_packed16 only: the plane is one component of interleaved unsigned 16-bit pixels (RG48, b64a): `in` points at the
	// component's first word, samples are xstride words apart, >> shift brings them to the codec precision, rows beyond
	// display_height repeat the last picture row (frame.c:6020-6024)
	int xstride, shift, display_height;
	int compand;                            // alpha plane of b64a: 0 < a < 4095 -> ((a * 223 + 128) >> 8) + 256 (frame.c:6696-6707)
	// v210 (10-bit 4:2:2, three samples per 32-bit word; convert.c:3968 ConvertV210RowToPlanar16s): layout 1 = luma, 2 = channel 1 (Cr),
	// 3 = channel 2 (Cb); `in` is the start of the frame, rows beyond display_height are zero (frame.c:1481 stops there), and from sample
	// tail_from on -- what the reference's 48-pixel SIMD loop leaves to its scalar loop -- channel 1 repeats the first Cr of every group of
	// three (the scalar loop stores `v` before it has read the next one, convert.c:4530-4535).  layout 0: everything above.
	// layout 4 / 5: 8-bit interleaved pixels, bottom / top row first (RG24, BGRA / BGRa; frame.c:6173 ConvertRGBtoRGB48, :6286 ConvertRGBAtoRGB48):
	// `in` is the start of the frame, in_pitch is in BYTES, xstride = bytes per pixel, tail_from = the component's byte inside the pixel;
	// sample = byte << 4 (compand: the alpha byte of a 4:4:4:4 encode, curved as b64a's); rows beyond display_height zero.
	// layout 6: 10-bit RGB in one 32-bit word per pixel (r210, DPX0: big-endian; AB10, AR10: little-endian; wavelet.c:3595): in_pitch in 32-bit
	// words, xstride = 1 for big-endian words, tail_from = bit position of the component; sample = field << 2; rows beyond display_height repeat
	// the last row (a choice: the reference's fused row pipeline treats them its own way, parity is claimed for heights that are multiples of 8).
	// layout 8 / 9: 8-bit interleaved pixels (bottom / top row first) converted to one plane of a 10-bit 4:2:2 frame on the way in (frame.c:378; RG24 /
	// BGRA / BGRa encoded as YUV 4:2:2): xstride = bytes per pixel, tail_from = the plane (0 Y, 1 v, 2 u), shift = the matrix as for layout 7.
	// layout 7: deep RGB converted to one plane of a 10-bit 4:2:2 frame on the way in (Codec/frame.c:6731 ConvertAnyDeep444to422; RG48 / b64a encoded
	// as YUV 4:2:2): `in` = the R word of the first pixel (G, B behind it), xstride = words per pixel, tail_from = the plane (0 Y, 1 channel 1 = v,
	// 2 channel 2 = u), shift = colour space (0 computer-systems 709, 1 video 709, 2 computer 601, 3 video 601); a chroma sample is the mean of
	// its pixel pair; rows beyond display_height repeat the last row.
	// layout 10 / 11: one component plane of a Bayer mosaic computed on the way in (BYR4: 16-bit photosites through the encode curve; BYR5: the packed 12-bit
	// rows, no curve -- what k_unpack_byr4 writes as planes, without the planes): `in` = the frame, in_pitch = words per mosaic row (BYR4), xstride = pixel order
	// (BAYER_FORMAT_*), tail_from = the plane (0 G, 1 R-G, 2 B-G, 3 G1-G2), shift = precision, width / display_height those of the component planes.
	int layout, tail_from;
	const uint16_t *curve;                  // layout 10: encode curve over 14-bit linear input
};

// One sample of plane `which` (0 Y, 1 v, 2 u) from the deep RGB pixels at p (luma: pixel x; chroma: pixels 2x, 2x + 1): the reference's integer
// matrices, arithmetic shifts of the signed sums, clamps to 10 bits (frame.c:6803-6870, :7040-7170).
__device__ __forceinline__ uint32_t rgb16_to_yuv_sample(const uint16_t *p, int wpp, int which, int color_space, int x)
{
	const int m[4][10] = { { 2998, 10060, 1016, 64, 1655, 5538, 7193, 7193, 6537, 655 }, { 3490, 11715, 1180, 0, 1917, 6455, 8372, 8372, 7602, 770 },
	                       { 4211, 8258, 1606, 64, 2425, 4768, 7193, 7193, 6029, 1163 }, { 4899, 9617, 1868, 0, 2818, 5554, 8372, 8372, 7012, 1360 } };
	const int cs = color_space & 3;
	if (which == 0) {
		const uint16_t *q = p + (size_t)x * wpp;
		const int y = ((m[cs][0] * (int)q[0] + m[cs][1] * (int)q[1] + m[cs][2] * (int)q[2]) >> 20) + m[cs][3];
		return (uint32_t)(y < 0 ? 0 : (y > 1023 ? 1023 : y));
	}
	int acc = 0;
#pragma unroll
	for (int k = 0; k < 2; k++) {
		const uint16_t *q = p + (size_t)(2 * x + k) * wpp;
		const int r = q[0], g = q[1], b = q[2];
		acc += which == 2 ? (-m[cs][4] * r - m[cs][5] * g + m[cs][6] * b) >> 20 : (m[cs][7] * r - m[cs][8] * g - m[cs][9] * b) >> 20;
	}
	acc = (acc >> 1) + 512;
	return (uint32_t)(acc < 0 ? 0 : (acc > 1023 ? 1023 : acc));
}

// One sample of plane `which` (0 Y, 1 v, 2 u) from 8-bit pixels (bytes B, G, R(, A)) encoded as 4:2:2 (frame.c:378 ConvertRGB32to10bitYUVFrame): the
// 13-bit coefficients of RGB2YUV.c:1404 on 15-bit samples (byte << 7), every product shifted down 16 on its own, the sum << 2 plus the 14-bit
// offset clamped to 14 bits; 10 bits of it.  Chroma is the EVEN pixel's, not the pair's mean (RGB2YUV.c:846-873 keeps the low word of every
// 32-bit pair; its averaging tail starts at width & ~15, and coded widths are multiples of 16 here).
__device__ __forceinline__ uint32_t rgb8_to_yuv_sample(const uint8_t *row, int bpp, int which, int color_space, int x)
{
	const int m[4][9] = { { 1499, 5029, 507, 827, 2768, 3596, 3596, 3268, 327 }, { 1744, 5857, 589, 958, 3227, 4186, 4186, 3801, 385 },
	                      { 2105, 4128, 802, 1212, 2383, 3596, 3596, 3014, 581 }, { 2449, 4808, 933, 1409, 2777, 4186, 4186, 3506, 679 } };
	const int cs = color_space & 3;
	const uint8_t *q = row + (size_t)(which ? 2 * x : x) * bpp;
	const int b = (int)q[0] << 7, g = (int)q[1] << 7, r = (int)q[2] << 7;
	int v;
	if (which == 0) v = ((((m[cs][0] * r) >> 16) + ((m[cs][1] * g) >> 16) + ((m[cs][2] * b) >> 16)) << 2) + ((cs & 1) ? 0 : 1024);
	else if (which == 2) v = ((((-m[cs][3] * r) >> 16) + ((-m[cs][4] * g) >> 16) + ((m[cs][5] * b) >> 16)) * 4) + 8192;
	else v = ((((m[cs][6] * r) >> 16) + ((-m[cs][7] * g) >> 16) + ((-m[cs][8] * b) >> 16)) * 4) + 8192;
	v = v < 0 ? 0 : (v > 16383 ? 16383 : v);
	return (uint32_t)v >> 4;
}

// One sample of component plane job.tail_from of a Bayer frame (FwdPlaneJob::layout 10 / 11), the arithmetic of k_unpack_byr4.
__device__ __forceinline__ uint32_t bayer_plane_sample(const FwdPlaneJob &job, int row, int x)
{
	int tl, tr, bl, br;
	if (job.layout == 11) {
		const uint8_t *base = (const uint8_t *)job.in + (size_t)row * job.width * 6, *nib = base + (size_t)job.width * 4;
		int v[4];
#pragma unroll
		for (int k = 0; k < 4; k++) { const int s = k * job.width + x; v[k] = ((int)base[s] << 4) | ((nib[s >> 1] >> (4 * (s & 1))) & 15); }
		tl = v[0]; tr = v[1]; bl = v[2]; br = v[3];
	} else {
		const uint16_t *l1 = (const uint16_t *)job.in + (size_t)(2 * row) * job.in_pitch, *l2 = l1 + job.in_pitch;
		const uint32_t a = *(const uint32_t *)(l1 + 2 * x), b = *(const uint32_t *)(l2 + 2 * x);
		tl = job.curve[(a & 0xffffu) >> 2]; tr = job.curve[a >> 18]; bl = job.curve[(b & 0xffffu) >> 2]; br = job.curve[b >> 18];
	}
	int r, g1, g2, bb;
	switch (job.xstride) {
	case 0: r = tl; g1 = tr; g2 = bl; bb = br; break;
	case 1: g1 = tl; r = tr; bb = bl; g2 = br; break;
	case 3: bb = tl; g1 = tr; g2 = bl; r = br; break;
	default: g1 = tl; bb = tr; r = bl; g2 = br; break;
	}
	const int mid = 1 << (job.shift - 1), g = (g1 + g2) >> 1;
	const int v = job.tail_from == 0 ? g : (job.tail_from == 1 ? ((r - g) >> 1) + mid : (job.tail_from == 2 ? ((bb - g) >> 1) + mid : (g1 - g2 + 2 * mid) >> 1));
	return (uint32_t)(uint16_t)v;
}

// Block lists (k_fwd_yuv422_strip_blocks -> k_ent_count_blocks, cfhd_entropy_kernels.h): the entropy coder wants the nonzero coefficients of the quantized
// level-1 bands, about one in eleven.  Instead of storing those bands densely and having k_ent_count read all of them back (4.2 GB per 512 1080p frames), the
// strip kernel -- which holds every band row in registers as blocks of 8 coefficients per lane -- stores only the blocks that are not all zero, compacted per
// (band row, wave) = "chunk" of FWD_CHUNK_BLOCKS blocks, plus the chunk's 64-bit occupancy mask.  blocks[]: one 16-byte slot per block of the pyramid, indexed by
// (element offset of the chunk's first coefficient) / 8 + rank of the block among the chunk's nonzero blocks; masks[]: one word per chunk.
enum { FWD_CHUNK_BLOCKS = 62, FWD_CHUNK_COLS = FWD_CHUNK_BLOCKS * 8 };
struct FwdBlockLists {
	uint4 *blocks;                          // null: off (the bands are stored densely only)
	unsigned long long *masks;              // this frame's chunk masks
	const int16_t *base;                    // this frame's pyramid: out[c][b] - base = element offset of a band
	int mask_base[3][4];                    // first chunk of band (c, b) in masks[] (chunks by band row, then by position in the row)
	int reserved[2];
};

struct FwdYuvJob {
	const uint8_t *in; int in_pitch;        // bytes
	int width, height, display_height;      // luma samples; rows >= display_height read as 0x80
	int uyvy, shift;
	int16_t *out[3][4]; int out_pitch[3];   // channel order Y, V, U (reference order)
	QuantParam q[3][4];
	FwdBlockLists lists;
};

struct InvPlaneJob {
	const int16_t *band[4]; int band_pitch;
	int width, height;                      // band dimensions
	int descale;                            // 0, or 2 when the encoder prescaled this level
	int16_t *out; int out_pitch;
	// k_inv_packed16 only: `out` is the component's first word inside interleaved unsigned 16-bit pixels, xstride words per pixel,
	// out_pitch in words; samples are clamped to `precision` bits and shifted up to 16; rows >= display_height are not written
	int xstride, precision, display_height;
	int alpha;                              // k_inv_packed16: this component is the companded alpha plane of an RGBA 4:4:4:4 sample
	int alpha_const;                        // k_inv_packed16, RGB 4:4:4 samples decoded to b64a (orc_inv_spatial_to_b64a_of_rgb444): word 0 of every pixel is this constant
	                                        // (0xfff0 in the reference) and only the last band column takes the scalar-tail clamp
	// k_inv_packed16, 8-bit RGB output (RG24, BGRA, BGRa of RGB 4:4:4 samples): `out` is the component's BYTE inside the first pixel, xstride =
	// bytes per pixel, out_pitch in BYTES; every byte = (12-bit component * 2 + 9 + r) >> 5 with a four-bit dither r per sample (the
	// reference's model, oracle/cfhd_oracle_inv.c orc_inv_spatial_to_rgb8); bottom_up: picture row y goes to output row display_height - 1 - y;
	// the fourth byte of four-byte pixels is 255.  bytes8 == 2: BGRA / BGRa of RGBA 4:4:4:4 samples (orc_inv_spatial_to_rgba8): no dither, every byte
	// = (12-bit component + 2) >> 4; the alpha plane (`alpha`) is expanded from the rounded value: ((a + 2 - 256) << 3) * 9400 >> 16 >> 4 (codec.h:164-165)
	int bytes8, bottom_up;
	uint32_t dither_seed;
	// k_inv_rgb10 (r210 / DPX0 / AB10 / AR10 output of RGB 4:4:4 samples: one 32-bit word per pixel; orc_inv_spatial_to_rgb10): bit position of this
	// plane's 10 bits, words stored byte-swapped; `out` = the frame, out_pitch in 32-bit words
	int bit_shift, big_endian;
	// k_inv_plane, two-frame groups: the wavelet goes through the reference's InvertSpatialQuantOverflowProtected16s (Codec/spatial.c:21114; the unprescaled
	// wavelets of a group, wavelet.c:5759 / :5886), whose border filter of the LAST coefficient row reads the LL band one row too high -- rows h-2, h-3, h-4 instead
	// of h-1, h-2, h-3: the row pointer's advance behind the middle rows is compiled out (spatial.c:21770-21776 `#if (0 && XMMOPT)`).  1: reproduce it (what the
	// reference decoder's pictures are made of: 42 dB instead of 47 on the bottom 16 rows' account), 0: the filter as meant.
	int ll_bottom_row_high;
};

struct InvYuvJob {
	const int16_t *band[3][4]; int band_pitch[3];
	int width, height;                      // luma band dimensions (chroma bands are width/2)
	int display_height;                     // output rows
	int uyvy, shift;
	uint32_t dither_seed;                   // per frame; the kernel xors in its launch-wide seed
	uint8_t *out; int out_pitch;            // bytes
	// k_inv_yuv422_rgb32: BGRA (bottom row first) / BGRa output of a 4:2:2 sample -- the reference's fused horizontal pass + 8-bit colour conversion
	// (Codec/spatial.c:29577 InvertHorizontalStripYUV16sToPackedRGB32, restated in oracle/cfhd_oracle_inv.c orc_inv_spatial_to_rgb32_of_yuv422): no dither,
	// matrix 601 or 709 (computer-systems range), bytes B, G, R, 255
	int bottom_up, matrix_601;
	// k_inv_yuv422_strip_blocks: the level-1 highpass bands arrive as block lists from the entropy decoder's tile pass (cfhd_core.h dec_block_list_layout): of every chunk
	// of 64 blocks of a band's flat raster the nonzero blocks, compacted at the chunk's first places in the band, and masks[mask_base[c][b] + chunk] says which they are
	const unsigned long long *masks; int mask_base[3][4];
};

struct HalfYuvJob {                         // k_half_yuv422: the level-1 lowpass planes as a half-resolution packed 8-bit 4:2:2 frame
	const int16_t *ll[3]; int pitch[3];     // LL1 of Y, V, U
	int width, rows;                        // luma band columns (= output pixels per row), output rows
	int uyvy;
	uint8_t *out; int out_pitch;            // bytes
	int matrix;                             // k_half_rgb24: 0 computer-systems 709, 1 video 709, 2 computer 601, 3 video 601 (as k_yu64_to_rgb24)
	// k_half_rgb24, the other RGB outputs of a 4:2:2 sample at half resolution: 0 RG24 (bottom row first); 1 BGRA / BGRa bytes B, G, R, 255 (the SSE2 loop of
	// frame.c:8504's RGB32 branch: half widths that are multiples of 16); 2 RG48 words R, G, B; 3 b64a words 0xffff, R, G, B (frame.c:9567 ConvertLowpass16sYUVtoRGB48)
	int mode, bottom_up;
};

struct HalfPackedJob {                      // k_half_packed16: level-1 lowpass planes of a 4:4:4(:4) sample as half-resolution 16-bit pixels
	const int16_t *ll[4]; int pitch;        // LL1 of the component planes (G, R, B[, A])
	int width, rows, nch;                   // band columns (= output pixels per row), output rows, components
	int word[4];                            // word of plane c inside the pixel
	int shift, alpha;                       // left shift to 16 bits (16 - precision - 2); alpha: plane 3 is the companded alpha of b64a
	uint16_t *out; int out_pitch;           // bytes
	// k_half_rgb (the other outputs of RGB 4:4:4 samples): mode 1 = 8-bit pixels B G R (A = 255), 2 = one 10-bit RGB word (word[c] = bit position of plane c),
	// 3 = b64a words with the constant alpha 65535
	int mode, bytes, bottom_up, big_endian; uint32_t dither_seed;
};

struct FwdFrameJob {                        // k_fwd_frame_yuv422: interlaced level 1 of a packed 8-bit 4:2:2 frame
	const uint8_t *in; int in_pitch;        // bytes
	int width, height, display_height;      // luma samples, picture rows; rows >= display_height read as 0x80
	int uyvy, shift;
	int16_t *out[3][4]; int out_pitch[3];   // Y, V, U: LL, LH, HL, HH (one band row per pair of picture rows)
	QuantParam q[3][4];                     // q[c][2] is the difference-coded HL band: its midpoint is divisor / prequant without the "-1" (spatial.c:5360-5363)
	FwdBlockLists lists;                    // (unused: the two level-1 jobs share one table)
};

struct BayerJob {                           // k_unpack_byr4: 16-bit Bayer mosaic -> component planes G, R-G, B-G, G1-G2
	const uint16_t *in; int in_pitch;       // words per mosaic row
	int width, height, display_height;      // component plane (half the mosaic); rows >= display_height repeat the last quad row
	int16_t *out[4]; int out_pitch;
	const uint16_t *curve;                  // encode curve over 14-bit linear input (log 90 by default)
	int order;                              // BAYER_FORMAT_*: 0 R G / G B, 1 G R / B G, 2 G B / R G, 3 B G / G R (CFHDMetadataTags.h:72-75)
	int precision;
	// BYR5 (frame.c:5473 ConvertBYR5ToFrame16s): `in` is a byte stream; per row pair 4 x width samples of 12 bits -- first their high bytes as four runs of `width`
	// (for order 0: R, G1, G2, B), then their low nibbles, two to a byte (the even sample's in the low half) -- and no curve: the values are used as they are
	int packed12;
};

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int sat16(int x) { return x < -32768 ? -32768 : (x > 32767 ? 32767 : x); }
__device__ __forceinline__ int adds16(int a, int b) { return sat16(a + b); }
__device__ __forceinline__ int subs16(int a, int b) { return sat16(a - b); }
__device__ __forceinline__ int lo16(uint32_t v) { return (int)(int16_t)(v & 0xffffu); }
__device__ __forceinline__ int hi16(uint32_t v) { return (int)(int16_t)(v >> 16); }
__device__ __forceinline__ uint32_t pack16(int lo, int hi) { return ((uint32_t)(uint16_t)lo) | ((uint32_t)(uint16_t)hi << 16); }

// packed 2 x int16 arithmetic (pk_adds, pk_subs, pk_sra, ...): cfhd_gfx950.h
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
	int r = (int)((a * (q.mult & 0xffffu)) >> 16);          // (mult < 2^16: cfhd_device.hip quant_param; said here, the product is the full-rate v_mul_u32_u24)
	return (int)(int16_t)(neg ? -r : r);
}

// the same on two packed values: |x| as an unsigned 16-bit number (wrapping negate, so -32768 -> 32768), + mid with 16-bit wrap,
// the 16 x 16 -> high 16 multiply per half, sign restored with (r ^ s) - s
__device__ __forceinline__ uint32_t pk_quantize(uint32_t v, const QuantParam &q)
{
	if (q.divisor <= 1) return v;
	const uint32_t s = pk_sra(v, 15);
	uint32_t a = pk_maxs(v, pk_negw(v));
	a = pk_addw(a, pk_set(q.mid));
	// both factors below 2^16, and said so: the compiler then multiplies with the full-rate v_mul_u32_u24 (halves selected by SDWA) instead of the quarter-rate
	// v_mul_lo_u32, and one v_perm_b32 collects the two high halves -- three instructions for what took eight issue slots' worth of multiplies and four of masks
	const uint32_t m = q.mult & 0xffffu;
	const uint32_t r = pk_hihi((a & 0xffffu) * m, (a >> 16) * m);
	return pk_addw(r ^ s, pk_negw(s));
}

// a quantizer every lane of the wave shares, moved to scalar registers: the `divisor <= 1` test becomes a scalar branch, mid and mult scalar operands
__device__ __forceinline__ QuantParam wave_uniform_quant(const QuantParam &q) { QuantParam u; u.mid = wave_uniform(q.mid); u.mult = (unsigned)wave_uniform((int)q.mult); u.divisor = wave_uniform(q.divisor); return u; }

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

// XCD-aware tile order.  Workgroup b of a launch runs on XCD b % 8 (observed dispatch order, used for speed only) and every XCD has
// its own L2, so in launch order horizontally adjacent tiles -- which share halo columns, i.e. whole cache lines of every band
// row -- would sit behind different L2s and fetch those lines from HBM twice.  The logical tile index gives each XCD one
// contiguous run of tiles (x fastest, then y, then the job) instead; neighbours then meet in the same L2 close in time.
struct TileId { int x, y, z; };
__device__ __forceinline__ TileId xcd_tile()
{
	const unsigned gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
	const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
	const unsigned q = total >> 3, r = total & 7u, xcd = lin & 7u, j = lin >> 3;
	const unsigned logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
	TileId t;
	t.x = (int)(logical % gx); t.y = (int)((logical / gx) % gy); t.z = (int)(logical / (gx * gy));
	return t;
}

__device__ __forceinline__ int window_first_row(int r, int half_height, int height) { return r == 0 ? 0 : (r == half_height - 1 ? height - 6 : 2 * r - 2); }
__device__ __forceinline__ int tile_first_row(int r0, int height) { int s = 2 * r0 - 2; if (s > height - 6) s = height - 6; return s < 0 ? 0 : s; }

// =============================================================================================
// Forward, int16 plane source
// =============================================================================================
// PACKED: level 1 of the 4:4:4(:4) formats straight from the interleaved 16-bit pixels (ConvertRGB48ToFrame16s frame.c:5968 /
// ConvertBGRA64ToFrame_4444_16s :6569 + FilterSpatialQuant16s).  gridDim.x = tiles_x * nch: the nch component planes of one tile
// are consecutive logical tiles, i.e. they run close together on one XCD and the pixel rows they share come out of its L2.
// One 10-bit sample of a v210 row (groups of six pixels in four words: Cb0 Y0 Cr0 | Y1 Cb1 Y2 | Cr1 Y3 Cb2 | Y4 Cr2 Y5, low bits first)
__device__ __forceinline__ uint32_t v210_sample(const uint32_t *row, int layout, int x, int tail_from)
{
	int g, word, shift;
	if (layout == 1) {
		g = x / 6; const int r = x - 6 * g;
		word = (0x332110 >> (4 * r)) & 15; shift = (0x2805500a >> (5 * r)) & 31;          // words 0 1 1 2 3 3, shifts 10 0 20 10 0 20
	} else {
		g = x / 3; int r = x - 3 * g;
		if (layout == 2) { if (x >= tail_from && r) r--; word = (0x320 >> (4 * r)) & 15; shift = (0x2814 >> (5 * r)) & 31; }   // Cr: words 0 2 3, shifts 20 0 10
		else { word = r; shift = 10 * r; }                                                                                  // Cb: words 0 1 2, shifts 0 10 20
	}
	return (row[4 * g + word] >> shift) & 0x3ffu;
}

template <bool PACKED>
__device__ __forceinline__ void fwd_plane_tile(const FwdPlaneJob *jobs, int nch)
{
	TileId tile = xcd_tile();
	int comp = 0;
	if (PACKED) { comp = tile.x % nch; tile.x /= nch; }
	__shared__ FwdPlaneJob s_job;
	stage_job(&s_job, &jobs[PACKED ? tile.z * nch + comp : tile.z]);
	const FwdPlaneJob &job = s_job;
	const int W = job.width, H = job.height, HW = W >> 1, HH = H >> 1;
	const int c0 = tile.x * TW, r0 = tile.y * TH;
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
			if (i < ROWS * (TW + 4) && y < H && dw >= 0 && dw < HW) {
				if (PACKED && job.layout >= 10) {
					const int yy = y < job.display_height ? y : job.display_height - 1;
					va[k] = bayer_plane_sample(job, yy, 2 * dw) | (bayer_plane_sample(job, yy, 2 * dw + 1) << 16);
				} else if (PACKED && job.layout == 7) {
					const int yy = y < job.display_height ? y : job.display_height - 1;
					const uint16_t *row = (const uint16_t *)job.in + (size_t)yy * job.in_pitch;
					va[k] = rgb16_to_yuv_sample(row, job.xstride, job.tail_from, job.shift, 2 * dw) | (rgb16_to_yuv_sample(row, job.xstride, job.tail_from, job.shift, 2 * dw + 1) << 16);
				} else if (PACKED && job.layout >= 8) {
					if (y < job.display_height) {
						const uint8_t *row = (const uint8_t *)job.in + (size_t)(job.layout == 8 ? job.display_height - 1 - y : y) * job.in_pitch;
						va[k] = rgb8_to_yuv_sample(row, job.xstride, job.tail_from, job.shift, 2 * dw) | (rgb8_to_yuv_sample(row, job.xstride, job.tail_from, job.shift, 2 * dw + 1) << 16);
					} else va[k] = job.tail_from ? 0x02000200u : 0x00400040u;       // rows below the picture: Y 64, chroma 512 (frame.c:466-500)
				} else if (PACKED && job.layout == 6) {
					const int yy = y < job.display_height ? y : job.display_height - 1;
					const uint32_t *row = (const uint32_t *)job.in + (size_t)yy * job.in_pitch;
					uint32_t p0 = row[2 * dw], p1 = row[2 * dw + 1];
					if (job.xstride) { p0 = __builtin_bswap32(p0); p1 = __builtin_bswap32(p1); }
					va[k] = (((p0 >> job.tail_from) & 0x3ffu) << 2) | (((p1 >> job.tail_from) & 0x3ffu) << 18);
				} else if (PACKED && job.layout >= 4) {
					if (y < job.display_height) {
						const uint8_t *row = (const uint8_t *)job.in + (size_t)(job.layout == 4 ? job.display_height - 1 - y : y) * job.in_pitch + job.tail_from;
						uint32_t s0 = (uint32_t)row[(size_t)(2 * dw) * job.xstride] << 4, s1 = (uint32_t)row[(size_t)(2 * dw + 1) * job.xstride] << 4;
						if (job.compand) {           // alpha of BGRA / BGRa encoded as 4:4:4:4 (frame.c:6415 ConvertRGBAtoRGBA64: the open interval ends at 255 << 4)
							if (s0 > 0 && s0 < 4080) s0 = ((s0 * 223 + 128) >> 8) + 256;
							if (s1 > 0 && s1 < 4080) s1 = ((s1 * 223 + 128) >> 8) + 256;
						}
						va[k] = s0 | (s1 << 16);
					}
				} else if (PACKED && job.layout) {
					if (y < job.display_height) {
						const uint32_t *row = (const uint32_t *)((const uint16_t *)job.in + (size_t)y * job.in_pitch);
						va[k] = v210_sample(row, job.layout, 2 * dw, job.tail_from) | (v210_sample(row, job.layout, 2 * dw + 1, job.tail_from) << 16);
					}
				} else if (PACKED) {
					const int yy = y < job.display_height ? y : job.display_height - 1;
					const uint16_t *px = (const uint16_t *)job.in + (size_t)yy * job.in_pitch + (size_t)(2 * dw) * job.xstride;
					uint32_t s0 = (uint32_t)px[0] >> job.shift, s1 = (uint32_t)px[job.xstride] >> job.shift;
					if (job.compand) {
						if (s0 > 0 && s0 < 4095) s0 = ((s0 * 223 + 128) >> 8) + 256;
						if (s1 > 0 && s1 < 4095) s1 = ((s1 * 223 + 128) >> 8) + 256;
					}
					va[k] = s0 | (s1 << 16);
				} else va[k] = *(const uint32_t *)(job.in + (size_t)y * job.in_pitch + 2 * dw);
			}
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

__global__ void __launch_bounds__(NTHREADS) k_fwd_plane(const FwdPlaneJob *jobs) { fwd_plane_tile<false>(jobs, 1); }
__global__ void __launch_bounds__(NTHREADS) k_fwd_packed16(const FwdPlaneJob *jobs, int nch) { fwd_plane_tile<true>(jobs, nch); }

// =============================================================================================
// Forward level 1, packed 8-bit 4:2:2 source (all three channels of a tile in one workgroup)
// =============================================================================================
__global__ void __launch_bounds__(NTHREADS) k_fwd_yuv422(const FwdYuvJob *jobs)
{
	const TileId tile = xcd_tile();
	__shared__ FwdYuvJob s_job;
	stage_job(&s_job, &jobs[tile.z]);
	const FwdYuvJob &job = s_job;
	const int W = job.width, H = job.height;          // luma samples
	const int DW = W >> 1;                            // dwords per row = luma output columns = chroma samples
	const int HH = H >> 1;
	const int c0 = tile.x * TW, r0 = tile.y * TH;   // luma output tile origin
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

// Inverse tile: 64 x 16 band coefficients -> 128 x 32 outputs.  The four bands of the tile are staged in LDS as dwords (two
// adjacent columns each, columns c0-2 .. c0+ITW+1 = IDW dwords per row; the vertical-lowpass bands with one row above and one
// below, ILROWS rows), the vertical pass runs on column pairs with packed saturating math and leaves its results in LDS, the
// horizontal pass combines three neighbouring dwords per output pair.
enum { ITW = 64, ITH = 16, IDW = ITW / 2 + 2, ILROWS = ITH + 2 };

__device__ __forceinline__ int inv_tile_first_row(int r0, int h, int extra = 0) { int s = r0 - 1; if (s > h - 3 - extra) s = h - 3 - extra; return s < 0 ? 0 : s; }
__device__ __forceinline__ int inv_window_first_row(int r, int h) { return r == 0 ? 0 : (r == h - 1 ? h - 3 : r - 1); }

// Vertical synthesis of two adjacent columns: a, b, c = three consecutive rows of the vertical-lowpass band starting at
// inv_window_first_row(r), hi = the vertical-highpass row r.  Interior rows in the reference's SIMD order (spatial.c:22080-22150),
// the first and the last row of the band in 32-bit arithmetic (:21975-22030, :22330-22400).
__device__ __forceinline__ void inv_vert_pk(uint32_t a, uint32_t b, uint32_t c, uint32_t hi, int pos, uint32_t &even, uint32_t &odd)
{
	if (pos == 1) {
		uint32_t e = pk_subs(a, c); e = pk_adds(e, pk_set(4)); e = pk_sra(e, 3); e = pk_adds(e, b); e = pk_adds(e, hi); even = pk_sra(e, 1);
		uint32_t o = pk_subs(0u, a); o = pk_adds(o, c); o = pk_adds(o, pk_set(4)); o = pk_sra(o, 3); o = pk_adds(o, b); o = pk_subs(o, hi); odd = pk_sra(o, 1);
	} else {
		int e0, o0, e1, o1;
		if (pos == 0) { inv_vert(0, lo16(a), lo16(b), lo16(c), lo16(hi), 0, e0, o0); inv_vert(0, hi16(a), hi16(b), hi16(c), hi16(hi), 0, e1, o1); }
		else { inv_vert(lo16(b), lo16(c), 0, lo16(a), lo16(hi), 2, e0, o0); inv_vert(hi16(b), hi16(c), 0, hi16(a), hi16(hi), 2, e1, o1); }
		even = pack16(e0, e1); odd = pack16(o0, o1);
	}
}

// Horizontal synthesis of two lanes at once, interior columns, before the final >>1 / doubling (InvertHorizontalStrip16s.c:371-402).
__device__ __forceinline__ void inv_horiz_pk(uint32_t lm1, uint32_t l0, uint32_t lp1, uint32_t hi, uint32_t &even, uint32_t &odd)
{
	uint32_t e = pk_subs(lm1, lp1); e = pk_adds(e, pk_set(4)); e = pk_sra(e, 3); e = pk_adds(e, l0); even = pk_adds(e, hi);
	uint32_t o = pk_subs(lp1, lm1); o = pk_adds(o, pk_set(4)); o = pk_sra(o, 3); o = pk_adds(o, l0); odd = pk_subs(o, hi);
}

// First / last column of a row (32-bit arithmetic, InvertHorizontalStrip16s.c:172-198, :409-438): l[] = six consecutive columns,
// l[idx] = the border column itself.
__device__ __forceinline__ void inv_horiz_border(const int *l, int idx, int hi, int pos, int &even, int &odd)
{
	if (pos == 0) inv_horiz(0, l[idx], l[idx + 1], l[idx + 2], hi, 0, even, odd);
	else inv_horiz(l[idx - 1], l[idx], 0, l[idx - 2], hi, 2, even, odd);
}

// Loads of one staged region (two bands b0, b1 of the same geometry), all issued before the first LDS store: item i ->
// (band q, row j, dword d).  The band pointers are passed in registers and the loads go through the global address space
// (global_load_dword): a flat load would tick lgkmcnt as well and every LDS access in between would drain the loads in flight.
template <int NROWS, int NDW, int N>
__device__ __forceinline__ void inv_stage_load(uint32_t (&va)[N], const int16_t *b0, const int16_t *b1, int pitch, int row0, int h, int dw0, int wdw)
{
	// branch-free: every lane loads from clamped (always valid) coordinates, what lies outside the band is zeroed afterwards
#pragma unroll
	for (int k = 0; k < N; k++) {
		const int i = (int)threadIdx.x + k * NTHREADS;
		const int q = i / (NROWS * NDW), rem = i - q * (NROWS * NDW), j = rem / NDW, d = rem - j * NDW;
		int row = row0 + j, dw = dw0 + d;
		row = row < h ? row : h - 1; dw = dw < 0 ? 0 : (dw < wdw ? dw : wdw - 1);
		va[k] = CFHD_LDG32((q == 1 ? b1 : b0) + (size_t)row * pitch + 2 * dw);
	}
#pragma unroll
	for (int k = 0; k < N; k++) {
		const int i = (int)threadIdx.x + k * NTHREADS;
		const int q = i / (NROWS * NDW), rem = i - q * (NROWS * NDW), j = rem / NDW, d = rem - j * NDW;
		const int row = row0 + j, dw = dw0 + d;
		if (!(row < h && dw >= 0 && dw < wdw)) va[k] = 0;
	}
}

// Counter-based stand-in for the reference's libc rand() dither: one word of random bits per (frame seed, output row, pixel group).
__device__ __forceinline__ uint32_t dither_word(uint32_t seed, int row, int group)
{
	uint32_t x = seed ^ ((uint32_t)row * 0x9E3779B1u) ^ ((uint32_t)group * 0x85EBCA77u);
	x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
	return x;
}

// Dither bits of the 8-bit 4:2:2 outputs (the reference draws rand() & 1 per sample; which sample gets which bit of the hash is this library's choice, and it is made for the
// strip kernels, where one lane converts the 16 samples of a block of 8 band columns): a group of 32 pixels of an output row = two luma blocks = one V and one U block owns
// two hash words, word 2g for its luma blocks (the odd block takes the word rotated right by 8) and word 2g + 1 for its chroma blocks (V as it is, U rotated by 8).
// Sample s = 4 d + k of a block (d = 0..3, k = 0..3) then reads bit 1 + d + 4 (k & 1) + 16 (k >> 1) of the block's word X, so that (X >> d) & 0x20002 is twice the dither
// of the packed pair (s(4d), s(4d+2)) and (X >> (d + 4)) & 0x20002 that of (s(4d+1), s(4d+3)) -- the form pk_to8_bytes adds (cfhd_gfx950.h).
__device__ __forceinline__ uint32_t dither422_block(uint32_t seed, int orow, int blk, int comp)      // comp: 0 luma block blk, 1 / 2 V / U block blk
{
	const uint32_t w = dither_word(seed, orow, comp == 0 ? (blk & ~1) : 2 * blk + 1);
	return rotr32(w, (comp == 0 ? (blk & 1) != 0 : comp == 2) ? 8u : 0u);
}
__device__ __forceinline__ uint32_t dither422_bit(uint32_t X, int s) { return (X >> (1 + (s >> 2) + 4 * (s & 1) + 16 * ((s >> 1) & 1))) & 1u; }
// The same bits for the kernels that convert one chroma band column cc (two V, two U samples) and its two luma columns (four samples) per item:
// bit 0-3 = luma samples 4 cc .. 4 cc + 3, bit 4 / 6 = V samples 2 cc, 2 cc + 1, bit 5 / 7 = U samples 2 cc, 2 cc + 1.
__device__ __forceinline__ uint32_t dither422_column(uint32_t seed, int orow, int cc)
{
	const uint32_t Xl = dither422_block(seed, orow, cc >> 2, 0), Xv = dither422_block(seed, orow, cc >> 3, 1), Xu = dither422_block(seed, orow, cc >> 3, 2);
	const int sl = 4 * (cc & 3), sc = 2 * (cc & 7);
	return dither422_bit(Xl, sl) | dither422_bit(Xl, sl + 1) << 1 | dither422_bit(Xl, sl + 2) << 2 | dither422_bit(Xl, sl + 3) << 3
	     | dither422_bit(Xv, sc) << 4 | dither422_bit(Xu, sc) << 5 | dither422_bit(Xv, sc + 1) << 6 | dither422_bit(Xu, sc + 1) << 7;
}

// 12-bit component -> 16-bit output word of the 4:4:4(:4) formats; v = lowfilter +/- high before the >>1.  The vector columns of
// InvertHorizontalStrip16sToRow16u clamp to `precision` bits and shift up (InvertHorizontalStrip16s.c:16596, :16724-16750); the
// columns its scalar loop handles (band columns >= w - w%8 - 9, :16876-16990) shift first and saturate to 65535.
__device__ __forceinline__ uint32_t to16(int v, int precision, bool tail)
{
	int x = v >> 1;
	if (x < 0) x = 0;
	if (tail) { x <<= 16 - precision; return (uint32_t)(x > 65535 ? 65535 : x); }
	const int top = (1 << precision) - 1;
	return (uint32_t)(x > top ? top : x) << (16 - precision);
}

// Alpha plane of RGBA 4:4:4:4 -> b64a: the encoder companded alpha into [256, 4095 * 223 / 256 + 256] (frame.c:6696-6707); the decoder expands the
// finished 16-bit word again (codec.h:164-165; the scalar loop of Convert4444LinesToOutput, bayer.c:16212-16226).
__device__ __forceinline__ uint32_t expand_alpha16(uint32_t word)
{
	int a = (int)(word >> 4) - 256;
	a = (a * 8 * 9400) >> 12;
	return (uint32_t)(a < 0 ? 0 : (a > 65535 ? 65535 : a));
}

// PACKED: the last level of the 4:4:4(:4) formats (wavelet.c:4947 TransformInverseRGB444ToRGB48: InvertSpatial*Row16sToYUV16 per
// component + ConvertPlanarRGB16uToPackedRGB48): same synthesis, every sample converted with to16() and stored as one word of
// the interleaved pixel.  gridDim.x = tiles_x * nch as in k_fwd_packed16.
// MODE 1 (own kernel, k_inv_rgb10: the code below it is compiled out of the other instantiations): every plane's sample becomes 10 bits of its
// pixel's 32-bit word -- (value before the final >> 1, + 3) >> 3, clamped -- collected in LDS by plain read-modify-write (the same thread
// handles the same pixels for every plane, a barrier between planes) and stored as whole words.
template <bool PACKED, int MODE = 0>
__device__ __forceinline__ void inv_plane_tile(const InvPlaneJob *jobs, int nch, int wps = 0, uint32_t launch_seed = 0u)
{
	// wps (PACKED): 16-bit words per sample position of plane 0 in the output row -- nch for the interleaved RGB(A) pixels; 2 for YU64
	// (words Y0 C1 Y1 C2: luma every second word, the two half-width chroma planes every fourth, InvPlaneJob::xstride), where a tile of
	// ITW luma band columns covers ITW / 2 chroma band columns.
	const TileId tile = xcd_tile();
	__shared__ InvPlaneJob s_job;
	__shared__ uint32_t s_low[2][ILROWS][IDW];          // LL, LH
	__shared__ uint32_t s_high[2][ITH][IDW];            // HL, HH
	__shared__ uint32_t s_v[2][2][ITH][IDW];            // [row parity][horizontal L/H]
	// PACKED: the workgroup runs the tile of every component in turn and collects the finished words in LDS, pixel by pixel; the rows of the
	// tile then leave as whole pixels in coalesced dwords (one workgroup per component wrote 2 bytes of every 6 or 8: a quarter of each line)
	__shared__ uint16_t s_out[PACKED ? 2 * ITH * 2 * ITW * 4 : 1];
	const int tid = threadIdx.x;
	const uint16_t *frame = nullptr;                     // PACKED: first word of the packed frame
	int w_first = 0;                                     // PACKED: band width of plane 0 (the geometry the tile grid is laid out on)
	if (PACKED) { uintptr_t lo = (uintptr_t)jobs[tile.z * nch].out; for (int c = 1; c < nch; c++) { const uintptr_t p = (uintptr_t)jobs[tile.z * nch + c].out; lo = p < lo ? p : lo; } frame = (const uint16_t *)lo; if (jobs[tile.z * nch].alpha_const) frame -= 1; /* the constant alpha word lies in front of the planes' words */ }
	for (int comp = 0; comp < (PACKED ? nch : 1); comp++) {
	if (comp) __syncthreads();                           // the previous component's tile is finished with s_job and the staging arrays
	stage_job(&s_job, &jobs[PACKED ? tile.z * nch + comp : tile.z]);
	const InvPlaneJob &job = s_job;
	const int w = job.width, h = job.height;
	const int tw = PACKED ? ITW * wps / job.xstride : ITW;     // band columns of this plane under the tile
	const int c0 = tile.x * tw, r0 = tile.y * ITH;
	const bool active = (c0 < w) && (r0 < h);
	const int ll_high = (!PACKED && job.ll_bottom_row_high && h >= 4) ? 1 : 0;      // (InvPlaneJob::ll_bottom_row_high: the last row's LL window starts a row higher)
	const int rs = inv_tile_first_row(r0, h, ll_high);
	const bool bytes8 = PACKED && job.bytes8;
	const int word = PACKED ? (bytes8 ? (int)((const uint8_t *)job.out - (const uint8_t *)frame) : (int)((const uint16_t *)job.out - frame)) : 0;
	if (PACKED && comp == 0) { w_first = w; }
	if (active) {
		enum { NL = (2 * ILROWS * IDW + NTHREADS - 1) / NTHREADS, NH = (2 * ITH * IDW + NTHREADS - 1) / NTHREADS };
		uint32_t vl[NL], vh[NH];
		const int dw0 = (c0 >> 1) - 1, wdw = (w + 1) >> 1;
		inv_stage_load<ILROWS, IDW>(vl, job.band[0], job.band[1], job.band_pitch, rs, h, dw0, wdw);
		inv_stage_load<ITH, IDW>(vh, job.band[2], job.band[3], job.band_pitch, r0, h, dw0, wdw);
#pragma unroll
		for (int k = 0; k < NL; k++) { const int i = tid + k * NTHREADS; if (i < 2 * ILROWS * IDW) (&s_low[0][0][0])[i] = vl[k]; }
#pragma unroll
		for (int k = 0; k < NH; k++) { const int i = tid + k * NTHREADS; if (i < 2 * ITH * IDW) (&s_high[0][0][0])[i] = vh[k]; }
	}
	__syncthreads();
	if (active) {
		for (int i = tid; i < 2 * ITH * IDW; i += NTHREADS) {
			const int q = i / (ITH * IDW), rem = i - q * (ITH * IDW), rl = rem / IDW, d = rem - rl * IDW;
			const int r = r0 + rl;
			if (r >= h) continue;
			const int j = inv_window_first_row(r, h) - rs - ((q == 0 && r == h - 1) ? ll_high : 0), pos = r == 0 ? 0 : (r == h - 1 ? 2 : 1);
			uint32_t e, o;
			inv_vert_pk(s_low[q][j][d], s_low[q][j + 1][d], s_low[q][j + 2][d], s_high[q][rl][d], pos, e, o);
			s_v[0][q][rl][d] = e; s_v[1][q][rl][d] = o;
		}
	}
	__syncthreads();
	if (active) {
		for (int i = tid; i < 2 * ITH * (ITW / 2); i += NTHREADS) {
			const int par = i / (ITH * (ITW / 2)), rem = i - par * (ITH * (ITW / 2)), rl = rem / (ITW / 2), p = rem - rl * (ITW / 2);
			const int r = r0 + rl, c = c0 + 2 * p;
			if (r >= h || c >= w || 2 * p >= tw) continue;
			const uint32_t *L = &s_v[par][0][rl][p + 1];
			const uint32_t dm = L[-1], d0 = L[0], dp = L[1], hh = s_v[par][1][rl][p + 1];
			uint32_t even, odd;
			inv_horiz_pk((dm >> 16) | (d0 << 16), d0, (d0 >> 16) | (dp << 16), hh, even, odd);
			if (PACKED) {
				const int orow = 2 * r + par;
				if (orow >= job.display_height) continue;
				int e[2] = { lo16(even), hi16(even) }, o[2] = { lo16(odd), hi16(odd) };     // columns c, c + 1 before the >>1
				if (c == 0 || c >= w - 2) {
					const int l[6] = { lo16(dm), hi16(dm), lo16(d0), hi16(d0), lo16(dp), hi16(dp) };
#pragma unroll
					for (int k = 0; k < 2; k++) {
						const int col = c + k;
						if (col == 0 || col == w - 1) inv_horiz_border(l, 2 + k, k ? hi16(hh) : lo16(hh), col == 0 ? 0 : 2, e[k], o[k]);
					}
				}
				if (MODE == 1) {
					uint32_t *d32 = (uint32_t *)s_out + (size_t)(2 * rl + par) * (2 * ITW) + 4 * p;
#pragma unroll
					for (int k = 0; k < 2; k++) {
						if (c + k >= w) break;
						int se = e[k] >> 3, so = o[k] >> 3;      // (13 bits -> 10; the format's lowpass bias of 6, decoder.c:12304, is the rounding)
						se = se < 0 ? 0 : (se > 1023 ? 1023 : se); so = so < 0 ? 0 : (so > 1023 ? 1023 : so);
						const uint32_t be = (uint32_t)se << job.bit_shift, bo = (uint32_t)so << job.bit_shift;
						if (comp == 0) { d32[2 * k] = be; d32[2 * k + 1] = bo; } else { d32[2 * k] |= be; d32[2 * k + 1] |= bo; }
					}
					continue;
				}
				const int tail0 = job.alpha_const ? w - 1 : w - (w & 7) - 9;      // (b64a of an RGB 4:4:4 sample: only the last band column behaves like the scalar tail)
				const int xs = job.xstride;
				const size_t at = (size_t)(2 * rl + par) * (2 * ITW) * wps + (size_t)(4 * p) * xs + word;      // sample 2 (c - c0) of tile row 2 rl + par
				uint16_t *dst = s_out + at;
				uint8_t *dst8 = (uint8_t *)s_out + at;
				const uint32_t dz = bytes8 ? dither_word((job.dither_seed ^ launch_seed) + (uint32_t)word * 0x632BE5ABu, orow, c >> 1) : 0u;      // seven bits for each of the thread's four samples
#pragma unroll
				for (int k = 0; k < 2; k++) {
					if (c + k >= w) break;
					const bool tail = c + k >= tail0;
					uint32_t we = to16(e[k], job.precision, tail), wo = to16(o[k], job.precision, tail);
					if (job.alpha && !bytes8) { we = expand_alpha16(we); wo = expand_alpha16(wo); }
					if (job.bytes8 == 2) {
						int be = (int)(we >> 4), bo = (int)(wo >> 4);      // (12-bit values; the lowpass bias of 8, decoder.c:12294, has become their + 2)
						if (job.alpha) {
							be -= 256; bo -= 256;
							be = be < 0 ? 0 : ((be << 3) * 9400) >> 20; bo = bo < 0 ? 0 : ((bo << 3) * 9400) >> 20;
						} else { be >>= 4; bo >>= 4; }
						dst8[(2 * k) * xs] = (uint8_t)(be > 255 ? 255 : be);
						dst8[(2 * k + 1) * xs] = (uint8_t)(bo > 255 ? 255 : bo);
						continue;
					}
					if (bytes8) {
						we = we + 39u + ((dz >> (16 * k)) & 127u); wo = wo + 39u + ((dz >> (16 * k + 8)) & 127u);      // convert.c:6151 with shift 8: (rand() & 127) + 10 * 127 / 32, unsigned saturating add
						we = (we > 65535u ? 65535u : we) >> 8; wo = (wo > 65535u ? 65535u : wo) >> 8;
						dst8[(2 * k) * xs] = (uint8_t)(we > 255u ? 255u : we);
						dst8[(2 * k + 1) * xs] = (uint8_t)(wo > 255u ? 255u : wo);
						continue;
					}
					dst[(2 * k) * xs] = (uint16_t)we;
					dst[(2 * k + 1) * xs] = (uint16_t)wo;
				}
				continue;
			}
			if (job.descale) { even = pk_adds(even, even); odd = pk_adds(odd, odd); }
			else { even = pk_sra(even, 1); odd = pk_sra(odd, 1); }
			uint32_t o0 = pk_lolo(even, odd), o1 = pk_hihi(even, odd);      // (even, odd) of column c and of column c + 1
			if (c == 0 || c >= w - 2) {
				const int l[6] = { lo16(dm), hi16(dm), lo16(d0), hi16(d0), lo16(dp), hi16(dp) };
#pragma unroll
				for (int k = 0; k < 2; k++) {
					const int col = c + k;
					if (col != 0 && col != w - 1) continue;
					int e, o;
					inv_horiz_border(l, 2 + k, k ? hi16(hh) : lo16(hh), col == 0 ? 0 : 2, e, o);
					if (job.descale) { e = sat16(e * 2); o = sat16(o * 2); } else { e = sat16(e >> 1); o = sat16(o >> 1); }
					if (k) o1 = pack16(e, o); else o0 = pack16(e, o);
				}
			}
			int16_t *dst = job.out + (size_t)(2 * r + par) * job.out_pitch + 2 * c;
			if (c + 1 < w) { uint2 v2; v2.x = o0; v2.y = o1; *(uint2 *)dst = v2; }
			else *(uint32_t *)dst = o0;
		}
	}
	}
	if (PACKED) {
		__syncthreads();
		const InvPlaneJob &job = s_job;                   // (height, display height and pitch are the same for every component)
		const int w = w_first, h = job.height, c0 = tile.x * ITW, r0 = tile.y * ITH;
		if (c0 < w && r0 < h) {
			const int npx = 2 * ((w - c0) < ITW ? (w - c0) : ITW);                       // sample positions of plane 0 in the tile's rows inside the frame
			if (MODE == 1) {
				for (int i = tid; i < 2 * ITH * npx; i += NTHREADS) {
					const int orl = i / npx, d = i - orl * npx;
					const int orow = 2 * r0 + orl;
					if (orow >= job.display_height || orow >= 2 * h) continue;
					uint32_t v = ((const uint32_t *)s_out)[(size_t)orl * (2 * ITW) + d];
					if (job.big_endian) v = __builtin_bswap32(v);
					((uint32_t *)frame)[(size_t)orow * job.out_pitch + (size_t)(2 * c0) + d] = v;
				}
			} else if (job.bytes8) {
				// wps = bytes per pixel here; an even band width makes npx a multiple of 4: whole dwords for three-byte pixels too
				const int row_dw = npx * wps / 4;
				for (int i = tid; i < 2 * ITH * row_dw; i += NTHREADS) {
					const int orl = i / row_dw, d = i - orl * row_dw;
					const int orow = 2 * r0 + orl;
					if (orow >= job.display_height || orow >= 2 * h) continue;
					uint32_t v = *(const uint32_t *)((const uint8_t *)s_out + (size_t)orl * (2 * ITW) * wps + 4 * d);
					if (wps == 4 && job.bytes8 == 1) v |= 0xff000000u;                       // alpha (RGB 4:4:4 samples have none)
					const int yrow = job.bottom_up ? job.display_height - 1 - orow : orow;
					*(uint32_t *)((uint8_t *)frame + (size_t)yrow * job.out_pitch + (size_t)(2 * c0) * wps + 4 * d) = v;
				}
			} else {
			const int row_dw = npx * wps / 2;                                            // (an even number: whole dwords)
			for (int i = tid; i < 2 * ITH * row_dw; i += NTHREADS) {
				const int orl = i / row_dw, d = i - orl * row_dw;
				const int orow = 2 * r0 + orl;
				if (orow >= job.display_height || orow >= 2 * h) continue;
				uint32_t v = *(const uint32_t *)(s_out + (size_t)orl * (2 * ITW) * wps + 2 * d);
				if (job.alpha_const && !(d & 1)) v = (v & 0xffff0000u) | (uint32_t)job.alpha_const;      // (four words per pixel: every second dword starts with the alpha word)
				*(uint32_t *)((uint16_t *)frame + (size_t)orow * job.out_pitch + (size_t)(2 * c0) * wps + 2 * d) = v;
			}
			}
		}
	}
}

__global__ void __launch_bounds__(NTHREADS) k_inv_plane(const InvPlaneJob *jobs) { inv_plane_tile<false>(jobs, 1); }
__global__ void __launch_bounds__(NTHREADS) k_inv_packed16(const InvPlaneJob *jobs, int nch, int wps, uint32_t launch_seed) { inv_plane_tile<true>(jobs, nch, wps, launch_seed); }
__global__ void __launch_bounds__(NTHREADS) k_inv_rgb10(const InvPlaneJob *jobs) { inv_plane_tile<true, 1>(jobs, 3, 3, 0u); }

// 10 -> 8 bit reduction of one reconstructed sample v (= lowfilter +/- high, before the >>1):
// negative values clamp to zero first (the +2048 / subs_epu16 pair, InvertHorizontalStrip16s.c:4086-4089),
// dither 0/1 is added before the shift (:3869-3893, rand()&1 per SIMD lane in the reference), result clamps to 8 bits.
__device__ __forceinline__ uint32_t to8(int v, int shift, int dither)
{
	if (v < 0) v = 0;
	int x = ((v >> 1) + dither) >> shift;
	return (uint32_t)(x > 255 ? 255 : x);
}
// pk_to8: the same on two 16-bit lanes (cfhd_gfx950.h)

// One pixel of the RGB32 output from the samples in front of the colour conversion (value before the last >> 1, >> 1, >> shift: ys, and the chroma samples of
// its pixel pair).  vector: the band column lies in the part of the row the reference's SSE2 loop serves (samples clamped to 8 bits, 16-bit products with six
// fraction bits); else its scalar tail (no clamp, seven / eight fraction bits).  Returns B | G << 8 | R << 16 | 255 << 24.
__device__ __forceinline__ uint32_t yuv_to_bgra(int ys, int us, int vs, bool vector, int r_vmult, int g_vmult, int g_umult, int b_umult)
{
	const int ymult = 128 * 149, luma_offset = 16;
	int rr, gg, bb;
	if (vector) {
		int yy = ys - luma_offset, uu = us, vv = vs, t;
		yy = yy < 0 ? 0 : (yy > 255 ? 255 : yy); uu = (uu < 0 ? 0 : (uu > 255 ? 255 : uu)) - 128; vv = (vv < 0 ? 0 : (vv > 255 ? 255 : vv)) - 128;
		yy = (int)(int16_t)((((int)(int16_t)(yy << 7) * ymult) >> 16) << 1);
		t = (int)(int16_t)(vv * r_vmult) >> 1; rr = adds16(adds16(yy, t), 32) >> 6;
		t = (int)(int16_t)(vv * g_vmult) >> 2; gg = subs16(yy, t); t = (int)(int16_t)(uu * g_umult) >> 2; gg = subs16(gg, t); gg = adds16(gg, 32) >> 6;
		t = (int)(int16_t)(uu * b_umult); bb = adds16(adds16(yy, t), 32) >> 6;
	} else {
		const int yy = ((ys - luma_offset) * ymult) >> 7, uu = us - 128, vv = vs - 128;
		rr = (yy + r_vmult * vv + 64) >> 7;
		gg = (yy * 2 - g_umult * uu - g_vmult * vv + 128) >> 8;
		bb = (yy + 2 * b_umult * uu + 64) >> 7;
	}
	rr = rr < 0 ? 0 : (rr > 255 ? 255 : rr); gg = gg < 0 ? 0 : (gg > 255 ? 255 : gg); bb = bb < 0 ? 0 : (bb > 255 ? 255 : bb);
	return (uint32_t)bb | ((uint32_t)gg << 8) | ((uint32_t)rr << 16) | 0xff000000u;
}

// RGB32: the output stage writes four BGRA pixels per item instead of eight bytes of packed 4:2:2 (k_inv_yuv422_rgb32)
template <bool RGB32>
__device__ __forceinline__ void inv_yuv422_tile(const InvYuvJob *jobs, uint32_t launch_seed)
{
	const TileId tile = xcd_tile();
	__shared__ InvYuvJob s_job;
	stage_job(&s_job, &jobs[tile.z]);
	const InvYuvJob &job = s_job;
	const uint32_t seed = job.dither_seed ^ launch_seed;
	const int w = job.width, h = job.height;          // luma band ; chroma bands are w/2 wide
	const int c0 = tile.x * ITW, r0 = tile.y * ITH;
	const int cw = w >> 1, cc0 = c0 >> 1;
	enum { CDW = ITW / 2 + 4, CLD = CDW / 2 };          // chroma columns cc0-2 .. cc0+33, one (V, U) dword each; CLD dwords per band row to load
	__shared__ uint32_t s_ylow[2][ILROWS][IDW], s_yhigh[2][ITH][IDW], s_vy[2][2][ITH][IDW];
	__shared__ uint32_t s_clow[2][ILROWS][CDW], s_chigh[2][ITH][CDW], s_vc[2][2][ITH][CDW];
	const bool active = (c0 < w) && (r0 < h);
	const int tid = threadIdx.x;
	const int rs = inv_tile_first_row(r0, h);
	if (active) {
		enum { NL = (2 * ILROWS * IDW + NTHREADS - 1) / NTHREADS, NH = (2 * ITH * IDW + NTHREADS - 1) / NTHREADS,
		       NCL = (2 * ILROWS * CLD + NTHREADS - 1) / NTHREADS, NCH = (2 * ITH * CLD + NTHREADS - 1) / NTHREADS };
		uint32_t yl[NL], yh[NH], cl[2][NCL], ch[2][NCH];
		const int dw0 = (c0 >> 1) - 1, wdw = (w + 1) >> 1, cdw0 = (cc0 >> 1) - 1, cwdw = (cw + 1) >> 1;
		inv_stage_load<ILROWS, IDW>(yl, job.band[0][0], job.band[0][1], job.band_pitch[0], rs, h, dw0, wdw);
		inv_stage_load<ITH, IDW>(yh, job.band[0][2], job.band[0][3], job.band_pitch[0], r0, h, dw0, wdw);
#pragma unroll
		for (int x = 0; x < 2; x++) {                    // V, U
			inv_stage_load<ILROWS, CLD>(cl[x], job.band[1 + x][0], job.band[1 + x][1], job.band_pitch[1 + x], rs, h, cdw0, cwdw);
			inv_stage_load<ITH, CLD>(ch[x], job.band[1 + x][2], job.band[1 + x][3], job.band_pitch[1 + x], r0, h, cdw0, cwdw);
		}
#pragma unroll
		for (int k = 0; k < NL; k++) { const int i = tid + k * NTHREADS; if (i < 2 * ILROWS * IDW) (&s_ylow[0][0][0])[i] = yl[k]; }
#pragma unroll
		for (int k = 0; k < NH; k++) { const int i = tid + k * NTHREADS; if (i < 2 * ITH * IDW) (&s_yhigh[0][0][0])[i] = yh[k]; }
		// chroma: the two columns of a loaded dword become the V (low) and U (high) halves of two (V, U) dwords
#pragma unroll
		for (int k = 0; k < NCL; k++) {
			const int i = tid + k * NTHREADS;
			if (i < 2 * ILROWS * CLD) {
				uint32_t *dst = &s_clow[0][0][0] + 2 * i;     // item (q, j, m) -> columns 2m, 2m+1 of row (q, j): the row-major index doubles
				dst[0] = pk_lolo(cl[0][k], cl[1][k]); dst[1] = pk_hihi(cl[0][k], cl[1][k]);
			}
		}
#pragma unroll
		for (int k = 0; k < NCH; k++) {
			const int i = tid + k * NTHREADS;
			if (i < 2 * ITH * CLD) {
				uint32_t *dst = &s_chigh[0][0][0] + 2 * i;
				dst[0] = pk_lolo(ch[0][k], ch[1][k]); dst[1] = pk_hihi(ch[0][k], ch[1][k]);
			}
		}
	}
	__syncthreads();
	if (active) {
		for (int i = tid; i < 2 * ITH * (IDW + CDW); i += NTHREADS) {
			const int q = i / (ITH * (IDW + CDW)), rem = i - q * (ITH * (IDW + CDW)), rl = rem / (IDW + CDW), d = rem - rl * (IDW + CDW);
			const int r = r0 + rl;
			if (r >= h) continue;
			const int j = inv_window_first_row(r, h) - rs, pos = r == 0 ? 0 : (r == h - 1 ? 2 : 1);
			uint32_t e, o;
			if (d < IDW) {
				inv_vert_pk(s_ylow[q][j][d], s_ylow[q][j + 1][d], s_ylow[q][j + 2][d], s_yhigh[q][rl][d], pos, e, o);
				s_vy[0][q][rl][d] = e; s_vy[1][q][rl][d] = o;
			} else {
				const int dc = d - IDW;
				inv_vert_pk(s_clow[q][j][dc], s_clow[q][j + 1][dc], s_clow[q][j + 2][dc], s_chigh[q][rl][dc], pos, e, o);
				s_vc[0][q][rl][dc] = e; s_vc[1][q][rl][dc] = o;
			}
		}
	}
	__syncthreads();
	if (active) {
		const int sh = job.shift;
		// one item = one chroma column (V and U side by side) + its two luma columns = 4 luma + 2 x 2 chroma samples = 8 output bytes
		for (int i = tid; i < 2 * ITH * (ITW / 2); i += NTHREADS) {
			const int par = i / (ITH * (ITW / 2)), rem = i - par * (ITH * (ITW / 2)), rl = rem / (ITW / 2), p = rem - rl * (ITW / 2);
			const int r = r0 + rl, cc = cc0 + p, orow = 2 * r + par;
			if (r >= h || cc >= cw || orow >= job.display_height) continue;
			const uint32_t *L = &s_vy[par][0][rl][p + 1];
			const uint32_t dm = L[-1], d0 = L[0], dp = L[1], yhh = s_vy[par][1][rl][p + 1];
			uint32_t ye, yo;                              // (column 2cc, column 2cc+1): even outputs, odd outputs
			inv_horiz_pk((dm >> 16) | (d0 << 16), d0, (d0 >> 16) | (dp << 16), yhh, ye, yo);
			const uint32_t *C = &s_vc[par][0][rl][p + 2];
			const uint32_t cm = C[-1], cz = C[0], cp = C[1], chh = s_vc[par][1][rl][p + 2];
			uint32_t ce, co;                              // (V, U): even output, odd output
			inv_horiz_pk(cm, cz, cp, chh, ce, co);
			if (RGB32) {
				// the samples before the colour conversion, in 32 bits: value before the last >> 1 (border columns: the border taps), >> 1, >> shift, no dither
				int yv[4] = { lo16(ye), lo16(yo), hi16(ye), hi16(yo) };                  // pixels 4 cc .. 4 cc + 3
				int vv[2] = { lo16(ce), lo16(co) }, uv[2] = { hi16(ce), hi16(co) };        // chroma samples 2 cc, 2 cc + 1
				const int c = 2 * cc;
				if (c == 0 || c + 1 == w - 1) {
					const int l[6] = { lo16(dm), hi16(dm), lo16(d0), hi16(d0), lo16(dp), hi16(dp) };
					if (c == 0) inv_horiz_border(l, 2, lo16(yhh), 0, yv[0], yv[1]);
					if (c + 1 == w - 1) inv_horiz_border(l, 3, hi16(yhh), 2, yv[2], yv[3]);
				}
				if (cc == 0 || cc == cw - 1) {
					const int pos = cc == 0 ? 0 : 2;
					const uint32_t far = pos == 0 ? C[2] : C[-2];
					if (pos == 0) { inv_horiz(0, lo16(cz), lo16(cp), lo16(far), lo16(chh), 0, vv[0], vv[1]); inv_horiz(0, hi16(cz), hi16(cp), hi16(far), hi16(chh), 0, uv[0], uv[1]); }
					else { inv_horiz(lo16(cm), lo16(cz), 0, lo16(far), lo16(chh), 2, vv[0], vv[1]); inv_horiz(hi16(cm), hi16(cz), 0, hi16(far), hi16(chh), 2, uv[0], uv[1]); }
				}
				int post_column = w - (w % 16);                     // band columns below it: the reference's vector loop (spatial.c:29640-29644)
				while (post_column > w - 4) post_column -= 16;
				const int r_vmult = job.matrix_601 ? 204 : 230, g_vmult = job.matrix_601 ? 208 : 137, g_umult = job.matrix_601 ? 100 : 55, b_umult = job.matrix_601 ? 129 : 135;
				uint4 px;
				uint32_t *pp = &px.x;
#pragma unroll
				for (int k = 0; k < 4; k++)
					pp[k] = yuv_to_bgra((yv[k] >> 1) >> sh, (uv[k >> 1] >> 1) >> sh, (vv[k >> 1] >> 1) >> sh, c + (k >> 1) < post_column, r_vmult, g_vmult, g_umult, b_umult);
				const int out_row = job.bottom_up ? job.display_height - 1 - orow : orow;
				*(uint4 *)(job.out + (size_t)out_row * job.out_pitch + 16 * (size_t)cc) = px;
				continue;
			}
			const uint32_t dz = sh >= 2 ? dither422_column(seed, orow, cc) : 0u;    // one bit per sample of the item, the assignment of the strip kernels
			// 8-bit samples in 16-bit lanes: te = (y0, y2), to = (y1, y3), tce = (v0, u0), tco = (v1, u1)
			uint32_t te = pk_to8(ye, sh, (dz & 1u) | ((dz << 14) & 0x10000u)), to = pk_to8(yo, sh, ((dz >> 1) & 1u) | ((dz << 13) & 0x10000u));
			uint32_t tce = pk_to8(ce, sh, ((dz >> 4) & 1u) | ((dz << 11) & 0x10000u)), tco = pk_to8(co, sh, ((dz >> 6) & 1u) | ((dz << 9) & 0x10000u));
			const int c = 2 * cc;
			if (c == 0 || c + 1 == w - 1) {               // first / last luma column: 32-bit arithmetic
				const int l[6] = { lo16(dm), hi16(dm), lo16(d0), hi16(d0), lo16(dp), hi16(dp) };
				int e, o;
				if (c == 0) {
					inv_horiz_border(l, 2, lo16(yhh), 0, e, o);
					te = (te & 0xffff0000u) | to8(e, sh, (int)(dz & 1u)); to = (to & 0xffff0000u) | to8(o, sh, (int)((dz >> 1) & 1u));
				}
				if (c + 1 == w - 1) {
					inv_horiz_border(l, 3, hi16(yhh), 2, e, o);
					te = (te & 0xffffu) | (to8(e, sh, (int)((dz >> 2) & 1u)) << 16); to = (to & 0xffffu) | (to8(o, sh, (int)((dz >> 3) & 1u)) << 16);
				}
			}
			if (cc == 0 || cc == cw - 1) {                // first / last chroma column, V and U alike
				// taps: first column -> columns cc, cc+1, cc+2; last column -> cc-2 (far), cc-1, cc
				const int pos = cc == 0 ? 0 : 2;
				const uint32_t far = pos == 0 ? C[2] : C[-2];
				int e, o;
				if (pos == 0) inv_horiz(0, lo16(cz), lo16(cp), lo16(far), lo16(chh), 0, e, o); else inv_horiz(lo16(cm), lo16(cz), 0, lo16(far), lo16(chh), 2, e, o);
				const uint32_t v0 = to8(e, sh, (int)((dz >> 4) & 1u)), v1 = to8(o, sh, (int)((dz >> 6) & 1u));
				if (pos == 0) inv_horiz(0, hi16(cz), hi16(cp), hi16(far), hi16(chh), 0, e, o); else inv_horiz(hi16(cm), hi16(cz), 0, hi16(far), hi16(chh), 2, e, o);
				const uint32_t u0 = to8(e, sh, (int)((dz >> 5) & 1u)), u1 = to8(o, sh, (int)((dz >> 7) & 1u));
				tce = v0 | (u0 << 16); tco = v1 | (u1 << 16);
			}
			const uint32_t y0 = te & 0xffu, y2 = te >> 16, y1 = to & 0xffu, y3 = to >> 16;
			const uint32_t v0 = tce & 0xffu, u0 = tce >> 16, v1 = tco & 0xffu, u1 = tco >> 16;
			uint2 o2;
			if (job.uyvy) { o2.x = u0 | (y0 << 8) | (v0 << 16) | (y1 << 24); o2.y = u1 | (y2 << 8) | (v1 << 16) | (y3 << 24); }
			else { o2.x = y0 | (u0 << 8) | (y1 << 16) | (v0 << 24); o2.y = y2 | (u1 << 8) | (y3 << 16) | (v1 << 24); }
			*(uint2 *)(job.out + (size_t)orow * job.out_pitch + 8 * (size_t)cc) = o2;
		}
	}
}
__global__ void __launch_bounds__(NTHREADS) k_inv_yuv422(const InvYuvJob *jobs, uint32_t launch_seed) { inv_yuv422_tile<false>(jobs, launch_seed); }
__global__ void __launch_bounds__(NTHREADS) k_inv_yuv422_rgb32(const InvYuvJob *jobs) { inv_yuv422_tile<true>(jobs, 0u); }

// =============================================================================================
// k_inv_yuv422_strip: the same last level as k_inv_yuv422, organised around registers instead of LDS tiles.
//
// k_inv_yuv422 is bound by instruction issue, not by bytes (tools/microbench_inv_yuv422.hip: its loads and stores alone run at
// 3.8 TB/s, the same bytes with 16-byte accesses at 5.7 TB/s, the kernel at 2.2 TB/s): per-item index arithmetic, two LDS round
// trips and dword accesses.  Here a workgroup owns a full-width strip of SR band rows and walks down it:
//   waves 0, 1: luma, one lane per block of 8 band columns (lanes 1..62 of a wave are stored, lanes 0 and 63 only supply neighbours, so
//               the two waves overlap by two blocks); wave 2: V; wave 3: U -- the chroma bands are half as wide, so the roles balance;
//   each lane streams its 8 columns of LL, LH, HL, HH with one 16-byte load per band row, keeps the three-row window of the vertical
//   filter in registers, gets the two neighbouring column pairs of the horizontal filter from the adjacent lanes, and produces 16
//   output samples per output row;
//   every lane leaves its 16 8-bit samples per output row in LDS (4 KB per row: Y | V | U), and behind one barrier per band row every
//   thread of the workgroup interleaves one 16-byte word of each of the two output rows (v_perm_b32) and stores it -- nothing but the
//   filter window stays in registers across the barrier, and the four waves share the interleave evenly.  The rows are double-buffered.
// Same arithmetic, same dither bits as k_inv_yuv422 -- the two kernels are interchangeable and tested against each other.
// Geometry served: width % 32 == 0 (chroma band a multiple of 8 columns), any width (segments of 124 luma blocks = 1984 pixels, one
// workgroup each, overlapping by the two neighbour lanes); others take k_inv_yuv422.
// =============================================================================================
enum { SR = 16, SBLK = 8, SLUMA_STEP = 62, SSEG = 2 * SLUMA_STEP, SROW = SSEG * 32 };   // SSEG: luma blocks per segment; SROW: bytes of a segment's output row

struct StripRow { uint32_t d[4]; };                   // 8 band columns = 4 column pairs
__device__ __forceinline__ StripRow strip_load(const int16_t *p) { const cfhd_u4 v = CFHD_LDG128(p); StripRow r; r.d[0] = v.x; r.d[1] = v.y; r.d[2] = v.z; r.d[3] = v.w; return r; }

// Horizontal synthesis + 10 -> 8 bit of one output row of a block: L, H = the vertically synthesised low / high rows of the block,
// prev / next = the neighbouring column pairs.  out[d] = the 8-bit samples 4d .. 4d+3 of the block, bytes in sample order.
// X = the block's dither word (dither422_block; 0 = no dither), sh1 = shift + 1.
__device__ __forceinline__ void strip_row_to8(const uint32_t (&L)[4], const uint32_t (&H)[4], uint32_t prev, uint32_t next, bool first, bool last,
                                              int sh1, uint32_t X, uint32_t (&out)[4])
{
	const uint32_t ext[6] = { prev, L[0], L[1], L[2], L[3], next };
#pragma unroll
	for (int d = 0; d < 4; d++) {
		const uint32_t dm = ext[d], d0 = ext[d + 1], dp = ext[d + 2];
		uint32_t e, o;
		inv_horiz_pk((dm >> 16) | (d0 << 16), d0, (d0 >> 16) | (dp << 16), H[d], e, o);
		out[d] = pk_to8_bytes(e, o, sh1, (X >> d) & 0x20002u, (X >> (d + 4)) & 0x20002u);
	}
	if (first) {                                          // column 0 of the band: 32-bit border taps
		const int l[6] = { 0, 0, lo16(L[0]), hi16(L[0]), lo16(L[1]), hi16(L[1]) };
		int e, o;
		inv_horiz_border(l, 2, lo16(H[0]), 0, e, o);
		out[0] = (out[0] & 0xffff0000u) | to8(e, sh1 - 1, (int)dither422_bit(X, 0)) | (to8(o, sh1 - 1, (int)dither422_bit(X, 1)) << 8);
	}
	if (last) {                                           // last column of the band
		const int l[6] = { lo16(L[2]), hi16(L[2]), lo16(L[3]), hi16(L[3]), 0, 0 };
		int e, o;
		inv_horiz_border(l, 3, hi16(H[3]), 2, e, o);
		out[3] = (out[3] & 0xffffu) | (to8(e, sh1 - 1, (int)dither422_bit(X, 14)) << 16) | (to8(o, sh1 - 1, (int)dither422_bit(X, 15)) << 24);
	}
}

// Block `fb` (flat raster, in blocks of 8 coefficients) of a band that is stored as block lists: zero unless its chunk's mask lists it, else the chunk's
// (rank of the block among the listed ones)-th stored block.  (Measured: fetching the two chunk masks of a wave's row ahead of the arithmetic, through the scalar path,
// cost 14 registers and a wave per SIMD: 1.45 -> 1.63 ms.  The dependent pair of loads stays.)
__device__ __forceinline__ StripRow strip_load_listed(const int16_t *band, const unsigned long long *bmasks, uint32_t fb)
{
	const unsigned long long m = bmasks[fb >> 6];
	const uint32_t bit = fb & 63u;
	StripRow r; r.d[0] = r.d[1] = r.d[2] = r.d[3] = 0u;
	if ((m >> bit) & 1ull) r = strip_load(band + ((size_t)(fb & ~63u) + (size_t)__popcll(m & ((1ull << bit) - 1ull))) * SBLK);
	return r;
}

template <int ROWS_PER_STRIP, bool LATE_LOADS = false, bool BLOCKS = false>
__device__ __forceinline__ void inv_yuv422_strip(const InvYuvJob *jobs, uint32_t launch_seed)
{
	const TileId tile = xcd_tile();
	__shared__ InvYuvJob s_job;
	stage_job(&s_job, &jobs[tile.z]);
	const InvYuvJob &job = s_job;
	__shared__ uint32_t s_rows[2][2][SROW / 4];           // [buffer][output row parity][bytes: Y samples of the row | V samples | U samples]
	const uint32_t seed = job.dither_seed ^ launch_seed;
	const int h = job.height, r0 = tile.y * ROWS_PER_STRIP;
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const bool luma = wave < 2;
	const int comp = luma ? 0 : wave - 1;                 // 0 Y, 1 V, 2 U
	const int w = luma ? job.width : job.width >> 1;      // band columns of this lane's component
	const int nblk = w / SBLK;
	// A workgroup serves one segment of SSEG luma blocks (tile.x) and the SSEG / 2 chroma blocks under it.  Lane l of a wave holds block
	// base - 1 + l: lanes 1..62 are stored, lanes 0 and 63 only supply the neighbouring column pairs (the second luma wave starts
	// 62 blocks further on); blocks outside the band are clamped and store nothing.
	const int seg_first = tile.x * SSEG;                  // first luma block of the segment
	const int base = luma ? seg_first + SLUMA_STEP * wave : seg_first >> 1;
	const int want = base - 1 + lane;
	const int blk = want < 0 ? 0 : (want < nblk ? want : nblk - 1);
	const bool stores = lane >= 1 && lane <= SLUMA_STEP && want < nblk;
	// LDS row layout in dwords, relative to the segment: Y samples (4 dwords per luma block) | V samples | U samples (4 per chroma block)
	const int v_base = SSEG * 4, u_base = v_base + SSEG * 2;
	const int lds_at = luma ? 4 * (want - seg_first) : (comp == 1 ? v_base : u_base) + 4 * (want - (seg_first >> 1));
	const int seg_blocks = job.width / SBLK - seg_first < SSEG ? job.width / SBLK - seg_first : SSEG;
	const int nquads = 2 * seg_blocks;                    // 16-byte words of the segment's part of an output row
	const bool first = blk == 0, last = blk == nblk - 1;
	const int pitch = job.band_pitch[comp];
	// the component is the same for the whole wave: band bases in scalar registers, the lane's block as one 32-bit element offset
	const int16_t *const bLL = wave_uniform_ptr(job.band[comp][0]), *const bLH = wave_uniform_ptr(job.band[comp][1]);
	const int16_t *const bHL = wave_uniform_ptr(job.band[comp][2]), *const bHH = wave_uniform_ptr(job.band[comp][3]);
	const uint32_t boff = (uint32_t)(SBLK * blk);
#define pLL (bLL + boff)
#define pLH (bLH + boff)
#define pHL (bHL + boff)
#define pHH (bHH + boff)
	if (r0 >= h) return;                                  // whole workgroup
	const int nrows = h - r0 < ROWS_PER_STRIP ? h - r0 : ROWS_PER_STRIP;
	int j = inv_window_first_row(r0, h);
	// the highpass bands: dense rows, or (BLOCKS) block lists -- row `row`'s block of this lane is block row * pitch / 8 + blk of the band's flat raster
	const unsigned long long *const mLH = BLOCKS ? wave_uniform_ptr(job.masks + job.mask_base[comp][1]) : nullptr;
	const unsigned long long *const mHL = BLOCKS ? wave_uniform_ptr(job.masks + job.mask_base[comp][2]) : nullptr;
	const unsigned long long *const mHH = BLOCKS ? wave_uniform_ptr(job.masks + job.mask_base[comp][3]) : nullptr;
	const uint32_t pitch8 = (uint32_t)pitch / SBLK;
#define LOAD_LH(row) (BLOCKS ? strip_load_listed(bLH, mLH, (uint32_t)(row) * pitch8 + (uint32_t)blk) : strip_load(pLH + (size_t)(row) * pitch))
#define LOAD_HL(row) (BLOCKS ? strip_load_listed(bHL, mHL, (uint32_t)(row) * pitch8 + (uint32_t)blk) : strip_load(pHL + (size_t)(row) * pitch))
#define LOAD_HH(row) (BLOCKS ? strip_load_listed(bHH, mHH, (uint32_t)(row) * pitch8 + (uint32_t)blk) : strip_load(pHH + (size_t)(row) * pitch))
	StripRow ll0 = strip_load(pLL + (size_t)j * pitch), ll1 = strip_load(pLL + (size_t)(j + 1) * pitch), ll2 = strip_load(pLL + (size_t)(j + 2) * pitch);
	StripRow lh0 = LOAD_LH(j), lh1 = LOAD_LH(j + 1), lh2 = LOAD_LH(j + 2);
	StripRow hl = LOAD_HL(r0), hh = LOAD_HH(r0);
	const int sh = job.shift;
	// what the interleave stage reads of the job, in scalar registers (the job lies in LDS: every use behind a barrier would be another ds_read)
	uint8_t *const out_base = wave_uniform_ptr(job.out);
	const int out_pitch = wave_uniform(job.out_pitch), display_height = wave_uniform(job.display_height), uyvy = wave_uniform(job.uyvy);
	const int dgroup = luma ? (blk & ~1) : 2 * blk + 1;          // dither422_block: the hash word of the lane's block and its rotation
	const uint32_t drot = (luma ? (blk & 1) != 0 : comp == 2) ? 8u : 0u;
	for (int s = 0; s < nrows; s++) {
		const int r = r0 + s;
		// loads of the next band row go out before this row's arithmetic
		const bool more = s + 1 < nrows;
		const int jn = more ? inv_window_first_row(r + 1, h) : j;
		const bool advance = jn != j;
		StripRow nll = ll2, nlh = lh2, nhl = hl, nhh = hh;
		if (!LATE_LOADS) {
			if (advance) { nll = strip_load(pLL + (size_t)(jn + 2) * pitch); nlh = LOAD_LH(jn + 2); }
			if (more) { nhl = LOAD_HL(r + 1); nhh = LOAD_HH(r + 1); }
		}
		// vertical synthesis: rows 2r (even) and 2r + 1 (odd) of the horizontal-low and horizontal-high halves
		const int pos = r == 0 ? 0 : (r == h - 1 ? 2 : 1);
		uint32_t Lv[2][4], Hv[2][4];
		if (pos == 1) {                                       // (one branch for the row, so that the eight packed bodies schedule as one block)
#pragma unroll
			for (int d = 0; d < 4; d++) {
				inv_vert_pk(ll0.d[d], ll1.d[d], ll2.d[d], hl.d[d], 1, Lv[0][d], Lv[1][d]);
				inv_vert_pk(lh0.d[d], lh1.d[d], lh2.d[d], hh.d[d], 1, Hv[0][d], Hv[1][d]);
			}
		} else {
#pragma unroll
			for (int d = 0; d < 4; d++) {
				inv_vert_pk(ll0.d[d], ll1.d[d], ll2.d[d], hl.d[d], pos, Lv[0][d], Lv[1][d]);
				inv_vert_pk(lh0.d[d], lh1.d[d], lh2.d[d], hh.d[d], pos, Hv[0][d], Hv[1][d]);
			}
		}
		// every lane leaves the 16 8-bit samples of its block in LDS (bytes in sample order), luma and chroma alike ...
		uint32_t (*rowbuf)[SROW / 4] = s_rows[s & 1];
#pragma unroll
		for (int par = 0; par < 2; par++) {
			const int orow = 2 * r + par;
			const uint32_t prev = __shfl(Lv[par][3], lane - 1), next = __shfl(Lv[par][0], lane + 1);
			// dither bits: the block's word of the output row (dither422_block)
			const uint32_t X = sh >= 2 ? rotr32(dither_word(seed, orow, dgroup), drot) : 0u;
			uint32_t out[4];
			strip_row_to8(Lv[par], Hv[par], prev, next, first, last, sh + 1, X, out);
			if (stores) {
				// row layout: Y samples | V samples | U samples
				uint32_t *dst = &rowbuf[par][lds_at];
#pragma unroll
				for (int d = 0; d < 4; d++) dst[d] = out[d];
			}
		}
		__syncthreads();
		// ... and every thread of the workgroup interleaves one 16-byte word (4 pixel pairs) of each of the two output rows
		if (tid < nquads) {
#pragma unroll
			for (int par = 0; par < 2; par++) {
				const int orow = 2 * r + par;
				if (orow >= display_height) continue;
				const uint32_t *row = rowbuf[par];
				const uint32_t y0 = row[2 * tid], y1 = row[2 * tid + 1], vv = row[v_base + tid], uu = row[u_base + tid];
				// (u0, v0, u1, v1) and (u2, v2, u3, v3), then y(2m), u(m), y(2m+1), v(m) -- or u, y, v, y for UYVY
				const uint32_t uv01 = byte_perm(vv, uu, 0x05010400u), uv23 = byte_perm(vv, uu, 0x07030602u);
				uint4 q;
				if (uyvy) {
					q.x = byte_perm(uv01, y0, 0x01050004u); q.y = byte_perm(uv01, y0, 0x03070206u);
					q.z = byte_perm(uv23, y1, 0x01050004u); q.w = byte_perm(uv23, y1, 0x03070206u);
				} else {
					q.x = byte_perm(uv01, y0, 0x05010400u); q.y = byte_perm(uv01, y0, 0x07030602u);
					q.z = byte_perm(uv23, y1, 0x05010400u); q.w = byte_perm(uv23, y1, 0x07030602u);
				}
				*(uint4 *)(out_base + (size_t)orow * out_pitch + 32 * (size_t)seg_first + 16 * (size_t)tid) = q;
			}
		}
		if (LATE_LOADS) {
			if (advance) { nll = strip_load(pLL + (size_t)(jn + 2) * pitch); nlh = LOAD_LH(jn + 2); }
			if (more) { nhl = LOAD_HL(r + 1); nhh = LOAD_HH(r + 1); }
		}
		if (advance) { ll0 = ll1; ll1 = ll2; ll2 = nll; lh0 = lh1; lh1 = lh2; lh2 = nlh; j = jn; }
		hl = nhl; hh = nhh;
	}
#undef pLL
#undef pLH
#undef pHL
#undef pHH
#undef LOAD_LH
#undef LOAD_HL
#undef LOAD_HH
}
// (Round 6, measured and not kept: the band rows three to a round with the window's registers renamed instead of moved, as in k_fwd_yuv422_strip -- 85 registers = five waves per SIMD:
// 1.07 -> 1.18 ms; held to 80 registers with amdgpu_waves_per_eu(6, 6): 1.07 ms, no gain for a loop body written as a macro.  tools/gpu_r06_t.sh.)
__global__ void __launch_bounds__(NTHREADS) k_inv_yuv422_strip(const InvYuvJob *jobs, uint32_t launch_seed) { inv_yuv422_strip<SR, true>(jobs, launch_seed); }
__global__ void __launch_bounds__(NTHREADS) k_inv_yuv422_strip_blocks(const InvYuvJob *jobs, uint32_t launch_seed) { inv_yuv422_strip<SR, true, true>(jobs, launch_seed); }

// =============================================================================================
// k_fwd_yuv422_strip: level 1 of the packed 4:2:2 formats with the organisation of k_inv_yuv422_strip (registers and lane exchange instead
// of LDS tiles, 16-byte accesses).  A workgroup walks down a full-width strip of SRF band rows:
//   waves 0, 1: one lane per 16 pixels (32 bytes of every picture row): 8 luma pairs -> 8 band columns; the lane also cuts the 4 V and
//               4 U sample pairs out of its pixels and leaves them in LDS;  wave 2: V, wave 3: U, one lane per 8 chroma band columns
//               (16 samples = 8 pairs read back from LDS).  From the pairs on, luma and chroma lanes run the same code:
//   horizontal 2/6 analysis of the row's 8 pairs (neighbour pairs from the adjacent lanes), results pushed into a six-row register
//   window; every second row the vertical 2/6 analysis + quantizer emits one row of LL, LH, HL, HH, 16 bytes per band.
// One barrier per picture row pair (the LDS chroma rows are double-buffered).  Same arithmetic as k_fwd_yuv422, tested against it.
// Geometry served: width % 32 == 0 (segments of 1984 pixels), 16-byte aligned rows; everything else takes k_fwd_yuv422.
// =============================================================================================
enum { SRF = 32, SFPLANE = 512 };                       // band rows per strip; chroma pairs per picture row (dwords) the LDS buffer holds per channel

// Horizontal analysis of one picture row of a block: p = its 8 sample pairs, prev / next = the neighbouring pairs.
__device__ __forceinline__ void strip_fwd_row(const uint32_t (&p)[8], uint32_t prev, uint32_t next, bool first, bool last, uint32_t (&L)[4], uint32_t (&H)[4], int prescale = 0)
{
	const uint32_t ext[10] = { prev, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], next };
#pragma unroll
	for (int m = 0; m < 4; m++) horiz_pair(&ext[2 * m], 0u, prescale, first && m == 0, false, last && m == 3, L[m], H[m]);
}

enum { FWD_LATE_LOADS = 2 };   // 0: a row pair ahead of the arithmetic, 1: behind the horizontal pass, 2: at the start of the pair
// Per-lane state of k_fwd_yuv422_strip.
struct FwdStrip {
	uint32_t LW[6][4], HW[6][4];                          // window of horizontally analysed rows: picture rows wtop .. wtop + 5
	cfhd_u4 raw[2][2];                                    // luma lanes: the 32 bytes of the next two picture rows, in flight
};

// Fetch of the lane's 32 bytes of picture rows y, y + 1 (luma lanes); rows below the picture read as 0x80 (encoder.c:2442-2478).
__device__ __forceinline__ void strip_fwd_fetch(FwdStrip &st, const FwdYuvJob &job, const uint8_t *in, int y)
{
#pragma unroll
	for (int k = 0; k < 2; k++) {
		if (y + k < job.display_height) { const uint8_t *p = in + (size_t)(y + k) * job.in_pitch; st.raw[k][0] = CFHD_LDG128(p); st.raw[k][1] = CFHD_LDG128(p + 16); }
		else { cfhd_u4 g; g.x = g.y = g.z = g.w = 0x80808080u; st.raw[k][0] = g; st.raw[k][1] = g; }
	}
}

// Pushes picture rows y, y + 1 into window slots SLOT, SLOT + 1: the luma lanes take them from st.raw (and start the fetch of rows
// y + 2, y + 3 when `prefetch`), leave the chroma sample pairs in LDS; the chroma lanes pick theirs up behind the barrier.
template <int SLOT>
__device__ __forceinline__ void strip_fwd_push(FwdStrip &st, const FwdYuvJob &job, const uint8_t *in, uint32_t (*buf)[2][SFPLANE], int y, bool prefetch,
                                               bool luma, int comp, int lds_write_at, int lds_read_at, int lane, bool first, bool last, int shift, int ysh0, uint32_t usel, uint32_t vsel)
{
	uint32_t p[2][8];
	if (luma) {
		if (FWD_LATE_LOADS == 2) strip_fwd_fetch(st, job, in, y);      // no load in flight across the vertical pass at all
		uint32_t a[2][8];
#pragma unroll
		for (int k = 0; k < 2; k++) {
			a[k][0] = st.raw[k][0].x; a[k][1] = st.raw[k][0].y; a[k][2] = st.raw[k][0].z; a[k][3] = st.raw[k][0].w;
			a[k][4] = st.raw[k][1].x; a[k][5] = st.raw[k][1].y; a[k][6] = st.raw[k][1].z; a[k][7] = st.raw[k][1].w;
		}
		if (prefetch && !FWD_LATE_LOADS) strip_fwd_fetch(st, job, in, y + 2);       // the next pair's loads go out before this pair's arithmetic
#pragma unroll
		for (int k = 0; k < 2; k++) {
#pragma unroll
			for (int i = 0; i < 8; i++) p[k][i] = ((a[k][i] >> ysh0) & 0x00ff00ffu) << shift;
			if (lds_write_at >= 0) {
#pragma unroll
				for (int i = 0; i < 4; i++) {
					buf[k][0][lds_write_at + i] = byte_perm(a[k][2 * i + 1], a[k][2 * i], vsel) << shift;
					buf[k][1][lds_write_at + i] = byte_perm(a[k][2 * i + 1], a[k][2 * i], usel) << shift;
				}
			}
		}
	}
	__syncthreads();
	if (!luma) {
#pragma unroll
		for (int k = 0; k < 2; k++) {
#pragma unroll
			for (int i = 0; i < 8; i++) p[k][i] = buf[k][comp - 1][lds_read_at + i < 0 ? 0 : lds_read_at + i];
		}
	}
#pragma unroll
	for (int k = 0; k < 2; k++) {
		const uint32_t prev = __shfl(p[k][7], lane - 1), next = __shfl(p[k][0], lane + 1);
		strip_fwd_row(p[k], prev, next, first, last, st.LW[SLOT + k], st.HW[SLOT + k]);
	}
	// (FWD_LATE_LOADS) the next pair's loads go out behind this pair's arithmetic: nothing but the window is live across it, which is
	// worth one more wave per SIMD -- and the kernel's throughput follows the number of resident waves, not the distance of its loads
	if (luma && prefetch && FWD_LATE_LOADS == 1) strip_fwd_fetch(st, job, in, y + 2);
}

// BLOCKS: the quantized bands leave as block lists (FwdBlockLists); DENSE: and as dense rows of the pyramid as well (whatever reads the pyramid afterwards --
// the host writer, a download of the coefficients -- needs them; the GPU entropy stage alone does not).
template <int ROWS_PER_STRIP, bool BLOCKS = false, bool DENSE = true>
__device__ __forceinline__ void fwd_yuv422_strip(const FwdYuvJob *jobs)
{
	const TileId tile = xcd_tile();
	__shared__ FwdYuvJob s_job;
	stage_job(&s_job, &jobs[tile.z]);
	const FwdYuvJob &job = s_job;
	__shared__ uint32_t s_pairs[2][2][2][SFPLANE];        // [buffer][row of the pair][V, U][chroma sample pairs of the row]
	const int W = job.width, H = job.height, HH = H >> 1;
	const int r0 = tile.y * ROWS_PER_STRIP;
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const bool luma = wave < 2;
	const int comp = luma ? 0 : wave - 1;                 // 0 Y, 1 V, 2 U
	const QuantParam q_lh = wave_uniform_quant(job.q[comp][1]), q_hl = wave_uniform_quant(job.q[comp][2]), q_hh = wave_uniform_quant(job.q[comp][3]);      // (a wave = one component)
	const int nblk = luma ? W / 16 : W / 32;              // blocks of 8 band columns
	// segments and lanes as in k_inv_yuv422_strip: lane l of a wave holds block base - 1 + l, lanes 1..62 store
	const int seg_first = tile.x * SSEG;
	const int base = luma ? seg_first + SLUMA_STEP * wave : seg_first >> 1;
	const int want = base - 1 + lane;
	const int blk = want < 0 ? 0 : (want < nblk ? want : nblk - 1);
	const bool stores = lane >= 1 && lane <= SLUMA_STEP && want < nblk;
	const bool first = blk == 0, last = blk == nblk - 1;
	// chroma sample pairs in LDS, relative to luma block seg_first - 1 (4 pairs per luma block): every luma block of the segment and
	// its two neighbours is written once (wave 0: lanes 0..62, wave 1: lanes 1..63); chroma lane l reads the 8 pairs from 8 l - 4 on
	// (the outer halves of the two halo lanes are never used)
	const int lds_write_at = (luma && want >= 0 && want < nblk && (wave == 0 ? lane <= SLUMA_STEP : lane >= 1)) ? 4 * (want - seg_first + 1) : -1;
	const int lds_read_at = 8 * lane - 4;
	if (r0 >= HH) return;
	const int r1 = r0 + ROWS_PER_STRIP < HH ? r0 + ROWS_PER_STRIP : HH;
	const int shift = job.shift;
	const int ysh0 = job.uyvy ? 8 : 0, ub = job.uyvy ? 0 : 1, vb = job.uyvy ? 2 : 3;       // byte lanes of Y0, U, V in a pixel-pair dword
	const uint32_t usel = (uint32_t)ub | 0x0c00u | ((uint32_t)(4 + ub) << 16) | 0x0c000000u;   // (byte ub of a0, 0, byte ub of a1, 0)
	const uint32_t vsel = (uint32_t)vb | 0x0c00u | ((uint32_t)(4 + vb) << 16) | 0x0c000000u;
	const uint8_t *in = job.in + 32 * (size_t)blk;
	// block lists: this wave's chunk of every band row = the blocks its lanes 1 .. SLUMA_STEP store (chunk `base / FWD_CHUNK_BLOCKS` of the row)
	static_assert(SLUMA_STEP == FWD_CHUNK_BLOCKS, "a chunk is what one wave stores of a band row");
	const int chunks_per_row = BLOCKS ? wave_uniform((nblk + FWD_CHUNK_BLOCKS - 1) / FWD_CHUNK_BLOCKS) : 0, chunk = BLOCKS ? wave_uniform(base / FWD_CHUNK_BLOCKS) : 0;
	// per band, wave-uniform (scalar registers; hoisted: the barrier inside the row loop keeps the compiler from keeping LDS job fields across rows): the slot of the
	// chunk's first block in band row 0 and the chunk's mask in band row 0
	uint4 *slot0[3]; unsigned long long *mask0[3];
	const int opitch = wave_uniform(job.out_pitch[comp]), pitch8 = opitch >> 3;
	int16_t *outp[4];                                     // the dense rows: scalar bases, the lane's block as a 32-bit offset
#pragma unroll
	for (int b = 0; b < 4; b++) outp[b] = wave_uniform_ptr(job.out[comp][b]);
	if (BLOCKS) {
#pragma unroll
		for (int b = 1; b < 4; b++) {
			slot0[b - 1] = wave_uniform_ptr(job.lists.blocks + (size_t)(job.out[comp][b] - job.lists.base) / 8 + (size_t)(FWD_CHUNK_BLOCKS * chunk));
			mask0[b - 1] = wave_uniform_ptr(job.lists.masks + job.lists.mask_base[comp][b] + chunk);
		}
	}
	FwdStrip st;
	int wtop = window_first_row(r0, HH, H);
	const int lastrow = window_first_row(r1 - 1, HH, H) + 5;     // last picture row this strip reads
	int t = 0;                                            // row pairs pushed so far (selects the LDS buffer)
#define CFHD_PUSH(SLOT, Y) strip_fwd_push<SLOT>(st, job, in, s_pairs[t & 1], (Y), (Y) + 2 <= lastrow, luma, comp, lds_write_at, lds_read_at, lane, first, last, shift, ysh0, usel, vsel); t++
	if (luma && FWD_LATE_LOADS != 2) strip_fwd_fetch(st, job, in, wtop);
	CFHD_PUSH(0, wtop); CFHD_PUSH(2, wtop + 2); CFHD_PUSH(4, wtop + 4);
	// One band row: vertical analysis + quantizer of row r from the window, then the stores.
	auto band_row = [&](const int r, const int pos) {
		// vertical analysis + quantizer of band row r
		uint32_t o[4][4];
#pragma unroll
		for (int d = 0; d < 4; d++) {
			uint32_t ll, lh, hl, hh;
			if (pos == 1) {
				ll = pk_adds(st.LW[2][d], st.LW[3][d]); hl = pk_hp_mid(st.LW[0][d], st.LW[1][d], st.LW[2][d], st.LW[3][d], st.LW[4][d], st.LW[5][d]);
				lh = pk_adds(st.HW[2][d], st.HW[3][d]); hh = pk_hp_mid(st.HW[0][d], st.HW[1][d], st.HW[2][d], st.HW[3][d], st.HW[4][d], st.HW[5][d]);
			} else {
				int res[4][2];
#pragma unroll
				for (int e = 0; e < 2; e++) {
					int a[6], b[6];
#pragma unroll
					for (int k = 0; k < 6; k++) { a[k] = e ? hi16(st.LW[k][d]) : lo16(st.LW[k][d]); b[k] = e ? hi16(st.HW[k][d]) : lo16(st.HW[k][d]); }
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
			o[0][d] = ll; o[1][d] = pk_quantize(lh, q_lh);                              // the lowpass band is never quantized (quantize.c:3216)
			o[2][d] = pk_quantize(hl, q_hl); o[3][d] = pk_quantize(hh, q_hh);
		}
		if (BLOCKS) {
			// the quantized bands as block lists: a lane's block goes out only when it holds a nonzero coefficient, to the slot its rank among the
			// chunk's nonzero blocks gives it (lanes in block order: a ballot + v_mbcnt), and lane 0 leaves the chunk's occupancy mask
#pragma unroll
			for (int b = 1; b < 4; b++) {
				const bool nz = stores && (o[b][0] | o[b][1] | o[b][2] | o[b][3]) != 0u;
				const unsigned long long m = __ballot(nz);
				uint4 *rowslots = wave_uniform_ptr(slot0[b - 1] + (size_t)(r * pitch8));      // (scalar base + one 32-bit lane offset)
				if (nz) store_u32x4_global(rowslots + wave_mbcnt(m), o[b][0], o[b][1], o[b][2], o[b][3]);      // (lane 0 never stores: the rank among lanes 1 ..)
				if (lane == 0 && chunk < chunks_per_row) mask0[b - 1][r * chunks_per_row] = m >> 1;       // bit i: block i of the chunk (lane i + 1); a wave beyond the last block of the row has no chunk
			}
		}
		if (stores) {
#pragma unroll
			for (int b = 0; b < 4; b++) {
				if (b && BLOCKS && !DENSE) continue;
				store_u32x4_global(outp[b] + (uint32_t)(r * opitch + SBLK * blk), o[b][0], o[b][1], o[b][2], o[b][3]);
			}
		}
	};
	auto window_down = [&]() {                               // the window moves down by two picture rows
#pragma unroll
		for (int k = 0; k < 4; k++) {
#pragma unroll
			for (int d = 0; d < 4; d++) { st.LW[k][d] = st.LW[k + 2][d]; st.HW[k][d] = st.HW[k + 2][d]; }
		}
	};
	auto any_row = [&](const int r) {
		const int need = window_first_row(r, HH, H);
		if (need != wtop) { window_down(); wtop = need; CFHD_PUSH(4, wtop + 4); }
		band_row(r, r == 0 ? 0 : (r == HH - 1 ? 2 : 1));
	};
	// Rows 2 .. HH - 2 behind the strip's first: the window moves with every row and the taps are the interior ones.  Unrolled by three, the six window rows come
	// back to the registers they started in, so the moves of window_down() are names, not instructions (they were 64 v_mov_b32 per band row, an eighth of the kernel).
	int r = r0;
	for (; r < r1 && (r == r0 || r < 2); r++) any_row(r);
	const int mid_end = r1 < HH - 1 ? r1 : HH - 1;
	// (written out: the compiler does not unroll a loop around a barrier by itself when the trip count is not a known multiple)
#define CFHD_MID_ROW(R) window_down(); wtop += 2; CFHD_PUSH(4, wtop + 4); band_row((R), 1)
	for (; r + 3 <= mid_end; r += 3) { CFHD_MID_ROW(r); CFHD_MID_ROW(r + 1); CFHD_MID_ROW(r + 2); }
	for (; r < mid_end; r++) { CFHD_MID_ROW(r); }
#undef CFHD_MID_ROW
	for (; r < r1; r++) any_row(r);
#undef CFHD_PUSH
}
__global__ void __launch_bounds__(NTHREADS) k_fwd_yuv422_strip(const FwdYuvJob *jobs) { fwd_yuv422_strip<SRF>(jobs); }
__global__ void __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) k_fwd_yuv422_strip_blocks(const FwdYuvJob *jobs) { fwd_yuv422_strip<SRF, true, false>(jobs); }
__global__ void __launch_bounds__(NTHREADS) k_fwd_yuv422_strip_blocks_dense(const FwdYuvJob *jobs) { fwd_yuv422_strip<SRF, true, true>(jobs); }

// =============================================================================================
// k_inv_plane_strip / k_fwd_plane_strip: levels 2 and 3 (int16 planes on both sides) in the register-strip organisation.  These planes
// are narrow (1080p: 60 / 30 / 15 blocks of 8 band columns), so a wave carries 64 >> glog planes side by side, 1 << glog lanes each
// (the same strip of consecutive jobs: geometry and row position are wave-uniform), and needs neither LDS nor barriers: the neighbour
// columns come from the adjacent lanes, and the lanes at the edges of a group sit on the band borders, whose taps do not look outside.
// One launch per group of equally wide channels (luma | the two chroma planes of 4:2:2 | all planes of 4:4:4).
// Wider planes (more than 64 blocks: level 2 of 4K / 8K frames) take one plane per wave in segments of 62 blocks (strip_who, nseg > 1).
// Geometry served: band width a multiple of 8; everything else takes k_inv_plane / k_fwd_plane.
// =============================================================================================
enum { SRP = 16 };

struct StripWho { int job; int blk; bool stores; int strip; };
// wave -> (strip of rows, group of 64 >> glog consecutive selected jobs); lane -> (job of the group, block of 8 columns)
// nseg > 1 (planes of more than 64 blocks: level 2 of 4K and 8K frames): one plane per wave, cut into segments of PLSTEP blocks; lanes 0 and 63 of a wave carry the
// blocks next to its segment and only feed their neighbours (as in the level-1 strip kernels).
enum { PLSTEP = 62 };
__device__ __forceinline__ StripWho strip_who(int nframes, int nch, int c0, int nc, int glog, int nstrips, int nblk, bool *wave_idle, int nseg = 1)
{
	const int lane = threadIdx.x & 63, gwave = (int)blockIdx.x * (NTHREADS / 64) + (int)(threadIdx.x >> 6);
	const bool segments = nseg > 1;
	const int per_wave = segments ? 1 : 64 >> glog, sub = segments ? 0 : lane >> glog;
	const int strip = gwave % nstrips, rest = gwave / nstrips, seg = segments ? rest % nseg : 0;
	const int b = segments ? seg * PLSTEP - 1 + lane : (lane & ((1 << glog) - 1));
	const int first_sel = (segments ? rest / nseg : rest) * per_wave, nsel = nframes * nc;
	*wave_idle = first_sel >= nsel;
	int sel = first_sel + sub;
	const bool job_ok = sel < nsel;
	if (!job_ok) sel = nsel - 1;
	StripWho w;
	w.job = (sel / nc) * nch + c0 + sel % nc;
	w.stores = job_ok && b >= 0 && b < nblk && (!segments || (lane >= 1 && lane <= PLSTEP));
	w.blk = b < 0 ? 0 : (b < nblk ? b : nblk - 1);
	w.strip = strip;
	return w;
}

__global__ void __launch_bounds__(NTHREADS) k_inv_plane_strip(const InvPlaneJob *jobs, int nframes, int nch, int c0, int nc, int glog, int nstrips, int w, int h, int nseg)
{
	const int lane = threadIdx.x & 63;
	const int nblk = w / SBLK;
	bool idle;
	const StripWho who = strip_who(nframes, nch, c0, nc, glog, nstrips, nblk, &idle, nseg);
	if (idle) return;                                     // whole wave
	const InvPlaneJob *job = &jobs[who.job];
	const int blk = who.blk, pitch = job->band_pitch, descale = job->descale;
	const bool first = blk == 0, last = blk == nblk - 1;
	const int16_t *pLL = job->band[0] + SBLK * blk, *pLH = job->band[1] + SBLK * blk, *pHL = job->band[2] + SBLK * blk, *pHH = job->band[3] + SBLK * blk;
	int16_t *out = job->out + 2 * SBLK * blk;
	const int out_pitch = job->out_pitch;
	const int r0 = who.strip * SRP;
	const int nrows = h - r0 < SRP ? h - r0 : SRP;
	int j = inv_window_first_row(r0, h);
	StripRow ll0 = strip_load(pLL + (size_t)j * pitch), ll1 = strip_load(pLL + (size_t)(j + 1) * pitch), ll2 = strip_load(pLL + (size_t)(j + 2) * pitch);
	StripRow lh0 = strip_load(pLH + (size_t)j * pitch), lh1 = strip_load(pLH + (size_t)(j + 1) * pitch), lh2 = strip_load(pLH + (size_t)(j + 2) * pitch);
	StripRow hl = strip_load(pHL + (size_t)r0 * pitch), hh = strip_load(pHH + (size_t)r0 * pitch);
	for (int s = 0; s < nrows; s++) {
		const int r = r0 + s;
		const bool more = s + 1 < nrows;
		const int jn = more ? inv_window_first_row(r + 1, h) : j;
		const bool advance = jn != j;
		StripRow nll = ll2, nlh = lh2, nhl = hl, nhh = hh;
		const int pos = r == 0 ? 0 : (r == h - 1 ? 2 : 1);
		uint32_t Lv[2][4], Hv[2][4];
		if (pos == 1) {                                       // (one branch for the row, as in k_inv_yuv422_strip)
#pragma unroll
			for (int d = 0; d < 4; d++) {
				inv_vert_pk(ll0.d[d], ll1.d[d], ll2.d[d], hl.d[d], 1, Lv[0][d], Lv[1][d]);
				inv_vert_pk(lh0.d[d], lh1.d[d], lh2.d[d], hh.d[d], 1, Hv[0][d], Hv[1][d]);
			}
		} else {
#pragma unroll
			for (int d = 0; d < 4; d++) {
				inv_vert_pk(ll0.d[d], ll1.d[d], ll2.d[d], hl.d[d], pos, Lv[0][d], Lv[1][d]);
				inv_vert_pk(lh0.d[d], lh1.d[d], lh2.d[d], hh.d[d], pos, Hv[0][d], Hv[1][d]);
			}
		}
#pragma unroll
		for (int par = 0; par < 2; par++) {
			const uint32_t (&L)[4] = Lv[par];
			const uint32_t (&H)[4] = Hv[par];
			const uint32_t prev = __shfl(L[3], lane - 1), next = __shfl(L[0], lane + 1);
			const uint32_t ext[6] = { prev, L[0], L[1], L[2], L[3], next };
			uint32_t o[8];
#pragma unroll
			for (int d = 0; d < 4; d++) {
				const uint32_t dm = ext[d], d0 = ext[d + 1], dp = ext[d + 2];
				uint32_t even, odd;
				inv_horiz_pk((dm >> 16) | (d0 << 16), d0, (d0 >> 16) | (dp << 16), H[d], even, odd);
				if (descale) { even = pk_adds(even, even); odd = pk_adds(odd, odd); }
				else { even = pk_sra(even, 1); odd = pk_sra(odd, 1); }
				o[2 * d] = pk_lolo(even, odd); o[2 * d + 1] = pk_hihi(even, odd);      // (even, odd) of column 2d and of column 2d + 1
			}
			if (first) {
				const int l[6] = { 0, 0, lo16(L[0]), hi16(L[0]), lo16(L[1]), hi16(L[1]) };
				int e, od;
				inv_horiz_border(l, 2, lo16(H[0]), 0, e, od);
				if (descale) { e = sat16(e * 2); od = sat16(od * 2); } else { e = sat16(e >> 1); od = sat16(od >> 1); }
				o[0] = pack16(e, od);
			}
			if (last) {
				const int l[6] = { lo16(L[2]), hi16(L[2]), lo16(L[3]), hi16(L[3]), 0, 0 };
				int e, od;
				inv_horiz_border(l, 3, hi16(H[3]), 2, e, od);
				if (descale) { e = sat16(e * 2); od = sat16(od * 2); } else { e = sat16(e >> 1); od = sat16(od >> 1); }
				o[7] = pack16(e, od);
			}
			if (who.stores) {
				uint4 *dst = (uint4 *)(out + (size_t)(2 * r + par) * out_pitch);
				uint4 q0, q1;
				q0.x = o[0]; q0.y = o[1]; q0.z = o[2]; q0.w = o[3]; q1.x = o[4]; q1.y = o[5]; q1.z = o[6]; q1.w = o[7];
				dst[0] = q0; dst[1] = q1;
			}
		}
		// the next row's loads go out behind this row's arithmetic (fewer live registers, one more wave per SIMD; cf. k_inv_yuv422_strip)
		if (advance) { nll = strip_load(pLL + (size_t)(jn + 2) * pitch); nlh = strip_load(pLH + (size_t)(jn + 2) * pitch); }
		if (more) { nhl = strip_load(pHL + (size_t)(r + 1) * pitch); nhh = strip_load(pHH + (size_t)(r + 1) * pitch); }
		if (advance) { ll0 = ll1; ll1 = ll2; ll2 = nll; lh0 = lh1; lh1 = lh2; lh2 = nlh; j = jn; }
		hl = nhl; hh = nhh;
	}
}

// Vertical 2/6 analysis + quantizer of one band row from the six-row window (shared by the strip kernels).
template <int ND>
__device__ __forceinline__ void strip_fwd_emit(const uint32_t (&LW)[6][ND], const uint32_t (&HW)[6][ND], int pos, const QuantParam &q_lh, const QuantParam &q_hl, const QuantParam &q_hh,
                                               uint32_t (&o)[4][ND])
{
#pragma unroll
	for (int d = 0; d < ND; d++) {
		uint32_t ll, lh, hl, hh;
		if (pos == 1) {
			ll = pk_adds(LW[2][d], LW[3][d]); hl = pk_hp_mid(LW[0][d], LW[1][d], LW[2][d], LW[3][d], LW[4][d], LW[5][d]);
			lh = pk_adds(HW[2][d], HW[3][d]); hh = pk_hp_mid(HW[0][d], HW[1][d], HW[2][d], HW[3][d], HW[4][d], HW[5][d]);
		} else {
			int res[4][2];
#pragma unroll
			for (int e = 0; e < 2; e++) {
				int a[6], b[6];
#pragma unroll
				for (int k = 0; k < 6; k++) { a[k] = e ? hi16(LW[k][d]) : lo16(LW[k][d]); b[k] = e ? hi16(HW[k][d]) : lo16(HW[k][d]); }
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
		o[0][d] = ll; o[1][d] = pk_quantize(lh, q_lh);                                    // the lowpass band is never quantized (quantize.c:3216)
		o[2][d] = pk_quantize(hl, q_hl); o[3][d] = pk_quantize(hh, q_hh);
	}
}

// Two rows of an int16 plane -> window slots SLOT, SLOT + 1 (k_fwd_plane_strip).
template <int SLOT>
__device__ __forceinline__ void strip_plane_push(uint32_t (&LW)[6][4], uint32_t (&HW)[6][4], const int16_t *in, int pitch, int y, int prescale, int lane, bool first, bool last)
{
	cfhd_u4 raw[2][2];
#pragma unroll
	for (int k = 0; k < 2; k++) { const int16_t *p = in + (size_t)(y + k) * pitch; raw[k][0] = CFHD_LDG128(p); raw[k][1] = CFHD_LDG128(p + 8); }
#pragma unroll
	for (int k = 0; k < 2; k++) {
		const uint32_t p[8] = { raw[k][0].x, raw[k][0].y, raw[k][0].z, raw[k][0].w, raw[k][1].x, raw[k][1].y, raw[k][1].z, raw[k][1].w };
		const uint32_t prev = __shfl(p[7], lane - 1), next = __shfl(p[0], lane + 1);
		strip_fwd_row(p, prev, next, first, last, LW[SLOT + k], HW[SLOT + k], prescale);
	}
}

// (Occupancy attributes measured in round 6, tools/gpu_r06_t.sh: amdgpu_waves_per_eu(4, 4) here 0.51 -> 0.58 ms, (6, 6) on k_inv_plane_strip 0.47 -> 0.62 ms, (5, 5) on
// k_fwd_yuv422_strip_blocks 1.04 -> 1.06 ms -- the spills cost more than the waves bring; the kernels keep their natural register counts.)
__global__ void __launch_bounds__(NTHREADS) k_fwd_plane_strip(const FwdPlaneJob *jobs, int nframes, int nch, int c0, int nc, int glog, int nstrips, int W, int H, int nseg)
{
	const int lane = threadIdx.x & 63;
	const int nblk = W / (2 * SBLK), HH = H >> 1;
	bool idle;
	const StripWho who = strip_who(nframes, nch, c0, nc, glog, nstrips, nblk, &idle, nseg);
	if (idle) return;                                     // whole wave
	const FwdPlaneJob *job = &jobs[who.job];
	const int blk = who.blk, pitch = job->in_pitch, prescale = job->prescale, out_pitch = job->out_pitch;
	const bool first = blk == 0, last = blk == nblk - 1;
	const int16_t *in = job->in + 2 * SBLK * blk;
	const QuantParam q_lh = job->q[1], q_hl = job->q[2], q_hh = job->q[3];
	int16_t *const out0 = job->out[0] + SBLK * blk, *const out1 = job->out[1] + SBLK * blk, *const out2 = job->out[2] + SBLK * blk, *const out3 = job->out[3] + SBLK * blk;
	const int r0 = who.strip * SRP, r1 = r0 + SRP < HH ? r0 + SRP : HH;
	uint32_t LW[6][4], HW[6][4];
	int wtop = window_first_row(r0, HH, H);
	strip_plane_push<0>(LW, HW, in, pitch, wtop, prescale, lane, first, last);
	strip_plane_push<2>(LW, HW, in, pitch, wtop + 2, prescale, lane, first, last);
	strip_plane_push<4>(LW, HW, in, pitch, wtop + 4, prescale, lane, first, last);
	for (int r = r0; r < r1; r++) {
		const int need = window_first_row(r, HH, H);
		if (need != wtop) {
#pragma unroll
			for (int k = 0; k < 4; k++) {
#pragma unroll
				for (int d = 0; d < 4; d++) { LW[k][d] = LW[k + 2][d]; HW[k][d] = HW[k + 2][d]; }
			}
			wtop = need;
			strip_plane_push<4>(LW, HW, in, pitch, wtop + 4, prescale, lane, first, last);
		}
		uint32_t o[4][4];
		strip_fwd_emit(LW, HW, r == 0 ? 0 : (r == HH - 1 ? 2 : 1), q_lh, q_hl, q_hh, o);
		if (who.stores) {
			const size_t at = (size_t)r * out_pitch;
			uint4 v;
			v.x = o[0][0]; v.y = o[0][1]; v.z = o[0][2]; v.w = o[0][3]; *(uint4 *)(out0 + at) = v;
			v.x = o[1][0]; v.y = o[1][1]; v.z = o[1][2]; v.w = o[1][3]; *(uint4 *)(out1 + at) = v;
			v.x = o[2][0]; v.y = o[2][1]; v.z = o[2][2]; v.w = o[2][3]; *(uint4 *)(out2 + at) = v;
			v.x = o[3][0]; v.y = o[3][1]; v.z = o[3][2]; v.w = o[3][3]; *(uint4 *)(out3 + at) = v;
		}
	}
}

// =============================================================================================
// k_fwd_packed16_strip: level 1 of RG48 / b64a (interleaved 16-bit pixels, FwdPlaneJob layout 0) in the register-strip organisation.
// One lane = 8 pixels of every picture row of its strip, ALL component planes: WPP 16-byte loads bring the pixels in once, the components
// are cut out of the registers (constant word positions), and every plane runs the horizontal 2/6 analysis of its 4 sample pairs
// (neighbour pairs from the adjacent lanes) into its own six-row window; every second row the vertical analysis + quantizer emits one row
// of LL, LH, HL, HH per plane, 8 bytes per band.  A wave covers a segment of 62 blocks (lanes 0 and 63 only feed their neighbours) and
// PSR band rows; no LDS, no barriers.  Same arithmetic as k_fwd_packed16 (tested against it), which stays for other geometries, small
// launches and the other word layouts.  Geometry served: width % 8 == 0, 16-byte aligned rows.
// WPP words per pixel, NCH planes: RG48 (3, 3): planes G R B = words 1 0 2; b64a (4, 4): planes G R B A = words 2 1 3 0; b64a to RGB 4:4:4 (4, 3).
// =============================================================================================
enum { PSR = 32, PSTEP = 62 };
template <int WPP> __device__ __forceinline__ constexpr int packed16_word(int c) { return WPP == 3 ? (c == 0 ? 1 : (c == 1 ? 0 : 2)) : (c == 0 ? 2 : (c == 1 ? 1 : (c == 2 ? 3 : 0))); }

template <int WPP> struct PxRows { cfhd_u4 v[2][WPP]; };     // the lane's 8 pixels of two picture rows
template <int WPP>
__device__ __forceinline__ void px_fetch(PxRows<WPP> &R, const uint16_t *px, int in_pitch, int y, int display_height)
{
#pragma unroll
	for (int k = 0; k < 2; k++) {
		const int yy = y + k < display_height ? y + k : display_height - 1;     // rows beyond the picture repeat the last one (frame.c:6020-6024)
		const uint16_t *p = px + (size_t)yy * in_pitch;
#pragma unroll
		for (int j = 0; j < WPP; j++) R.v[k][j] = CFHD_LDG128(p + 8 * j);
	}
}
// sample pair m (pixels 2m, 2m + 1) of the component at word `word` of every pixel, >> shift
template <int WPP>
__device__ __forceinline__ uint32_t px_pair(const uint32_t (&w)[4 * WPP], int m, int word, int shift, bool compand)
{
	uint32_t s[2];
#pragma unroll
	for (int e = 0; e < 2; e++) {
		const int i = (2 * m + e) * WPP + word;           // 16-bit word of the lane's row
		const uint32_t d = w[i >> 1];
		s[e] = (i & 1) ? d >> (16 + shift) : (d & 0xffffu) >> shift;
		if (compand && s[e] > 0 && s[e] < 4095) s[e] = ((s[e] * 223 + 128) >> 8) + 256;     // alpha, frame.c:6696-6707
	}
	return s[0] | (s[1] << 16);
}
// Two picture rows of every plane -> window slots SLOT, SLOT + 1
template <int WPP, int NCH, int SLOT>
__device__ __forceinline__ void px_push(uint32_t (&LW)[NCH][6][2], uint32_t (&HW)[NCH][6][2], const PxRows<WPP> &R, int shift, bool compand_alpha, int prescale, int lane, bool first, bool last)
{
#pragma unroll
	for (int k = 0; k < 2; k++) {
		uint32_t w[4 * WPP];
#pragma unroll
		for (int j = 0; j < WPP; j++) { w[4 * j] = R.v[k][j].x; w[4 * j + 1] = R.v[k][j].y; w[4 * j + 2] = R.v[k][j].z; w[4 * j + 3] = R.v[k][j].w; }
#pragma unroll
		for (int c = 0; c < NCH; c++) {
			uint32_t ext[6];
#pragma unroll
			for (int m = 0; m < 4; m++) ext[1 + m] = px_pair<WPP>(w, m, packed16_word<WPP>(c), shift, compand_alpha && c == 3);
			ext[0] = __shfl(ext[4], lane - 1); ext[5] = __shfl(ext[1], lane + 1);
#pragma unroll
			for (int m = 0; m < 2; m++) horiz_pair(&ext[2 * m], 0u, prescale, first && m == 0, false, last && m == 1, LW[c][SLOT + k][m], HW[c][SLOT + k][m]);
		}
	}
}

// Two waves per SIMD (180 / 236 registers, nothing spilled) with the next rows' loads in flight beat three or four waves with spills:
// RG48 1.09 vs 1.28 / 3.15 ms per 48 4K frames, b64a 1.03 vs 2.87 ms per 8 8K frames (the tiled kernel: 1.90 / 2.18 ms).
#ifndef CFHD_PX_WAVES
#define CFHD_PX_WAVES 2
#endif
template <int WPP, int NCH>
__global__ void __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(CFHD_PX_WAVES, CFHD_PX_WAVES))) k_fwd_packed16_strip(const FwdPlaneJob *jobs, int nframes, int nseg, int nstrips)
{
	const int lane = threadIdx.x & 63, gwave = (int)blockIdx.x * (NTHREADS / 64) + wave_uniform((int)(threadIdx.x >> 6));     // (scalar: the job fields stay out of the vector registers)
	const int seg = gwave % nseg, strip = (gwave / nseg) % nstrips, frame = gwave / (nseg * nstrips);
	if (frame >= nframes) return;                         // whole wave
	const FwdPlaneJob *job = jobs + (size_t)frame * NCH;
	const int W = job->width, H = job->height, HH = H >> 1, nblk = W / 8;
	const int b = seg * PSTEP - 1 + lane;
	const bool stores = b >= 0 && b < nblk && lane >= 1 && lane <= PSTEP;
	const int blk = b < 0 ? 0 : (b >= nblk ? nblk - 1 : b);
	const bool first = blk == 0, last = blk == nblk - 1;
	const int in_pitch = job->in_pitch, shift = job->shift, dh = job->display_height, prescale = job->prescale, out_pitch = job->out_pitch;
	const bool compand_alpha = NCH == 4 && job[NCH - 1].compand;
	const uint16_t *px = (const uint16_t *)job->in - packed16_word<WPP>(0) + (size_t)blk * 8 * WPP;
	const int r0 = strip * PSR, r1 = r0 + PSR < HH ? r0 + PSR : HH;
	QuantParam q[NCH][3];
	int16_t *out[NCH][4];
#pragma unroll
	for (int c = 0; c < NCH; c++) {
#pragma unroll
		for (int k = 0; k < 3; k++) q[c][k] = job[c].q[1 + k];
#pragma unroll
		for (int k = 0; k < 4; k++) out[c][k] = job[c].out[k];      // (wave-uniform: scalar registers; the lane's place goes into the 32-bit offset below)
	}
	uint32_t LW[NCH][6][2], HW[NCH][6][2];
	int wtop = window_first_row(r0, HH, H);
	PxRows<WPP> R;
	px_fetch<WPP>(R, px, in_pitch, wtop, dh);
	px_push<WPP, NCH, 0>(LW, HW, R, shift, compand_alpha, prescale, lane, first, last);
	px_fetch<WPP>(R, px, in_pitch, wtop + 2, dh);
	px_push<WPP, NCH, 2>(LW, HW, R, shift, compand_alpha, prescale, lane, first, last);
	px_fetch<WPP>(R, px, in_pitch, wtop + 4, dh);
	px_push<WPP, NCH, 4>(LW, HW, R, shift, compand_alpha, prescale, lane, first, last);
	// the rows of the next window step are fetched a band row ahead of their use
	int fetched = -1;
	for (int r = r0; r < r1; r++) {
		const int need = window_first_row(r, HH, H);
		if (need != wtop) {
			if (fetched != need + 4) px_fetch<WPP>(R, px, in_pitch, need + 4, dh);
#pragma unroll
			for (int c = 0; c < NCH; c++) {
#pragma unroll
				for (int k = 0; k < 4; k++) { LW[c][k][0] = LW[c][k + 2][0]; LW[c][k][1] = LW[c][k + 2][1]; HW[c][k][0] = HW[c][k + 2][0]; HW[c][k][1] = HW[c][k + 2][1]; }
			}
			wtop = need;
			px_push<WPP, NCH, 4>(LW, HW, R, shift, compand_alpha, prescale, lane, first, last);
		}
		if (r + 1 < r1) {
			const int next_need = window_first_row(r + 1, HH, H);
			if (next_need != wtop) { px_fetch<WPP>(R, px, in_pitch, next_need + 4, dh); fetched = next_need + 4; }
		}
		const int pos = r == 0 ? 0 : (r == HH - 1 ? 2 : 1);
		const uint32_t at = ((uint32_t)r * (uint32_t)out_pitch + (uint32_t)blk * 4u) * 2u;     // bytes into the band
#pragma unroll
		for (int c = 0; c < NCH; c++) {
			uint32_t o[4][2];
			strip_fwd_emit<2>(LW[c], HW[c], pos, q[c][0], q[c][1], q[c][2], o);
			if (stores) {
#pragma unroll
				for (int bnd = 0; bnd < 4; bnd++) { uint2 v; v.x = o[bnd][0]; v.y = o[bnd][1]; *(uint2 *)((char *)out[c][bnd] + at) = v; }
			}
		}
	}
}

// =============================================================================================
// k_fwd_bayer_strip: level 1 of a 16-bit Bayer mosaic (BYR4) in the register-strip organisation -- k_fwd_packed16_strip with the mosaic as the packed source.
// One lane = 8 photosite quads (16 photosites of two mosaic rows: four 16-byte loads) of every quad row of its strip: every photosite goes through the encode
// curve once (the 14-bit LUT sits in LDS: 32 KB per workgroup, filled once), every quad gives one sample of each of the four component planes (the arithmetic of
// k_unpack_byr4, frame.c:5219-5393), and every plane runs the horizontal 2/6 analysis of its 4 sample pairs into its own six-row window; every second quad row the
// vertical analysis + quantizer emits one row of LL, LH, HL, HH per plane.  The component planes k_unpack_byr4 writes (8 bytes per quad out, 8 back in) never exist.
// Same arithmetic as k_unpack_byr4 + k_fwd_plane (tested against them), which stay for BYR5, other geometries and small launches.
// Geometry served: plane width % 8 == 0, mosaic rows 16-byte aligned.
// =============================================================================================
struct ByRows { cfhd_u4 v[2][2][2]; };      // [quad row][mosaic row of the pair][half of the lane's 16 photosites]
__device__ __forceinline__ void by_fetch(ByRows &R, const uint16_t *mosaic, int in_pitch, int y, int display_height)
{
#pragma unroll
	for (int k = 0; k < 2; k++) {
		const int yy = y + k < display_height ? y + k : display_height - 1;     // quad rows beyond the picture repeat the last one
		const uint16_t *p = mosaic + (size_t)(2 * yy) * in_pitch;
		R.v[k][0][0] = CFHD_LDG128(p); R.v[k][0][1] = CFHD_LDG128(p + 8);
		R.v[k][1][0] = CFHD_LDG128(p + in_pitch); R.v[k][1][1] = CFHD_LDG128(p + in_pitch + 8);
	}
}
template <int SLOT>
__device__ __forceinline__ void by_push(uint32_t (&LW)[4][6][2], uint32_t (&HW)[4][6][2], const ByRows &R, const uint16_t *s_curve, int order, int mid, int prescale, int lane, bool first, bool last)
{
#pragma unroll
	for (int k = 0; k < 2; k++) {
		const uint32_t top[8] = { R.v[k][0][0].x, R.v[k][0][0].y, R.v[k][0][0].z, R.v[k][0][0].w, R.v[k][0][1].x, R.v[k][0][1].y, R.v[k][0][1].z, R.v[k][0][1].w };
		const uint32_t bot[8] = { R.v[k][1][0].x, R.v[k][1][0].y, R.v[k][1][0].z, R.v[k][1][0].w, R.v[k][1][1].x, R.v[k][1][1].y, R.v[k][1][1].z, R.v[k][1][1].w };
		int o[4][8];
#pragma unroll
		for (int q = 0; q < 8; q++) {
			const int tl = s_curve[(top[q] & 0xffffu) >> 2], tr = s_curve[top[q] >> 18], bl = s_curve[(bot[q] & 0xffffu) >> 2], br = s_curve[bot[q] >> 18];
			int r, g1, g2, bb;
			switch (order) {
			case 0: r = tl; g1 = tr; g2 = bl; bb = br; break;
			case 1: g1 = tl; r = tr; bb = bl; g2 = br; break;
			case 3: bb = tl; g1 = tr; g2 = bl; r = br; break;
			default: g1 = tl; bb = tr; r = bl; g2 = br; break;
			}
			const int g = (g1 + g2) >> 1;
			o[0][q] = g; o[1][q] = ((r - g) >> 1) + mid; o[2][q] = ((bb - g) >> 1) + mid; o[3][q] = (g1 - g2 + 2 * mid) >> 1;
		}
#pragma unroll
		for (int c = 0; c < 4; c++) {
			uint32_t ext[6];
#pragma unroll
			for (int m = 0; m < 4; m++) ext[1 + m] = pack16(o[c][2 * m], o[c][2 * m + 1]);
			ext[0] = __shfl(ext[4], lane - 1); ext[5] = __shfl(ext[1], lane + 1);
#pragma unroll
			for (int m = 0; m < 2; m++) horiz_pair(&ext[2 * m], 0u, prescale, first && m == 0, false, last && m == 1, LW[c][SLOT + k][m], HW[c][SLOT + k][m]);
		}
	}
}

__global__ void __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(CFHD_PX_WAVES, CFHD_PX_WAVES))) k_fwd_bayer_strip(const FwdPlaneJob *jobs, const BayerJob *bayer, int nframes, int nseg, int nstrips)
{
	__shared__ uint16_t s_curve[1 << 14];
	{
		const uint4 *src = (const uint4 *)bayer[0].curve;      // (one curve per batch: every frame's job points at it)
		uint4 *dst = (uint4 *)s_curve;
		for (int i = threadIdx.x; i < (1 << 14) * 2 / 16; i += NTHREADS) dst[i] = src[i];
	}
	__syncthreads();
	const int lane = threadIdx.x & 63, gwave = (int)blockIdx.x * (NTHREADS / 64) + wave_uniform((int)(threadIdx.x >> 6));
	const int seg = gwave % nseg, strip = (gwave / nseg) % nstrips, frame = gwave / (nseg * nstrips);
	if (frame >= nframes) return;                         // whole wave (behind the only barrier)
	const FwdPlaneJob *job = jobs + (size_t)frame * 4;
	const BayerJob *bj = bayer + frame;
	const int W = job->width, H = job->height, HH = H >> 1, nblk = W / 8;
	const int b = seg * PSTEP - 1 + lane;
	const bool stores = b >= 0 && b < nblk && lane >= 1 && lane <= PSTEP;
	const int blk = b < 0 ? 0 : (b >= nblk ? nblk - 1 : b);
	const bool first = blk == 0, last = blk == nblk - 1;
	const int in_pitch = bj->in_pitch, dh = bj->display_height, order = bj->order, mid = 1 << (bj->precision - 1), prescale = job->prescale, out_pitch = job->out_pitch;
	const uint16_t *px = bj->in + (size_t)blk * 16;
	const int r0 = strip * PSR, r1 = r0 + PSR < HH ? r0 + PSR : HH;
	QuantParam q[4][3];
	int16_t *out[4][4];
#pragma unroll
	for (int c = 0; c < 4; c++) {
#pragma unroll
		for (int k = 0; k < 3; k++) q[c][k] = job[c].q[1 + k];
#pragma unroll
		for (int k = 0; k < 4; k++) out[c][k] = job[c].out[k];
	}
	uint32_t LW[4][6][2], HW[4][6][2];
	int wtop = window_first_row(r0, HH, H);
	ByRows R;
	by_fetch(R, px, in_pitch, wtop, dh);
	by_push<0>(LW, HW, R, s_curve, order, mid, prescale, lane, first, last);
	by_fetch(R, px, in_pitch, wtop + 2, dh);
	by_push<2>(LW, HW, R, s_curve, order, mid, prescale, lane, first, last);
	by_fetch(R, px, in_pitch, wtop + 4, dh);
	by_push<4>(LW, HW, R, s_curve, order, mid, prescale, lane, first, last);
	int fetched = -1;
	for (int r = r0; r < r1; r++) {
		const int need = window_first_row(r, HH, H);
		if (need != wtop) {
			if (fetched != need + 4) by_fetch(R, px, in_pitch, need + 4, dh);
#pragma unroll
			for (int c = 0; c < 4; c++) {
#pragma unroll
				for (int k = 0; k < 4; k++) { LW[c][k][0] = LW[c][k + 2][0]; LW[c][k][1] = LW[c][k + 2][1]; HW[c][k][0] = HW[c][k + 2][0]; HW[c][k][1] = HW[c][k + 2][1]; }
			}
			wtop = need;
			by_push<4>(LW, HW, R, s_curve, order, mid, prescale, lane, first, last);
		}
		if (r + 1 < r1) {
			const int next_need = window_first_row(r + 1, HH, H);
			if (next_need != wtop) { by_fetch(R, px, in_pitch, next_need + 4, dh); fetched = next_need + 4; }
		}
		const int pos = r == 0 ? 0 : (r == HH - 1 ? 2 : 1);
		const uint32_t at = ((uint32_t)r * (uint32_t)out_pitch + (uint32_t)blk * 4u) * 2u;     // bytes into the band
#pragma unroll
		for (int c = 0; c < 4; c++) {
			uint32_t o[4][2];
			strip_fwd_emit<2>(LW[c], HW[c], pos, q[c][0], q[c][1], q[c][2], o);
			if (stores) {
#pragma unroll
				for (int bnd = 0; bnd < 4; bnd++) { uint2 v; v.x = o[bnd][0]; v.y = o[bnd][1]; *(uint2 *)((char *)out[c][bnd] + at) = v; }
			}
		}
	}
}

// =============================================================================================
// k_inv_packed16_strip: the last level of RGB 4:4:4 -> RG48 / RGBA 4:4:4:4 -> b64a in the register-strip organisation, the mirror image of
// k_fwd_packed16_strip.  One lane = 4 band columns of every plane (8-byte loads of LL, LH, HL, HH; a three-row window of the vertical-lowpass
// bands in registers) -> 8 finished pixels of two output rows, all components, stored as NCH 16-byte words per row.  Segments of 62
// blocks per wave (lanes 0 and 63 only feed their neighbours), QSR band rows per strip; no LDS.  Same arithmetic and the same words as
// k_inv_packed16 (tested against it), which stays for other geometries and small launches.  Geometry served: band width % 4 == 0.
// =============================================================================================
enum { QSR = 16 };
struct InvPxRow { cfhd_u2 ll, lh, hl, hh; };
__device__ __forceinline__ cfhd_u2 ldg64_at(const int16_t *base, uint32_t byte_off) { return CFHD_LDG64((const char *)base + byte_off); }

template <int NCH>
__global__ void __launch_bounds__(NTHREADS) k_inv_packed16_strip(const InvPlaneJob *jobs, int nframes, int nseg, int nstrips)
{
	const int lane = threadIdx.x & 63, gwave = (int)blockIdx.x * (NTHREADS / 64) + wave_uniform((int)(threadIdx.x >> 6));
	const int seg = gwave % nseg, strip = (gwave / nseg) % nstrips, frame = gwave / (nseg * nstrips);
	if (frame >= nframes) return;                         // whole wave
	const InvPlaneJob *job = jobs + (size_t)frame * NCH;
	const int w = job->width, h = job->height, nblk = w / 4;
	const int b = seg * PSTEP - 1 + lane;
	const bool stores = b >= 0 && b < nblk && lane >= 1 && lane <= PSTEP;
	const int blk = b < 0 ? 0 : (b >= nblk ? nblk - 1 : b);
	const bool first = blk == 0, last = blk == nblk - 1;
	const int pitch = job->band_pitch, precision = job->precision, dh = job->display_height, out_pitch = job->out_pitch;
	const bool alpha = NCH == 4 && job[NCH - 1].alpha;
	uint16_t *const frame_out = (uint16_t *)job->out - packed16_word<NCH>(0);
	const int16_t *band[NCH][4];
#pragma unroll
	for (int c = 0; c < NCH; c++) {
#pragma unroll
		for (int k = 0; k < 4; k++) band[c][k] = job[c].band[k];      // (wave-uniform: scalar registers)
	}
	const int tail0 = w - (w & 7) - 9;                    // band columns from here on: the reference's scalar loop (to16)
	const int r0 = strip * QSR, nrows = h - r0 < QSR ? h - r0 : QSR;
	const uint32_t col_off = (uint32_t)blk * 8u;          // bytes into a band row
	int j = inv_window_first_row(r0, h);
	cfhd_u2 ll[NCH][3], lh[NCH][3], hl[NCH], hh[NCH];
#pragma unroll
	for (int c = 0; c < NCH; c++) {
#pragma unroll
		for (int k = 0; k < 3; k++) { const uint32_t at = (uint32_t)(j + k) * (uint32_t)pitch * 2u + col_off; ll[c][k] = ldg64_at(band[c][0], at); lh[c][k] = ldg64_at(band[c][1], at); }
		const uint32_t at = (uint32_t)r0 * (uint32_t)pitch * 2u + col_off;
		hl[c] = ldg64_at(band[c][2], at); hh[c] = ldg64_at(band[c][3], at);
	}
	for (int s = 0; s < nrows; s++) {
		const int r = r0 + s;
		const bool more = s + 1 < nrows;
		const int jn = more ? inv_window_first_row(r + 1, h) : j;
		const bool advance = jn != j;
		// the next band row's loads are in flight while this one is synthesised
		cfhd_u2 nll[NCH], nlh[NCH], nhl[NCH], nhh[NCH];
#pragma unroll
		for (int c = 0; c < NCH; c++) {
			nll[c] = ll[c][2]; nlh[c] = lh[c][2]; nhl[c] = hl[c]; nhh[c] = hh[c];
			if (advance) { const uint32_t at = (uint32_t)(jn + 2) * (uint32_t)pitch * 2u + col_off; nll[c] = ldg64_at(band[c][0], at); nlh[c] = ldg64_at(band[c][1], at); }
			if (more) { const uint32_t at = (uint32_t)(r + 1) * (uint32_t)pitch * 2u + col_off; nhl[c] = ldg64_at(band[c][2], at); nhh[c] = ldg64_at(band[c][3], at); }
		}
		const int pos = r == 0 ? 0 : (r == h - 1 ? 2 : 1);
		uint32_t Lv[NCH][2][2], Hv[NCH][2][2];              // [plane][row parity][column pair]
#pragma unroll
		for (int c = 0; c < NCH; c++) {
			inv_vert_pk(ll[c][0].x, ll[c][1].x, ll[c][2].x, hl[c].x, pos, Lv[c][0][0], Lv[c][1][0]);
			inv_vert_pk(ll[c][0].y, ll[c][1].y, ll[c][2].y, hl[c].y, pos, Lv[c][0][1], Lv[c][1][1]);
			inv_vert_pk(lh[c][0].x, lh[c][1].x, lh[c][2].x, hh[c].x, pos, Hv[c][0][0], Hv[c][1][0]);
			inv_vert_pk(lh[c][0].y, lh[c][1].y, lh[c][2].y, hh[c].y, pos, Hv[c][0][1], Hv[c][1][1]);
		}
#pragma unroll
		for (int par = 0; par < 2; par++) {
			uint32_t px[4 * NCH];                             // the lane's 8 pixels of this output row, two 16-bit words per dword
#pragma unroll
			for (int i = 0; i < 4 * NCH; i++) px[i] = 0u;
#pragma unroll
			for (int c = 0; c < NCH; c++) {
				const uint32_t L0 = Lv[c][par][0], L1 = Lv[c][par][1], H0 = Hv[c][par][0], H1 = Hv[c][par][1];
				const uint32_t prev = __shfl(L1, lane - 1), next = __shfl(L0, lane + 1);
				uint32_t ev[2], od[2];
				inv_horiz_pk((prev >> 16) | (L0 << 16), L0, (L0 >> 16) | (L1 << 16), H0, ev[0], od[0]);
				inv_horiz_pk((L0 >> 16) | (L1 << 16), L1, (L1 >> 16) | (next << 16), H1, ev[1], od[1]);
				int e[4] = { lo16(ev[0]), hi16(ev[0]), lo16(ev[1]), hi16(ev[1]) }, o[4] = { lo16(od[0]), hi16(od[0]), lo16(od[1]), hi16(od[1]) };     // columns 0..3 before the >>1
				if (first) { const int l[6] = { 0, 0, lo16(L0), hi16(L0), lo16(L1), hi16(L1) }; inv_horiz_border(l, 2, lo16(H0), 0, e[0], o[0]); }
				if (last) { const int l[6] = { lo16(L0), hi16(L0), lo16(L1), hi16(L1), 0, 0 }; inv_horiz_border(l, 3, hi16(H1), 2, e[3], o[3]); }
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const bool tail = 4 * blk + k >= tail0;
					uint32_t we = to16(e[k], precision, tail), wo = to16(o[k], precision, tail);
					if (alpha && c == 3) { we = expand_alpha16(we); wo = expand_alpha16(wo); }
					const int ie = (2 * k) * NCH + packed16_word<NCH>(c), io = (2 * k + 1) * NCH + packed16_word<NCH>(c);     // 16-bit word of the lane's row
					px[ie >> 1] |= we << (16 * (ie & 1)); px[io >> 1] |= wo << (16 * (io & 1));
				}
			}
			const int orow = 2 * r + par;
			if (stores && orow < dh) {
				uint4 *dst = (uint4 *)(frame_out + (size_t)orow * out_pitch + (size_t)blk * 8 * NCH);
#pragma unroll
				for (int i = 0; i < NCH; i++) { uint4 v; v.x = px[4 * i]; v.y = px[4 * i + 1]; v.z = px[4 * i + 2]; v.w = px[4 * i + 3]; dst[i] = v; }
			}
		}
#pragma unroll
		for (int c = 0; c < NCH; c++) {
			if (advance) { ll[c][0] = ll[c][1]; ll[c][1] = ll[c][2]; ll[c][2] = nll[c]; lh[c][0] = lh[c][1]; lh[c][1] = lh[c][2]; lh[c][2] = nlh[c]; }
			hl[c] = nhl[c]; hh[c] = nhh[c];
		}
		if (advance) j = jn;
	}
}

// =============================================================================================
// Half-resolution decode of 4:2:2 samples (CFHD_DECODED_RESOLUTION_HALF): the reference does not run the last wavelet level at all
// and shows the level-1 lowpass planes, four times the 10-bit sample, as the picture: decoder.c:14124 -> :22883 CopyLowpass16sToBuffer
// -> frame.c:11742 ConvertLowpass16s10bitToYUV, scalar loop: SATURATE_8U(value >> 4), bytes Y U Y V from the planes Y, U, V (channel
// order of the codec: Y, V, U).  No dither.  One thread = 8 pixels = 16 output bytes.
// =============================================================================================
__device__ __forceinline__ uint32_t half_pair8(uint32_t v)       // two int16 lowpass values -> two 8-bit samples in 16-bit lanes
{
	int a = lo16(v) >> 4, b = hi16(v) >> 4;
	a = a < 0 ? 0 : (a > 255 ? 255 : a); b = b < 0 ? 0 : (b > 255 ? 255 : b);
	return (uint32_t)a | ((uint32_t)b << 16);
}
__global__ void __launch_bounds__(NTHREADS) k_half_yuv422(const HalfYuvJob *jobs)
{
	const HalfYuvJob &job = jobs[blockIdx.z];
	const int row = blockIdx.y, c8 = (int)(blockIdx.x * NTHREADS + threadIdx.x);        // block of 8 pixels
	if (8 * c8 >= job.width) return;
	const cfhd_u4 y = CFHD_LDG128(job.ll[0] + (size_t)row * job.pitch[0] + 8 * c8);
	const uint2 v = *(const uint2 *)(job.ll[1] + (size_t)row * job.pitch[1] + 4 * c8);
	const uint2 u = *(const uint2 *)(job.ll[2] + (size_t)row * job.pitch[2] + 4 * c8);
	const uint32_t yy[4] = { half_pair8(y.x), half_pair8(y.y), half_pair8(y.z), half_pair8(y.w) };      // (y0, y1) ...
	const uint32_t vv[2] = { half_pair8(v.x), half_pair8(v.y) }, uu[2] = { half_pair8(u.x), half_pair8(u.y) };   // (c0, c1), (c2, c3)
	uint32_t o[4];
#pragma unroll
	for (int k = 0; k < 4; k++) {                         // pixel pair k: luma pair k, chroma sample k
		const uint32_t cu = (k & 1) ? uu[k >> 1] >> 16 : uu[k >> 1] & 0xffu, cv = (k & 1) ? vv[k >> 1] >> 16 : vv[k >> 1] & 0xffu;
		const uint32_t y0 = yy[k] & 0xffu, y1 = yy[k] >> 16;
		o[k] = job.uyvy ? (cu | (y0 << 8) | (cv << 16) | (y1 << 24)) : (y0 | (cu << 8) | (y1 << 16) | (cv << 24));
	}
	uint4 q; q.x = o[0]; q.y = o[1]; q.z = o[2]; q.w = o[3];
	*(uint4 *)(job.out + (size_t)row * job.out_pitch + 16 * (size_t)c8) = q;
}

// The same for RGB 4:4:4 -> RG48 and RGBA 4:4:4:4 -> b64a.  RG48: decoder.c:26752 CopyLowpassRGB444ToBuffer -> frame.c:7256
// ConvertLowpassRGB444ToRGB48: value << (16 - precision - 2), saturated to [0, 65535].  b64a goes through the reference's planar
// 16-bit rows as at full resolution (bayer.c:12900: clamped to 14 bits, then shifted; the alpha word expanded like k_inv_packed16's).  Pinned on the reference
// decoder (tests/test_oracle_vs_ref.py).  One thread = 8 pixels = 3 or 4 16-byte words.
__global__ void __launch_bounds__(NTHREADS) k_half_packed16(const HalfPackedJob *jobs)
{
	const HalfPackedJob &job = jobs[blockIdx.z];
	const int row = blockIdx.y, c8 = (int)(blockIdx.x * NTHREADS + threadIdx.x);
	if (8 * c8 >= job.width) return;
	uint32_t px[8][2];                                   // pixel k: words 0, 1 in px[k][0], words 2, 3 in px[k][1]
#pragma unroll
	for (int k = 0; k < 8; k++) px[k][0] = px[k][1] = 0;
	for (int c = 0; c < job.nch; c++) {
		const cfhd_u4 v = CFHD_LDG128(job.ll[c] + (size_t)row * job.pitch + 8 * c8);
		const uint32_t d[4] = { v.x, v.y, v.z, v.w };
		const int wd = job.word[c];
#pragma unroll
		for (int k = 0; k < 8; k++) {
			int x = (k & 1) ? hi16(d[k >> 1]) : lo16(d[k >> 1]);
			// RG48: shifted, then saturated to 65535 (frame.c:7340); the planar 16-bit rows of the b64a route clamp to 14 bits first, so a
			// clipped highlight reads 65532 there (bayer.c:12921-12934)
			if (job.alpha) { const int top = 65535 >> job.shift; x = x < 0 ? 0 : (x > top ? top : x); }
			x <<= job.shift;
			uint32_t w16 = (uint32_t)(x < 0 ? 0 : (x > 65535 ? 65535 : x));
			if (job.alpha && c == 3) w16 = expand_alpha16(w16);
			const uint32_t placed = w16 << (16 * (wd & 1));
			if (wd < 2) px[k][0] |= placed; else px[k][1] |= placed;    // (uniform) keeps the pixel registers statically indexed
		}
	}
	uint8_t *dst = (uint8_t *)job.out + (size_t)row * job.out_pitch;
	if (job.nch == 4) {
		uint4 *o = (uint4 *)(dst + 64 * (size_t)c8);
#pragma unroll
		for (int q = 0; q < 4; q++) { uint4 t; t.x = px[2 * q][0]; t.y = px[2 * q][1]; t.z = px[2 * q + 1][0]; t.w = px[2 * q + 1][1]; o[q] = t; }
	} else {
		// three words per pixel: 8 pixels = 24 words = 12 dwords
		uint32_t w[12];
#pragma unroll
		for (int k = 0; k < 8; k += 2) {                 // pixels k, k + 1 -> 3 dwords: (w0 w1) (w2 | w0') (w1' w2')
			w[3 * (k >> 1)] = px[k][0];
			w[3 * (k >> 1) + 1] = (px[k][1] & 0xffffu) | (px[k + 1][0] << 16);
			w[3 * (k >> 1) + 2] = (px[k + 1][0] >> 16) | (px[k + 1][1] << 16);
		}
		uint4 *o = (uint4 *)(dst + 48 * (size_t)c8);
#pragma unroll
		for (int q = 0; q < 3; q++) { uint4 t; t.x = w[4 * q]; t.y = w[4 * q + 1]; t.z = w[4 * q + 2]; t.w = w[4 * q + 3]; o[q] = t; }
	}
}

// =============================================================================================
// Interlaced level 1 ("frame" wavelet), packed 8-bit 4:2:2 source.  Codec/wavelet.c:6076 TransformForwardFrameYUV: the two rows of
// a pair (one from each field) give the temporal pair low = r0 + r1, high = r1 - r0 of the samples << shift (temporal.c:1915), each
// of which is split horizontally with the 2/6 filter: (LL, LH) from low, (HL, HH) from high.  No vertical filter at this level, so
// a workgroup owns FRW row pairs x 128 luma band columns and needs no halo rows.  HL is quantized inside
// FilterHorizontalRowScaled16sDifferenceFiltered (spatial.c:5327) and stored as the difference to its left neighbour.
// =============================================================================================
enum { FTW = 128, FRW = 4, FDW = FTW + 4 };         // band columns per tile (luma; chroma half), row pairs per tile, staged dwords per row

__global__ void __launch_bounds__(NTHREADS) k_fwd_frame_yuv422(const FwdFrameJob *jobs)
{
	const TileId tile = xcd_tile();
	__shared__ FwdFrameJob s_job;
	stage_job(&s_job, &jobs[tile.z]);
	const FwdFrameJob &job = s_job;
	const int W = job.width, DW = W >> 1, HH = job.height >> 1;     // DW = pixel pairs per row = luma band columns
	const int c0 = tile.x * FTW, r0 = tile.y * FRW;
	// temporal low / high of the tile: luma sample pairs per dword; chroma (V, U) one sample each per pixel pair, kept as pairs of
	// consecutive samples per dword as well: index 0 = low, 1 = high
	__shared__ uint32_t s_y[2][FRW][FDW];
	__shared__ uint32_t s_v[2][FRW][FDW / 2 + 1], s_u[2][FRW][FDW / 2 + 1];
	const bool active = (c0 < DW) && (r0 < HH);
	const int tid = threadIdx.x;
	const int shift = job.shift;
	if (active) {
		// stage: dword d of the tile = pixel pair c0 - 2 + d (even so that chroma pairs stay whole)
		for (int i = tid; i < FRW * FDW; i += NTHREADS) {
			const int rl = i / FDW, d = i - rl * FDW;
			const int r = r0 + rl, dw = c0 - 2 + d;
			uint32_t a = 0x80808080u, b = 0x80808080u;
			if (r < HH && dw >= 0 && dw < DW) {
				const int y0 = 2 * r, y1 = 2 * r + 1;
				if (y0 < job.display_height) a = *(const uint32_t *)(job.in + (size_t)y0 * job.in_pitch + 4 * (size_t)dw);
				if (y1 < job.display_height) b = *(const uint32_t *)(job.in + (size_t)y1 * job.in_pitch + 4 * (size_t)dw);
			}
			const int ysh0 = job.uyvy ? 8 : 0, csh = job.uyvy ? 0 : 8;        // byte lanes: YUYV = Y0 U Y1 V, UYVY = U Y0 V Y1
			const uint32_t ya = pack16((int)((a >> ysh0) & 0xffu) << shift, (int)((a >> (ysh0 + 16)) & 0xffu) << shift);
			const uint32_t yb = pack16((int)((b >> ysh0) & 0xffu) << shift, (int)((b >> (ysh0 + 16)) & 0xffu) << shift);
			s_y[0][rl][d] = pk_adds(ya, yb); s_y[1][rl][d] = pk_subs(yb, ya);
			const int ua = (int)((a >> csh) & 0xffu) << shift, va = (int)((a >> (csh + 16)) & 0xffu) << shift;
			const int ub = (int)((b >> csh) & 0xffu) << shift, vb = (int)((b >> (csh + 16)) & 0xffu) << shift;
			int16_t *sv0 = (int16_t *)&s_v[0][rl][0], *sv1 = (int16_t *)&s_v[1][rl][0], *su0 = (int16_t *)&s_u[0][rl][0], *su1 = (int16_t *)&s_u[1][rl][0];
			sv0[d] = (int16_t)adds16(va, vb); sv1[d] = (int16_t)subs16(vb, va);
			su0[d] = (int16_t)adds16(ua, ub); su1[d] = (int16_t)subs16(ub, ua);
		}
	}
	__syncthreads();
	if (active) {
		// items: (row pair, signal: temporal low / high, channel slot, band column pair).  Luma has FTW/2 pairs, V and U FTW/4 each.
		enum { PAIRS = FTW / 2 + FTW / 4 + FTW / 4 };
		for (int i = tid; i < FRW * 2 * PAIRS; i += NTHREADS) {
			const int rl = i / (2 * PAIRS), rem = i - rl * (2 * PAIRS), sig = rem / PAIRS, slot = rem - sig * PAIRS;
			const int r = r0 + rl;
			if (r >= HH) continue;
			int ch, p, cw, cbase;                      // channel, pair index inside the tile, band width of the channel, first band column of the tile
			const uint32_t *row;
			if (slot < FTW / 2) { ch = 0; p = slot; cw = DW; cbase = c0; row = s_y[sig][rl]; }
			else if (slot < FTW / 2 + FTW / 4) { ch = 1; p = slot - FTW / 2; cw = DW >> 1; cbase = c0 >> 1; row = s_v[sig][rl]; }
			else { ch = 2; p = slot - FTW / 2 - FTW / 4; cw = DW >> 1; cbase = c0 >> 1; row = s_u[sig][rl]; }
			const int c = cbase + 2 * p;               // band column of the pair's first element
			if (c >= cw) continue;
			// samples x[2c-2 .. 2c+5] = sample pairs c-1 .. c+2; luma: staged dword index = c - (c0 - 2); chroma: samples start at pixel pair c0 - 2,
			// i.e. chroma sample pair k sits at dword (k - (c0 - 2) / 2)
			const int k0 = ch == 0 ? (c - (c0 - 2)) : (c - ((c0 - 2) >> 1));
			uint32_t d[4];
#pragma unroll
			for (int k = 0; k < 4; k++) d[k] = row[k0 - 1 + k];
			uint32_t lpk, hpk;
			horiz_pair(d, row[k0 - 2 >= 0 ? k0 - 2 : 0], 0, c == 0, c == cw - 1, c + 1 == cw - 1, lpk, hpk);
			const bool two = c + 1 < cw;
			const size_t o = (size_t)r * job.out_pitch[ch] + c;
			if (sig == 0) {
				// LL as it is, LH through the band quantizer
				const uint32_t lh = pack16(quantize(lo16(hpk), job.q[ch][1]), quantize(hi16(hpk), job.q[ch][1]));
				if (two) { *(uint32_t *)(job.out[ch][0] + o) = lpk; *(uint32_t *)(job.out[ch][1] + o) = lh; }
				else { job.out[ch][0][o] = (int16_t)lo16(lpk); job.out[ch][1][o] = (int16_t)lo16(lh); }
			} else {
				// HL: quantized lowpass of the temporal highpass, stored as the difference to the column on its left (0 in front of column 0)
				const uint32_t prevpair = d[0];
				const int qprev = c == 0 ? 0 : quantize(adds16(lo16(prevpair), hi16(prevpair)), job.q[ch][2]);
				const int q0 = quantize(lo16(lpk), job.q[ch][2]), q1 = quantize(hi16(lpk), job.q[ch][2]);
				const uint32_t hl = pack16(sat16(q0 - qprev), sat16(q1 - q0));
				const uint32_t hh = pack16(quantize(lo16(hpk), job.q[ch][3]), quantize(hi16(hpk), job.q[ch][3]));
				if (two) { *(uint32_t *)(job.out[ch][2] + o) = hl; *(uint32_t *)(job.out[ch][3] + o) = hh; }
				else { job.out[ch][2][o] = (int16_t)lo16(hl); job.out[ch][3][o] = (int16_t)lo16(hh); }
			}
		}
	}
}

// =============================================================================================
// k_inv_frame_yuv422_strip: the interlaced last level in the register-strip organisation -- the workgroup shape, the dither bits and the output stage of
// k_inv_yuv422_strip (two luma waves of 62 blocks, a V and a U wave, the 8-bit samples interleaved through LDS into 16-byte stores), with the inverse frame
// transform in place of the vertical synthesis: the horizontal synthesis of (LL, LH) is the temporal lowpass row, that of (HL, HH) the temporal highpass row
// (spatial.c:19302: the usual 2/6 synthesis, >> 1; packed saturating arithmetic inside the band, the 32-bit border taps at its ends, as k_inv_plane_strip);
// picture row 2r = low - high, row 2r + 1 = low + high (saturating, temporal.c:5961), then 10 -> 8 bits.  Four 16-byte loads per lane and band row, no window.
// k_inv_frame_yuv422(_quad) stay for other geometries and small launches.  Geometry served: as k_inv_yuv422_strip (luma band width % 16 == 0).
// =============================================================================================
enum { SRI = 16 };                                      // band rows (= picture row pairs) per workgroup of the two interlaced strip kernels
__device__ __forceinline__ void strip_row_synth16(const uint32_t (&L)[4], const uint32_t (&H)[4], uint32_t prev, uint32_t next, bool first, bool last, uint32_t (&E)[4], uint32_t (&O)[4])
{
	const uint32_t ext[6] = { prev, L[0], L[1], L[2], L[3], next };
#pragma unroll
	for (int d = 0; d < 4; d++) {
		const uint32_t dm = ext[d], d0 = ext[d + 1], dp = ext[d + 2];
		uint32_t e, o;
		inv_horiz_pk((dm >> 16) | (d0 << 16), d0, (d0 >> 16) | (dp << 16), H[d], e, o);
		E[d] = pk_sra(e, 1); O[d] = pk_sra(o, 1);
	}
	if (first) {
		const int l[6] = { 0, 0, lo16(L[0]), hi16(L[0]), lo16(L[1]), hi16(L[1]) };
		int e, o;
		inv_horiz_border(l, 2, lo16(H[0]), 0, e, o);
		E[0] = (E[0] & 0xffff0000u) | (uint32_t)(uint16_t)sat16(e >> 1); O[0] = (O[0] & 0xffff0000u) | (uint32_t)(uint16_t)sat16(o >> 1);
	}
	if (last) {
		const int l[6] = { lo16(L[2]), hi16(L[2]), lo16(L[3]), hi16(L[3]), 0, 0 };
		int e, o;
		inv_horiz_border(l, 3, hi16(H[3]), 2, e, o);
		E[3] = (E[3] & 0xffffu) | ((uint32_t)(uint16_t)sat16(e >> 1) << 16); O[3] = (O[3] & 0xffffu) | ((uint32_t)(uint16_t)sat16(o >> 1) << 16);
	}
}

template <bool BLOCKS>
__device__ __forceinline__ void inv_frame_yuv422_strip(const InvYuvJob *jobs, uint32_t launch_seed)
{
	const TileId tile = xcd_tile();
	__shared__ InvYuvJob s_job;
	stage_job(&s_job, &jobs[tile.z]);
	const InvYuvJob &job = s_job;
	__shared__ uint32_t s_rows[2][2][SROW / 4];           // [buffer][output row parity][bytes: Y samples of the row | V samples | U samples]
	const uint32_t seed = job.dither_seed ^ launch_seed;
	const int h = job.height, r0 = tile.y * SRI;
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const bool luma = wave < 2;
	const int comp = luma ? 0 : wave - 1;                 // 0 Y, 1 V, 2 U
	const int w = luma ? job.width : job.width >> 1;
	const int nblk = w / SBLK;
	const int seg_first = tile.x * SSEG;
	const int base = luma ? seg_first + SLUMA_STEP * wave : seg_first >> 1;
	const int want = base - 1 + lane;
	const int blk = want < 0 ? 0 : (want < nblk ? want : nblk - 1);
	const bool stores = lane >= 1 && lane <= SLUMA_STEP && want < nblk;
	const int v_base = SSEG * 4, u_base = v_base + SSEG * 2;
	const int lds_at = luma ? 4 * (want - seg_first) : (comp == 1 ? v_base : u_base) + 4 * (want - (seg_first >> 1));
	const int seg_blocks = job.width / SBLK - seg_first < SSEG ? job.width / SBLK - seg_first : SSEG;
	const int nquads = 2 * seg_blocks;
	const bool first = blk == 0, last = blk == nblk - 1;
	const int pitch = job.band_pitch[comp];
	const int16_t *const bLL = wave_uniform_ptr(job.band[comp][0]), *const bLH = wave_uniform_ptr(job.band[comp][1]);
	const int16_t *const bHL = wave_uniform_ptr(job.band[comp][2]), *const bHH = wave_uniform_ptr(job.band[comp][3]);
	const uint32_t boff = (uint32_t)(SBLK * blk);
	if (r0 >= h) return;                                  // whole workgroup
	const int nrows = h - r0 < SRI ? h - r0 : SRI;
	const int sh = job.shift;
	const int dgroup = luma ? (blk & ~1) : 2 * blk + 1;          // dither422_block
	const uint32_t drot = (luma ? (blk & 1) != 0 : comp == 2) ? 8u : 0u;
	for (int s = 0; s < nrows; s++) {
		const int r = r0 + s;
		// (BLOCKS: LH and HH arrive as block lists from the entropy decoder's tile pass, as in k_inv_yuv422_strip_blocks; HL, the difference-coded band, is dense)
		const uint32_t fb = (uint32_t)r * ((uint32_t)pitch / SBLK) + (uint32_t)blk;
		const StripRow ll = strip_load(bLL + boff + (size_t)r * pitch);
		const StripRow lh = BLOCKS ? strip_load_listed(bLH, job.masks + job.mask_base[comp][1], fb) : strip_load(bLH + boff + (size_t)r * pitch);
		const StripRow hl = strip_load(bHL + boff + (size_t)r * pitch);
		const StripRow hh = BLOCKS ? strip_load_listed(bHH, job.masks + job.mask_base[comp][3], fb) : strip_load(bHH + boff + (size_t)r * pitch);
		uint32_t El[4], Ol[4], Eh[4], Oh[4];
		strip_row_synth16(ll.d, lh.d, __shfl(ll.d[3], lane - 1), __shfl(ll.d[0], lane + 1), first, last, El, Ol);
		strip_row_synth16(hl.d, hh.d, __shfl(hl.d[3], lane - 1), __shfl(hl.d[0], lane + 1), first, last, Eh, Oh);
		uint32_t (*rowbuf)[SROW / 4] = s_rows[s & 1];
#pragma unroll
		for (int par = 0; par < 2; par++) {
			const int orow = 2 * r + par;
			// dither bits: as k_inv_yuv422_strip (dither422_block)
			const uint32_t X = sh >= 2 ? rotr32(dither_word(seed, orow, dgroup), drot) : 0u;
			if (stores) {
				uint32_t *dst = &rowbuf[par][lds_at];
#pragma unroll
				for (int d = 0; d < 4; d++) {
					const uint32_t e = par ? pk_adds(El[d], Eh[d]) : pk_subs(El[d], Eh[d]), o = par ? pk_adds(Ol[d], Oh[d]) : pk_subs(Ol[d], Oh[d]);
					dst[d] = pk_to8_bytes(e, o, sh + 1, (X >> d) & 0x20002u, (X >> (d + 4)) & 0x20002u);
				}
			}
		}
		__syncthreads();
		if (tid < nquads) {
#pragma unroll
			for (int par = 0; par < 2; par++) {
				const int orow = 2 * r + par;
				if (orow >= job.display_height) continue;
				const uint32_t *row = rowbuf[par];
				const uint32_t y0 = row[2 * tid], y1 = row[2 * tid + 1], vv = row[v_base + tid], uu = row[u_base + tid];
				const uint32_t uv01 = byte_perm(vv, uu, 0x05010400u), uv23 = byte_perm(vv, uu, 0x07030602u);
				uint4 q;
				if (job.uyvy) {
					q.x = byte_perm(uv01, y0, 0x01050004u); q.y = byte_perm(uv01, y0, 0x03070206u);
					q.z = byte_perm(uv23, y1, 0x01050004u); q.w = byte_perm(uv23, y1, 0x03070206u);
				} else {
					q.x = byte_perm(uv01, y0, 0x05010400u); q.y = byte_perm(uv01, y0, 0x07030602u);
					q.z = byte_perm(uv23, y1, 0x05010400u); q.w = byte_perm(uv23, y1, 0x07030602u);
				}
				*(uint4 *)(job.out + (size_t)orow * job.out_pitch + 32 * (size_t)seg_first + 16 * (size_t)tid) = q;
			}
		}
	}
}

__global__ void __launch_bounds__(NTHREADS) k_inv_frame_yuv422_strip(const InvYuvJob *jobs, uint32_t launch_seed) { inv_frame_yuv422_strip<false>(jobs, launch_seed); }
__global__ void __launch_bounds__(NTHREADS) k_inv_frame_yuv422_strip_blocks(const InvYuvJob *jobs, uint32_t launch_seed) { inv_frame_yuv422_strip<true>(jobs, launch_seed); }

// =============================================================================================
// k_fwd_frame_yuv422_strip: the interlaced level 1 in the register-strip organisation -- the workgroup shape and the unpacking of k_fwd_yuv422_strip (two luma
// waves of 62 blocks each, a V and a U wave fed through LDS; one lane = 8 band columns), with the frame transform in place of the vertical window: the two
// picture rows of a pair give temporal low = r0 + r1 and high = r1 - r0 of every sample pair, each goes through the horizontal 2/6 analysis, and the band
// row leaves at once -- LL, LH = quantized highpass of the low, HL = quantized lowpass of the high stored as the difference to its left neighbour (the
// quantized value of the previous lane's last column comes by lane exchange), HH.  16-byte loads and stores, no window, no halo rows.
// Same words as k_fwd_frame_yuv422 (tested against it), which stays for other geometries and small launches.  Geometry served: as k_fwd_yuv422_strip.
// =============================================================================================
__global__ void __launch_bounds__(NTHREADS) k_fwd_frame_yuv422_strip(const FwdFrameJob *jobs)
{
	const TileId tile = xcd_tile();
	__shared__ FwdFrameJob s_job;
	stage_job(&s_job, &jobs[tile.z]);
	const FwdFrameJob &job = s_job;
	__shared__ uint32_t s_pairs[2][2][2][SFPLANE];        // [buffer][row of the pair][V, U][chroma sample pairs of the row]
	const int W = job.width, HH = job.height >> 1;
	const int r0 = tile.y * SRI;
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const bool luma = wave < 2;
	const int comp = luma ? 0 : wave - 1;                 // 0 Y, 1 V, 2 U
	const QuantParam q_lh = wave_uniform_quant(job.q[comp][1]), q_hl = wave_uniform_quant(job.q[comp][2]), q_hh = wave_uniform_quant(job.q[comp][3]);      // (a wave = one component)
	const int nblk = luma ? W / 16 : W / 32;              // blocks of 8 band columns
	const int seg_first = tile.x * SSEG;
	const int base = luma ? seg_first + SLUMA_STEP * wave : seg_first >> 1;
	const int want = base - 1 + lane;
	const int blk = want < 0 ? 0 : (want < nblk ? want : nblk - 1);
	const bool stores = lane >= 1 && lane <= SLUMA_STEP && want < nblk;
	const bool first = blk == 0, last = blk == nblk - 1;
	const int lds_write_at = (luma && want >= 0 && want < nblk && (wave == 0 ? lane <= SLUMA_STEP : lane >= 1)) ? 4 * (want - seg_first + 1) : -1;
	const int lds_read_at = 8 * lane - 4;
	if (r0 >= HH) return;
	const int r1 = r0 + SRI < HH ? r0 + SRI : HH;
	const int shift = job.shift;
	const int ysh0 = job.uyvy ? 8 : 0, ub = job.uyvy ? 0 : 1, vb = job.uyvy ? 2 : 3;
	const uint32_t usel = (uint32_t)ub | 0x0c00u | ((uint32_t)(4 + ub) << 16) | 0x0c000000u;
	const uint32_t vsel = (uint32_t)vb | 0x0c00u | ((uint32_t)(4 + vb) << 16) | 0x0c000000u;
	const uint8_t *in = job.in + 32 * (size_t)blk;
	const int out_pitch = job.out_pitch[comp];
	int16_t *const o0 = job.out[comp][0] + SBLK * blk, *const o1 = job.out[comp][1] + SBLK * blk, *const o2 = job.out[comp][2] + SBLK * blk, *const o3 = job.out[comp][3] + SBLK * blk;
	const int dh = job.display_height, in_pitch = job.in_pitch;
	for (int r = r0; r < r1; r++) {
		uint32_t (*buf)[2][SFPLANE] = s_pairs[(r - r0) & 1];
		uint32_t p[2][8];
		if (luma) {
			cfhd_u4 raw[2][2];
#pragma unroll
			for (int k = 0; k < 2; k++) {
				const int y = 2 * r + k;
				if (y < dh) { const uint8_t *src = in + (size_t)y * in_pitch; raw[k][0] = CFHD_LDG128(src); raw[k][1] = CFHD_LDG128(src + 16); }
				else { cfhd_u4 g; g.x = g.y = g.z = g.w = 0x80808080u; raw[k][0] = g; raw[k][1] = g; }
			}
#pragma unroll
			for (int k = 0; k < 2; k++) {
				const uint32_t a[8] = { raw[k][0].x, raw[k][0].y, raw[k][0].z, raw[k][0].w, raw[k][1].x, raw[k][1].y, raw[k][1].z, raw[k][1].w };
#pragma unroll
				for (int i = 0; i < 8; i++) p[k][i] = ((a[i] >> ysh0) & 0x00ff00ffu) << shift;
				if (lds_write_at >= 0) {
#pragma unroll
					for (int i = 0; i < 4; i++) {
						buf[k][0][lds_write_at + i] = byte_perm(a[2 * i + 1], a[2 * i], vsel) << shift;
						buf[k][1][lds_write_at + i] = byte_perm(a[2 * i + 1], a[2 * i], usel) << shift;
					}
				}
			}
		}
		__syncthreads();                                  // (one per row pair: the buffers alternate, so the next pair's writes cannot pass this pair's reads)
		if (!luma) {
#pragma unroll
			for (int k = 0; k < 2; k++) {
#pragma unroll
				for (int i = 0; i < 8; i++) p[k][i] = buf[k][comp - 1][lds_read_at + i < 0 ? 0 : lds_read_at + i];
			}
		}
		// temporal pair of every sample (temporal.c:1915), then the horizontal analysis of both
		uint32_t lo[8], hi[8];
#pragma unroll
		for (int i = 0; i < 8; i++) { lo[i] = pk_adds(p[0][i], p[1][i]); hi[i] = pk_subs(p[1][i], p[0][i]); }
		uint32_t LL[4], LH[4], HLraw[4], HH[4];
		strip_fwd_row(lo, __shfl(lo[7], lane - 1), __shfl(lo[0], lane + 1), first, last, LL, LH);
		strip_fwd_row(hi, __shfl(hi[7], lane - 1), __shfl(hi[0], lane + 1), first, last, HLraw, HH);
		uint32_t hlq[4];
#pragma unroll
		for (int d = 0; d < 4; d++) { LH[d] = pk_quantize(LH[d], q_lh); HH[d] = pk_quantize(HH[d], q_hh); hlq[d] = pk_quantize(HLraw[d], q_hl); }
		// HL: the difference to the quantized value on its left, 0 in front of column 0 (spatial.c:5327 FilterHorizontalRowScaled16sDifferenceFiltered)
		uint32_t left = __shfl(hlq[3], lane - 1) >> 16;
		if (first) left = 0u;
		uint32_t HL[4];
#pragma unroll
		for (int d = 0; d < 4; d++) {
			const uint32_t shifted = (hlq[d] << 16) | (left & 0xffffu);      // (column 2d - 1, column 2d)
			HL[d] = pk_subs(hlq[d], shifted);
			left = hlq[d] >> 16;
		}
		if (stores) {
			const size_t at = (size_t)r * out_pitch;
			uint4 v;
			v.x = LL[0]; v.y = LL[1]; v.z = LL[2]; v.w = LL[3]; *(uint4 *)(o0 + at) = v;
			v.x = LH[0]; v.y = LH[1]; v.z = LH[2]; v.w = LH[3]; *(uint4 *)(o1 + at) = v;
			v.x = HL[0]; v.y = HL[1]; v.z = HL[2]; v.w = HL[3]; *(uint4 *)(o2 + at) = v;
			v.x = HH[0]; v.y = HH[1]; v.z = HH[2]; v.w = HH[3]; *(uint4 *)(o3 + at) = v;
		}
	}
}

// =============================================================================================
// Interlaced last level: the inverse of k_fwd_frame_yuv422 (Codec/decoder.c:21493 TransformInverseFrameToYUV, :24304 threaded;
// Codec/temporal.c:5961 InvertInterlacedRow16s10bitToYUV, :6498 ToUYVY).  Band row r of the level-1 wavelet gives two picture rows: the
// horizontal synthesis of (LL, LH) is the temporal lowpass row, that of (HL, HH) -- HL un-differenced by k_dec_undiff -- the temporal
// highpass row (spatial.c:19302 InvertHorizontalRow16s8sTo16sBuffered: the usual 2/6 synthesis, >> 1, saturated); row 2r = low - high,
// row 2r + 1 = low + high (saturating), then 10 -> 8 bits like every other 4:2:2 output.  No vertical filter, hence no halo rows and no
// LDS: one thread per chroma column (two luma columns: 4 + 2 + 2 samples = 8 bytes of each of the two rows), any width.
// =============================================================================================
__device__ __forceinline__ void frame_synth(const int16_t *lo, const int16_t *hi, int c, int w, int &even, int &odd)
{
	const int l0 = lo[c], h = hi[c];
	int e, o;
	if (c == 0) inv_horiz(0, l0, lo[1], lo[2], h, 0, e, o);
	else if (c == w - 1) inv_horiz(lo[c - 1], l0, 0, lo[c - 2], h, 2, e, o);
	else inv_horiz(lo[c - 1], l0, lo[c + 1], 0, h, 1, e, o);
	even = sat16(e >> 1); odd = sat16(o >> 1);
}

__global__ void __launch_bounds__(NTHREADS) k_inv_frame_yuv422(const InvYuvJob *jobs, uint32_t launch_seed)
{
	const InvYuvJob &job = jobs[blockIdx.z];
	const int w = job.width, cw = w >> 1;                // luma / chroma band columns
	const int cc = (int)(blockIdx.x * NTHREADS + threadIdx.x), r = blockIdx.y;
	if (cc >= cw || r >= job.height) return;
	const uint32_t seed = job.dither_seed ^ launch_seed;
	const int sh = job.shift;
	// temporal low / high samples: 4 luma (columns 2cc, 2cc + 1: even, odd each), 2 V, 2 U
	int tl[8], th[8];
	{
		const size_t o = (size_t)r * job.band_pitch[0];
		frame_synth(job.band[0][0] + o, job.band[0][1] + o, 2 * cc, w, tl[0], tl[1]);
		frame_synth(job.band[0][0] + o, job.band[0][1] + o, 2 * cc + 1, w, tl[2], tl[3]);
		frame_synth(job.band[0][2] + o, job.band[0][3] + o, 2 * cc, w, th[0], th[1]);
		frame_synth(job.band[0][2] + o, job.band[0][3] + o, 2 * cc + 1, w, th[2], th[3]);
	}
#pragma unroll
	for (int x = 0; x < 2; x++) {                         // V, U
		const size_t o = (size_t)r * job.band_pitch[1 + x];
		frame_synth(job.band[1 + x][0] + o, job.band[1 + x][1] + o, cc, cw, tl[4 + 2 * x], tl[5 + 2 * x]);
		frame_synth(job.band[1 + x][2] + o, job.band[1 + x][3] + o, cc, cw, th[4 + 2 * x], th[5 + 2 * x]);
	}
#pragma unroll
	for (int par = 0; par < 2; par++) {
		const int orow = 2 * r + par;
		if (orow >= job.display_height) continue;
		const uint32_t dz = sh >= 2 ? dither422_column(seed, orow, cc) : 0u;      // same bit assignment as the progressive kernels
		uint32_t b[8];
#pragma unroll
		for (int i = 0; i < 8; i++) {
			const int v = par ? adds16(tl[i], th[i]) : subs16(tl[i], th[i]);
			b[i] = to8(v, sh, (int)((dz >> (i == 5 ? 6 : (i == 6 ? 5 : i))) & 1u));     // (b[] holds v0, v1, u0, u1 behind the luma samples, the dither byte v0, u0, v1, u1); the temporal lowpass is the sum of the two rows: clamp at zero, halve, dither, >> shift, saturate (temporal.c:6071-6120)
		}
		const uint32_t y0 = b[0], y1 = b[1], y2 = b[2], y3 = b[3], v0 = b[4], v1 = b[5], u0 = b[6], u1 = b[7];
		uint2 o2;
		if (job.uyvy) { o2.x = u0 | (y0 << 8) | (v0 << 16) | (y1 << 24); o2.y = u1 | (y2 << 8) | (v1 << 16) | (y3 << 24); }
		else { o2.x = y0 | (u0 << 8) | (y1 << 16) | (v0 << 24); o2.y = y2 | (u1 << 8) | (y3 << 16) | (v1 << 24); }
		*(uint2 *)(job.out + (size_t)orow * job.out_pitch + 8 * (size_t)cc) = o2;
	}
}

// The same, four luma band columns (eight pixels of both picture rows) per thread: the bands are read with 8-byte loads, the columns next
// to a thread's four come from its neighbours' registers (the first and last lane of a wave fetch theirs), and each picture row leaves as one
// 16-byte store per thread.  Same bytes as k_inv_frame_yuv422 (same dither bit per sample); needs the luma band width to be a multiple of 4
// and 8-byte aligned band rows, which every frame the codec accepts has.
struct Quad16 { int v[4]; };
__device__ __forceinline__ Quad16 quad_load(const int16_t *p) { const uint2 q = *(const uint2 *)p; Quad16 r; r.v[0] = lo16(q.x); r.v[1] = hi16(q.x); r.v[2] = lo16(q.y); r.v[3] = hi16(q.y); return r; }
__device__ __forceinline__ Quad16 pair_load(const int16_t *p) { const uint32_t q = *(const uint32_t *)p; Quad16 r; r.v[0] = lo16(q); r.v[1] = hi16(q); r.v[2] = 0; r.v[3] = 0; return r; }
// horizontal synthesis of n (4 or 2) adjacent columns starting at band column c: out[2k], out[2k + 1] = even, odd sample of column c + k
template <int N>
__device__ __forceinline__ void frame_synth_run(const Quad16 &lo, const Quad16 &hi, int left, int right, int c, int w, int *out)
{
#pragma unroll
	for (int k = 0; k < N; k++) {
		const int col = c + k;
		const int lm1 = k ? lo.v[k - 1] : left, lp1 = k + 1 < N ? lo.v[k + 1] : right;
		int e, o;
		if (col == 0) inv_horiz(0, lo.v[k], lp1, k + 2 < N ? lo.v[k + 2] : right, hi.v[k], 0, e, o);      // (k + 2 < N: always, the first column is a thread's first)
		else if (col == w - 1) inv_horiz(lm1, lo.v[k], 0, k >= 2 ? lo.v[k - 2] : left, hi.v[k], 2, e, o);
		else inv_horiz(lm1, lo.v[k], lp1, 0, hi.v[k], 1, e, o);
		out[2 * k] = sat16(e >> 1); out[2 * k + 1] = sat16(o >> 1);
	}
}
__global__ void __launch_bounds__(NTHREADS) k_inv_frame_yuv422_quad(const InvYuvJob *jobs, uint32_t launch_seed)
{
	const InvYuvJob &job = jobs[blockIdx.z];
	const int w = job.width, cw = w >> 1;
	const int t = (int)(blockIdx.x * NTHREADS + threadIdx.x), r = blockIdx.y;
	const int c = 4 * t, cc = 2 * t;                          // first luma / chroma band column of this thread
	const bool have = c < w && r < job.height;
	const int lane = (int)(threadIdx.x & 63u);
	const uint32_t seed = job.dither_seed ^ launch_seed;
	const int sh = job.shift;
	int tl[16], th[16];                                       // temporal low / high samples: 8 luma, 4 V, 4 U
	// one (lowpass, highpass) band pair at a time: luma (LL, LH) -> temporal low, (HL, HH) -> temporal high, then the same for V and U
#pragma unroll
	for (int pair = 0; pair < 2; pair++) {
		int *dst = pair ? th : tl;
		{
			const size_t o = (size_t)r * job.band_pitch[0];
			const int16_t *lo = job.band[0][2 * pair] + o, *hi = job.band[0][2 * pair + 1] + o;
			Quad16 L = { { 0, 0, 0, 0 } }, H = { { 0, 0, 0, 0 } };
			if (have) { L = quad_load(lo + c); H = quad_load(hi + c); }
			int left = __shfl_up(L.v[3], 1u), right = __shfl_down(L.v[0], 1u);
			if (have && lane == 0 && c > 0) left = lo[c - 1];
			if (have && (lane == 63 || c + 4 >= w) && c + 4 < w) right = lo[c + 4];
			if (have) frame_synth_run<4>(L, H, left, right, c, w, dst);
		}
#pragma unroll
		for (int x = 0; x < 2; x++) {
			const size_t o = (size_t)r * job.band_pitch[1 + x];
			const int16_t *lo = job.band[1 + x][2 * pair] + o, *hi = job.band[1 + x][2 * pair + 1] + o;
			Quad16 L = { { 0, 0, 0, 0 } }, H = { { 0, 0, 0, 0 } };
			if (have) { L = pair_load(lo + cc); H = pair_load(hi + cc); }
			int left = __shfl_up(L.v[1], 1u), right = __shfl_down(L.v[0], 1u);
			if (have && lane == 0 && cc > 0) left = lo[cc - 1];
			if (have && (lane == 63 || cc + 2 >= cw) && cc + 2 < cw) right = lo[cc + 2];
			if (have) {
				// two columns: the far taps of the first / last column of the band lie outside a pair
				int out4[4];
#pragma unroll
				for (int k = 0; k < 2; k++) {
					const int col = cc + k;
					const int lm1 = k ? L.v[0] : left, lp1 = k ? right : L.v[1];
					int e, od;
					if (col == 0) inv_horiz(0, L.v[0], L.v[1], right, H.v[0], 0, e, od);
					else if (col == cw - 1) inv_horiz(L.v[0], L.v[1], 0, left, H.v[1], 2, e, od);
					else inv_horiz(lm1, L.v[k], lp1, 0, H.v[k], 1, e, od);
					out4[2 * k] = sat16(e >> 1); out4[2 * k + 1] = sat16(od >> 1);
				}
#pragma unroll
				for (int k = 0; k < 4; k++) dst[8 + 4 * x + k] = out4[k];
			}
		}
	}
	if (!have) return;
#pragma unroll
	for (int par = 0; par < 2; par++) {
		const int orow = 2 * r + par;
		if (orow >= job.display_height) continue;
		uint32_t words[4];
#pragma unroll
		for (int half = 0; half < 2; half++) {                 // chroma column cc + half: the bits the one-column kernel uses for it
			const int ccol = cc + half;
			const uint32_t dz = sh >= 2 ? dither422_column(seed, orow, ccol) : 0u;
			uint32_t b[8];
#pragma unroll
			for (int i = 0; i < 8; i++) {
				// sample order of the one-column kernel: four luma (columns 2 ccol, 2 ccol + 1: even, odd each), two V, two U
				const int idx = i < 4 ? 4 * half + i : (i < 6 ? 8 + 2 * half + (i - 4) : 12 + 2 * half + (i - 6));
				const int v = par ? adds16(tl[idx], th[idx]) : subs16(tl[idx], th[idx]);
				b[i] = to8(v, sh, (int)((dz >> (i == 5 ? 6 : (i == 6 ? 5 : i))) & 1u));
			}
			const uint32_t y0 = b[0], y1 = b[1], y2 = b[2], y3 = b[3], v0 = b[4], v1 = b[5], u0 = b[6], u1 = b[7];
			if (job.uyvy) { words[2 * half] = u0 | (y0 << 8) | (v0 << 16) | (y1 << 24); words[2 * half + 1] = u1 | (y2 << 8) | (v1 << 16) | (y3 << 24); }
			else { words[2 * half] = y0 | (u0 << 8) | (y1 << 16) | (v0 << 24); words[2 * half + 1] = y2 | (u1 << 8) | (y3 << 16) | (v1 << 24); }
		}
		uint4 o4; o4.x = words[0]; o4.y = words[1]; o4.z = words[2]; o4.w = words[3];
		*(uint4 *)(job.out + (size_t)orow * job.out_pitch + 16 * (size_t)t) = o4;
	}
}

// =============================================================================================
// Bayer input (ConvertBYR4ToFrame16s, frame.c:4993, curve branch :5219-5393): every 2x2 quad of the mosaic gives one sample of the
// four component planes: the encode curve LUT is applied to each photosite (>> 2 to the LUT's 14 bits), then g = (g1+g2)>>1,
// rg = ((r-g)>>1) + mid, bg = ((b-g)>>1) + mid, gd = (g1-g2+2*mid)>>1 with mid = 2^(precision-1).  One lane per quad; a wave reads
// 256 consecutive bytes of each of the two mosaic rows.  The planes then go through k_fwd_plane like any other level.
// =============================================================================================
__global__ void __launch_bounds__(NTHREADS) k_unpack_byr4(const BayerJob *jobs)
{
	__shared__ BayerJob s_job;
	stage_job(&s_job, &jobs[blockIdx.z]);
	const BayerJob &job = s_job;
	// two quads per thread: 8-byte loads of both mosaic rows, 4-byte stores into the four planes (plane widths are multiples of 8)
	const int x = 2 * (int)(blockIdx.x * NTHREADS + threadIdx.x), row = blockIdx.y;
	if (x >= job.width || row >= job.height) return;
	const int srow = row < job.display_height ? row : job.display_height - 1;
	int tl[2], tr[2], bl[2], br[2];
	if (job.packed12) {
		const uint8_t *base = (const uint8_t *)job.in + (size_t)srow * job.width * 6, *nib = base + (size_t)job.width * 4;
#pragma unroll
		for (int q = 0; q < 2; q++) {
			int v[4];
#pragma unroll
			for (int k = 0; k < 4; k++) { const int s = k * job.width + x + q; v[k] = ((int)base[s] << 4) | ((nib[s >> 1] >> (4 * (s & 1))) & 15); }
			tl[q] = v[0]; tr[q] = v[1]; bl[q] = v[2]; br[q] = v[3];      // the four runs in the order of a row pair's photosites
		}
	} else {
		const uint16_t *l1 = job.in + (size_t)(2 * srow) * job.in_pitch, *l2 = l1 + job.in_pitch;
		const uint2 a = *(const uint2 *)(l1 + 2 * x), b = *(const uint2 *)(l2 + 2 * x);      // (left, right) photosites of the two rows, two quads
		tl[0] = job.curve[(a.x & 0xffffu) >> 2]; tr[0] = job.curve[a.x >> 18]; bl[0] = job.curve[(b.x & 0xffffu) >> 2]; br[0] = job.curve[b.x >> 18];
		tl[1] = job.curve[(a.y & 0xffffu) >> 2]; tr[1] = job.curve[a.y >> 18]; bl[1] = job.curve[(b.y & 0xffffu) >> 2]; br[1] = job.curve[b.y >> 18];
	}
	const int mid = 1 << (job.precision - 1);
	int o0[2], o1[2], o2[2], o3[2];
#pragma unroll
	for (int q = 0; q < 2; q++) {
		int r, g1, g2, bb;
		switch (job.order) {
		case 0: r = tl[q]; g1 = tr[q]; g2 = bl[q]; bb = br[q]; break;
		case 1: g1 = tl[q]; r = tr[q]; bb = bl[q]; g2 = br[q]; break;
		case 3: bb = tl[q]; g1 = tr[q]; g2 = bl[q]; r = br[q]; break;
		default: g1 = tl[q]; bb = tr[q]; r = bl[q]; g2 = br[q]; break;
		}
		const int g = (g1 + g2) >> 1;
		o0[q] = g; o1[q] = ((r - g) >> 1) + mid; o2[q] = ((bb - g) >> 1) + mid; o3[q] = (g1 - g2 + 2 * mid) >> 1;
	}
	const size_t o = (size_t)row * job.out_pitch + x;
	*(uint32_t *)(job.out[0] + o) = pack16(o0[0], o0[1]);
	*(uint32_t *)(job.out[1] + o) = pack16(o1[0], o1[1]);
	*(uint32_t *)(job.out[2] + o) = pack16(o2[0], o2[1]);
	*(uint32_t *)(job.out[3] + o) = pack16(o3[0], o3[1]);
}

// =============================================================================================
// Two-frame group (cfhd_gop.h): the temporal step between the level-1 lowpass bands of the two frames
// =============================================================================================
// Forward (Codec/temporal.c:498 FilterTemporal16s): low = sat16(a + b), high = sat16(b - a) -- _mm_adds_epi16 / _mm_subs_epi16 in the reference,
// v_pk_add_i16 / v_pk_sub_i16 with clamp here.  The planes of a job share pitch and height (pad columns are zero in, zero out).
// Inverse (Codec/wavelet.c TransformInverseTemporal): frame 0 = sat16(low - high) >> 1, frame 1 = sat16(low + high) >> 1 in the columns the
// reference's 8-wide loop covers; the `width % 8` tail columns divide by two towards zero instead (its scalar loop).
struct GopTemporalJob { const int16_t *a, *b; int16_t *x, *y; int pitch, height, width; };
__global__ void __launch_bounds__(NTHREADS) k_gop_temporal_fwd(const GopTemporalJob *jobs)
{
	const GopTemporalJob job = jobs[blockIdx.y];
	const int ndw = job.pitch * job.height / 2;
	const uint32_t *a = (const uint32_t *)job.a, *b = (const uint32_t *)job.b;
	uint32_t *lo = (uint32_t *)job.x, *hi = (uint32_t *)job.y;
	for (int i = (int)(blockIdx.x * NTHREADS + threadIdx.x); i < ndw; i += (int)(gridDim.x * NTHREADS)) {
		const uint32_t va = a[i], vb = b[i];
		lo[i] = pk_adds(va, vb);
		hi[i] = pk_subs(vb, va);
	}
}
__global__ void __launch_bounds__(NTHREADS) k_gop_temporal_inv(const GopTemporalJob *jobs)
{
	const GopTemporalJob job = jobs[blockIdx.y];
	const int ndw = job.pitch * job.height / 2, row_dw = job.pitch / 2;
	const int tail_from = job.width - job.width % 8;       // first column of the reference's scalar loop
	const uint32_t *lo = (const uint32_t *)job.a, *hi = (const uint32_t *)job.b;
	uint32_t *f0 = (uint32_t *)job.x, *f1 = (uint32_t *)job.y;
	for (int i = (int)(blockIdx.x * NTHREADS + threadIdx.x); i < ndw; i += (int)(gridDim.x * NTHREADS)) {
		const uint32_t l = lo[i], h = hi[i];
		uint32_t even = pk_sra(pk_subs(l, h), 1), odd = pk_sra(pk_adds(l, h), 1);
		const int col = 2 * (i % row_dw);
		if (col >= tail_from) {                               // (low +- high) / 2 in 32-bit arithmetic, rounded towards zero
			const int l0 = (int)(int16_t)(l & 0xffffu), l1 = (int)(int16_t)(l >> 16), h0 = (int)(int16_t)(h & 0xffffu), h1 = (int)(int16_t)(h >> 16);
			even = (uint32_t)(uint16_t)(int16_t)((l0 - h0) / 2) | ((uint32_t)(uint16_t)(int16_t)((l1 - h1) / 2) << 16);
			odd = (uint32_t)(uint16_t)(int16_t)((l0 + h0) / 2) | ((uint32_t)(uint16_t)(int16_t)((l1 + h1) / 2) << 16);
		}
		f0[i] = even; f1[i] = odd;
	}
}

// v210 output (DecodeBatch): six pixels of a YU64 row (words Y0 C1 Y1 C2 per pixel pair, C1 = channel 1 = Cr, C2 = channel 2 = Cb) >> 6 into four
// 32-bit words Cb0 Y0 Cr0 | Y1 Cb1 Y2 | Cr1 Y3 Cb2 | Y4 Cr2 Y5, low bits first (oracle/cfhd_oracle_inv.c orc_inv_spatial_to_v210).  One thread per group.
__global__ void __launch_bounds__(NTHREADS) k_yu64_to_v210(const uint16_t *yu64, int in_pitch_words, size_t in_frame_words, uint32_t *out, int out_pitch_words, size_t out_frame_words, int groups)
{
	const int g = (int)(blockIdx.x * NTHREADS + threadIdx.x);
	if (g >= groups) return;
	const uint16_t *r = yu64 + (size_t)blockIdx.z * in_frame_words + (size_t)blockIdx.y * in_pitch_words + 12 * (size_t)g;
	uint32_t *o = out + (size_t)blockIdx.z * out_frame_words + (size_t)blockIdx.y * out_pitch_words + 4 * (size_t)g;
	uint32_t Y[6], Cb[3], Cr[3];
#pragma unroll
	for (int k = 0; k < 6; k++) Y[k] = (uint32_t)r[2 * k] >> 6;
#pragma unroll
	for (int k = 0; k < 3; k++) { Cr[k] = (uint32_t)r[4 * k + 1] >> 6; Cb[k] = (uint32_t)r[4 * k + 3] >> 6; }
	o[0] = Cb[0] | (Y[0] << 10) | (Cr[0] << 20);
	o[1] = Y[1] | (Cb[1] << 10) | (Y[2] << 20);
	o[2] = Cr[1] | (Y[3] << 10) | (Cb[2] << 20);
	o[3] = Y[4] | (Cr[2] << 10) | (Y[5] << 20);
}

// RG24 output of 4:2:2 samples (DecodeBatch): a pixel pair of a YU64 row (words Y0 C1 Y1 C2, C1 = channel 1 = Cr, C2 = channel 2 = Cb) through the reference's
// scalar conversion (convert.c:11392-11448 ConvertRow16uToDitheredRGB; oracle/cfhd_oracle_inv.c orc_inv_spatial_to_rgb24_of_yuv422): 15-bit dither per pixel,
// shared by its three components, from the counter-based hash that stands in for rand(); bytes B, G, R, bottom row first.  matrix: 0 computer-systems 709,
// 1 video 709, 2 computer 601, 3 video 601.  One thread per pixel pair.
// RG48 / b64a output of 4:2:2 samples: the YU64 rows (k_inv_packed16 into the scratch frame) through the reference's 16-bit colour conversion -- RGB2YUV.c:1308
// ChannelYUYV16toPlanarYUV16 (every chroma word serves its pixel pair) + :1760 PlanarYUV16toPlanarRGB16 (vector body: 15-bit samples, 13-bit coefficients, mulhi
// products, saturating sums, the clamp to 14 bits) + bayer.c:478 ConvertLinesToOutput at white point 16 (the words as they are; b64a: 0xffff in front).  Restated in
// oracle/cfhd_oracle_inv.c orc_yu64_to_rgb16, pinned on the reference decoder.  One thread per pixel pair: 8 bytes in, 12 or 16 out.
__global__ void __launch_bounds__(NTHREADS) k_yu64_to_rgb16(const uint16_t *yu64, int in_pitch_words, size_t in_frame_words, uint16_t *out, int out_pitch_words, size_t out_frame_words,
                                                            int pairs, int matrix_601, int b64a)
{
	const int p = (int)(blockIdx.x * blockDim.x + threadIdx.x), row = (int)blockIdx.y;
	if (p >= pairs) return;
	const cfhd_u2 in = CFHD_LDG64(yu64 + blockIdx.z * in_frame_words + (size_t)row * in_pitch_words + 4 * (size_t)p);      // Y0 C1 | Y1 C2: channel 1 = V, channel 2 = U
	const int Y[2] = { (int)(in.x & 0xffffu), (int)(in.y & 0xffffu) }, V = (int)(in.x >> 16), U = (int)(in.y >> 16);
	const int y_offset = 2048 + (matrix_601 ? -28 : -32), ymult = 9535 + (matrix_601 ? 14 : 11);
	const int r_vmult = (matrix_601 ? 13074 : 14688) + 6, g_vmult = matrix_601 ? 6661 : 4357, g_umult = matrix_601 ? 3210 : 1738, b_umult = matrix_601 ? 16534 : 17326;
	const int c_offset = (1 << 14) + (matrix_601 ? 23 : 22);
	const int uu = sat16((U >> 1) - c_offset), vv = sat16((V >> 1) - c_offset);
	const int rv = (vv * r_vmult) >> 16, gu = (uu * -g_umult) >> 16, gv = (vv * -g_vmult) >> 16, bu = (uu * b_umult) >> 16;
	uint16_t *o = out + blockIdx.z * out_frame_words + (size_t)row * out_pitch_words + (size_t)p * (b64a ? 8 : 6);
#pragma unroll
	for (int k = 0; k < 2; k++) {
		const int yy = (sat16((Y[k] >> 1) - y_offset) * ymult) >> 16;
		int comp[3] = { adds16(rv, yy), adds16(adds16(yy, gu), gv), adds16(bu, yy) };
#pragma unroll
		for (int c = 0; c < 3; c++) {
			int v = (int)(int16_t)(comp[c] << 2);
			v = adds16(v, 0x7fff - 0x3fff);
			v = (v & 0xffff) - (0x7fff - 0x3fff); v = v < 0 ? 0 : v;
			comp[c] = (v << 2) & 0xffff;
		}
		if (b64a) { o[4 * k] = 0xffff; o[4 * k + 1] = (uint16_t)comp[0]; o[4 * k + 2] = (uint16_t)comp[1]; o[4 * k + 3] = (uint16_t)comp[2]; }
		else { o[3 * k] = (uint16_t)comp[0]; o[3 * k + 1] = (uint16_t)comp[1]; o[3 * k + 2] = (uint16_t)comp[2]; }
	}
}

__global__ void __launch_bounds__(NTHREADS) k_yu64_to_rgb24(const uint16_t *yu64, int in_pitch_words, size_t in_frame_words, uint8_t *out, int out_pitch, size_t out_frame_bytes,
                                                            int pairs, int rows, int matrix, uint32_t seed)
{
	const int p = (int)(blockIdx.x * NTHREADS + threadIdx.x);
	if (p >= pairs) return;
	const int row = (int)blockIdx.y;
	const int m[4][6] = { { 16, 128 * 149, 230, 137, 55, 135 }, { 0, 128 * 128, 197, 118, 47, 116 }, { 16, 128 * 149, 204, 208, 100, 129 }, { 0, 128 * 128, 175, 179, 86, 111 } };
	const int *c = m[matrix & 3];
	const uint16_t *r = yu64 + (size_t)blockIdx.z * in_frame_words + (size_t)row * in_pitch_words + 4 * (size_t)p;
	uint8_t *o = out + (size_t)blockIdx.z * out_frame_bytes + (size_t)(rows - 1 - row) * out_pitch + 6 * (size_t)p;
	const int V = (int)r[1] - 32768, U = (int)r[3] - 32768;
	const uint32_t dz = dither_word(seed + 0x9E3779B9u * (uint32_t)(blockIdx.z + 1), row, p);
#pragma unroll
	for (int k = 0; k < 2; k++) {
		const int d = (int)((dz >> (16 * k)) & 0x7fffu);
		const int Y1 = (((int)r[2 * k] - (c[0] << 8)) * c[1]) >> 7;
		const int R = (Y1 + c[2] * V + d) >> 15, G = (Y1 - c[4] * (U >> 1) - c[3] * (V >> 1) + d) >> 15, B = (Y1 + 2 * c[5] * U + d) >> 15;
		o[3 * k + 0] = (uint8_t)(B < 0 ? 0 : (B > 255 ? 255 : B));
		o[3 * k + 1] = (uint8_t)(G < 0 ? 0 : (G > 255 ? 255 : G));
		o[3 * k + 2] = (uint8_t)(R < 0 ? 0 : (R > 255 ? 255 : R));
	}
}

// Half resolution of 4:2:2 samples as YU64 (decoder.c:23066 -> frame.c:11146 ConvertLowpass16sToYUV64, its 10-bit branch): the level-1 lowpass planes (12 bits for
// 10-bit samples, lowpass bias 4 as for every YU64 decode) clamped to [0, 4095], << 4, words Y0 C1 Y1 C2.  One thread per pixel pair.
__global__ void __launch_bounds__(NTHREADS) k_half_yu64(const HalfYuvJob *jobs)
{
	const HalfYuvJob &job = jobs[blockIdx.z];
	const int row = blockIdx.y, p = (int)(blockIdx.x * NTHREADS + threadIdx.x);
	if (2 * p >= job.width) return;
	auto word = [](int v) { return (uint32_t)(v < 0 ? 0 : (v > 4095 ? 4095 : v)) << 4; };
	const int16_t *y = job.ll[0] + (size_t)row * job.pitch[0] + 2 * p;
	uint2 px;
	px.x = word(y[0]) | (word(job.ll[1][(size_t)row * job.pitch[1] + p]) << 16);
	px.y = word(y[1]) | (word(job.ll[2][(size_t)row * job.pitch[2] + p]) << 16);
	*(uint2 *)(job.out + (size_t)row * job.out_pitch + 8 * (size_t)p) = px;
}

// Half resolution of 4:2:2 samples as RG24 (decoder.c:22924 -> frame.c:8504 ConvertLowpass16sToRGBNoIPPFast, whose vector code is compiled out -- MMXSUPPORTED is not defined --,
// so the scalar loop at :9153 does every column): the level-1 lowpass planes >> 4 to 8 bits (no clamp: color.h:49-51), Y = ((Y - y_offset) * ymult) >> 7,
// R = (Y + r_vmult V) >> 7, G = (2 Y - g_umult U - g_vmult V) >> 8, B = (Y + 2 b_umult U) >> 7, saturated to bytes B, G, R; no dither; bottom row first.
// One thread per pixel pair.
__global__ void __launch_bounds__(NTHREADS) k_half_rgb24(const HalfYuvJob *jobs)
{
	const HalfYuvJob &job = jobs[blockIdx.z];
	const int row = blockIdx.y, p = (int)(blockIdx.x * NTHREADS + threadIdx.x);
	if (2 * p >= job.width) return;
	const int m[4][6] = { { 16, 128 * 149, 230, 137, 55, 135 }, { 0, 128 * 128, 197, 118, 47, 116 }, { 16, 128 * 149, 204, 208, 100, 129 }, { 0, 128 * 128, 175, 179, 86, 111 } };
	const int *c = m[job.matrix & 3];
	const int16_t *y = job.ll[0] + (size_t)row * job.pitch[0] + 2 * p;
	if (job.mode == 1) {
		// BGRA / BGRa: samples packed to unsigned bytes, 16-bit vector arithmetic with six fraction bits and no rounding term (oracle_half_resolution_rgb32_of_yuv422)
		int vs = (int)job.ll[1][(size_t)row * job.pitch[1] + p] >> 4, us = (int)job.ll[2][(size_t)row * job.pitch[2] + p] >> 4;
		vs = (vs < 0 ? 0 : (vs > 255 ? 255 : vs)) - 128; us = (us < 0 ? 0 : (us > 255 ? 255 : us)) - 128;
		const int tr = (int)(int16_t)(vs * c[2]) >> 1, tgv = (int)(int16_t)(vs * c[3]) >> 2, tgu = (int)(int16_t)(us * c[4]) >> 2, tb = (int)(int16_t)(us * c[5]);
		uint32_t *o32 = (uint32_t *)(job.out + (size_t)(job.bottom_up ? job.rows - 1 - row : row) * job.out_pitch) + 2 * (size_t)p;
#pragma unroll
		for (int k = 0; k < 2; k++) {
			int ys = (int)y[k] >> 4;
			ys = (ys < 0 ? 0 : (ys > 255 ? 255 : ys)) - c[0];
			const int yy = (int)(int16_t)((((int)(int16_t)(ys << 7) * c[1]) >> 16) << 1);
			int R = adds16(yy, tr) >> 6, G = subs16(subs16(yy, tgv), tgu) >> 6, B = adds16(yy, tb) >> 6;
			R = R < 0 ? 0 : (R > 255 ? 255 : R); G = G < 0 ? 0 : (G > 255 ? 255 : G); B = B < 0 ? 0 : (B > 255 ? 255 : B);
			o32[k] = (uint32_t)B | ((uint32_t)G << 8) | ((uint32_t)R << 16) | 0xff000000u;
		}
		return;
	}
	if (job.mode >= 2) {
		// RG48 / b64a: the planes read as unsigned words << 4, the scalar arithmetic of frame.c:9567 in 32 bits (oracle_half_resolution_rgb16_of_yuv422)
		const int V = (int)(((uint32_t)(uint16_t)job.ll[1][(size_t)row * job.pitch[1] + p]) << 4) - 32768, U = (int)(((uint32_t)(uint16_t)job.ll[2][(size_t)row * job.pitch[2] + p]) << 4) - 32768;
		const int nw = job.mode == 3 ? 4 : 3;
		uint16_t *o16 = (uint16_t *)(job.out + (size_t)row * job.out_pitch) + (size_t)(2 * p) * nw;
#pragma unroll
		for (int k = 0; k < 2; k++) {
			const int Y = (int)((uint32_t)((int)(((uint32_t)(uint16_t)y[k]) << 4) - (c[0] << 8)) * (uint32_t)c[1]) >> 7;
			int R = (int)((uint32_t)Y + (uint32_t)(c[2] * V)) >> 7, G = (int)((uint32_t)(2 * Y) - (uint32_t)(c[4] * U) - (uint32_t)(c[3] * V)) >> 8, B = (int)((uint32_t)Y + (uint32_t)(2 * c[5] * U)) >> 7;
			R = R < 0 ? 0 : (R > 65535 ? 65535 : R); G = G < 0 ? 0 : (G > 65535 ? 65535 : G); B = B < 0 ? 0 : (B > 65535 ? 65535 : B);
			uint16_t *q = o16 + k * nw;
			if (nw == 4) { q[0] = 0xffff; q[1] = (uint16_t)R; q[2] = (uint16_t)G; q[3] = (uint16_t)B; } else { q[0] = (uint16_t)R; q[1] = (uint16_t)G; q[2] = (uint16_t)B; }
		}
		return;
	}
	const int V = ((int)job.ll[1][(size_t)row * job.pitch[1] + p] >> 4) - 128, U = ((int)job.ll[2][(size_t)row * job.pitch[2] + p] >> 4) - 128;
	uint8_t *o = job.out + (size_t)(job.rows - 1 - row) * job.out_pitch + 6 * (size_t)p;
#pragma unroll
	for (int k = 0; k < 2; k++) {
		const int Y = ((((int)y[k] >> 4) - c[0]) * c[1]) >> 7;
		const int R = (Y + c[2] * V) >> 7, G = (2 * Y - c[4] * U - c[3] * V) >> 8, B = (Y + 2 * c[5] * U) >> 7;
		o[3 * k + 0] = (uint8_t)(B < 0 ? 0 : (B > 255 ? 255 : B));
		o[3 * k + 1] = (uint8_t)(G < 0 ? 0 : (G > 255 ? 255 : G));
		o[3 * k + 2] = (uint8_t)(R < 0 ? 0 : (R > 255 ? 255 : R));
	}
}

// Half resolution of RGB 4:4:4 samples for the 8-bit, 10-bit and b64a outputs (decoder.c:26752 CopyLowpassRGB444ToBuffer -> frame.c:7150 ConvertLowpassRGB444ToRGB):
// the level-1 lowpass planes G, R, B (14 bits for 12-bit samples; they carry the lowpass bias of the output format, decoder.c:12290-12312: 8 for the 8-bit,
// 6 for the 10-bit formats), then
//   8 bit (frame.c:7226 / :7241 -> convert.c:6151 ConvertPlanarRGB16uToPackedRGB32 with shift 6): (v + 9 + r) clamped to 14 bits, >> 6, r = rand() & 31 shared by
//          the three components of a pixel (here: a counter-based hash); RG24 / BGRA bottom row first;
//   10 bit (frame.c:7662 ConvertLowpassRGB444ToRGB30): (v << 2) saturated to 16 bits, >> 6, at the format's bit positions;
//   b64a (frame.c:7494 ConvertLowpassRGB444ToB64A): (v << 2) saturated, alpha 65535.
// Pinned on the reference decoder (tests/test_oracle_vs_ref.py).  One thread per pixel: a quarter of the frame, nowhere near a bench line.
__global__ void __launch_bounds__(NTHREADS) k_half_rgb(const HalfPackedJob *jobs, uint32_t launch_seed)
{
	const HalfPackedJob &job = jobs[blockIdx.z];
	const int row = blockIdx.y, x = (int)(blockIdx.x * NTHREADS + threadIdx.x);
	if (x >= job.width) return;
	int v[3];
#pragma unroll
	for (int c = 0; c < 3; c++) v[c] = (int)job.ll[c][(size_t)row * job.pitch + x];      // G, R, B (the lowpass bias of the output format is in them: lowpass_bias())
	if (job.mode == 1) {
		const int r = (int)(dither_word(job.dither_seed ^ launch_seed, row, x) & 31u);
		const int yrow = job.bottom_up ? job.rows - 1 - row : row;
		uint8_t *o = (uint8_t *)job.out + (size_t)yrow * job.out_pitch + (size_t)x * job.bytes;
		const int order[3] = { 2, 0, 1 };                  // bytes B, G, R <- planes B, G, R
		const bool four = job.nch == 4;                    // BGRA / BGRa of an RGBA 4:4:4:4 sample: the planar-row route, no dither (as at full resolution: the 12-bit value >> 4)
#pragma unroll
		for (int k = 0; k < 3; k++) { int t = v[order[k]] + (four ? 0 : 9 + r); t = t < 0 ? 0 : (t > 0x3fff ? 0x3fff : t); o[k] = (uint8_t)(t >> 6); }
		if (four) {
			int a = (int)job.ll[3][(size_t)row * job.pitch + x];
			a = (a < 0 ? 0 : (a > 0x3fff ? 0x3fff : a)) >> 2;      // 12 bits, then the alpha expansion of codec.h:164-165
			a -= 256; a = a < 0 ? 0 : (((a << 3) * 9400) >> 16) >> 4;
			o[3] = (uint8_t)(a > 255 ? 255 : a);
		} else if (job.bytes == 4) o[3] = 255;
	} else if (job.mode == 2) {
		uint32_t w = 0;
#pragma unroll
		for (int c = 0; c < 3; c++) { int t = v[c] << 2; t = t < 0 ? 0 : (t > 65535 ? 65535 : t); w |= (uint32_t)(t >> 6) << job.word[c]; }
		if (job.big_endian) w = __builtin_bswap32(w);
		*(uint32_t *)((uint8_t *)job.out + (size_t)row * job.out_pitch + 4 * (size_t)x) = w;
	} else {
		uint32_t t[3];
#pragma unroll
		for (int c = 0; c < 3; c++) { const int q = v[c] << 2; t[c] = (uint32_t)(q < 0 ? 0 : (q > 65535 ? 65535 : q)); }
		uint2 px; px.x = 0xffffu | (t[1] << 16); px.y = t[0] | (t[2] << 16);      // words A, R, G, B
		*(uint2 *)((uint8_t *)job.out + (size_t)row * job.out_pitch + 8 * (size_t)x) = px;
	}
}

// BYR4 output of Bayer samples (DecodeBatch): the four component planes arrive as the 16-bit words R-G, G, B-G, G1-G2 of one photosite quad (the RG48 route of
// k_inv_packed16 with four planes = the reference's RawBayer16 rows) and leave as the quad's four samples -- r = ((rg - 32768) << 1) + g, b likewise,
// g1 = g + (gd - 32768), g2 = g - (gd - 32768), clamped to 16 bits, each through the linear-restore table [x >> 2] (bayer.c:13233 GenerateBYR2 with
// decoder.c:10714 BYR4LinearRestore; red-green phase: rows r g1 / g2 b).  One thread per quad: 8 bytes in, 2 x 4 bytes out, four gathers from a 32 KB table.
__global__ void __launch_bounds__(NTHREADS) k_bayer_to_byr4(const uint16_t *raw, int in_pitch_words, size_t in_frame_words, uint16_t *out, int out_pitch_words, size_t out_frame_words,
                                                            int quads, const uint16_t *curve)
{
	const int q = (int)(blockIdx.x * NTHREADS + threadIdx.x);
	if (q >= quads) return;
	const cfhd_u2 w = CFHD_LDG64(raw + (size_t)blockIdx.z * in_frame_words + (size_t)blockIdx.y * in_pitch_words + 4 * (size_t)q);
	const int rg = (int)(w.x & 0xffffu), g = (int)(w.x >> 16), bg = (int)(w.y & 0xffffu), gd = (int)(w.y >> 16) - 32768;
	int r = ((rg - 32768) << 1) + g, b = ((bg - 32768) << 1) + g, g1 = g + gd, g2 = g - gd;
	r = r < 0 ? 0 : (r > 0xffff ? 0xffff : r); b = b < 0 ? 0 : (b > 0xffff ? 0xffff : b);
	g1 = g1 < 0 ? 0 : (g1 > 0xffff ? 0xffff : g1); g2 = g2 < 0 ? 0 : (g2 > 0xffff ? 0xffff : g2);
	uint16_t *o = out + (size_t)blockIdx.z * out_frame_words + (size_t)(2 * blockIdx.y) * out_pitch_words + 2 * (size_t)q;
	*(uint32_t *)o = (uint32_t)curve[r >> 2] | ((uint32_t)curve[g1 >> 2] << 16);
	*(uint32_t *)(o + out_pitch_words) = (uint32_t)curve[g2 >> 2] | ((uint32_t)curve[b >> 2] << 16);
}

} // namespace dev
} // namespace cfhd
