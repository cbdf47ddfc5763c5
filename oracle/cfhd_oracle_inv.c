/* oracle/cfhd_oracle_inv.c -- TEST INFRASTRUCTURE ONLY (see cfhd_oracle.h).
 *
 * Inverse path: 2/6 biorthogonal synthesis (vertical then horizontal), the "descale" variant that
 * undoes the encoder's 2-bit prescale, and the last level fused with 8-bit 4:2:2 packing.
 */
#include "cfhd_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

static inline int sat16(int x) { return x < -32768 ? -32768 : (x > 32767 ? 32767 : x); }
static inline int adds(int a, int b) { return sat16(a + b); }
static inline int subs(int a, int b) { return sat16(a - b); }

/* Vertical synthesis of one coefficient row r into an even and an odd output row.
 * Codec/spatial.c:21877 InvertSpatialQuant16s: top border :21975-22030 (11,-4,1)/(5,4,-1),
 * middle SIMD order :22080-22150, bottom border :22330-22400 (5,4,-1)/(11,-4,1) mirrored.
 * low = vertical-lowpass band (LL or LH), high = vertical-highpass band (HL or HH). */
static void inv_vertical_row(const PIXEL16 *low, int low_pitch, const PIXEL16 *high_row, int r, int h, int w,
                             PIXEL16 *even_out, PIXEL16 *odd_out)
{
	int c;
	for (c = 0; c < w; c++) {
		int hi = high_row[c];
		int even, odd;
		if (r == 0) {
			int l0 = low[c], l1 = low[low_pitch + c], l2 = low[2 * low_pitch + c];
			even = (((11 * l0 - 4 * l1 + l2 + 4) >> 3) + hi) >> 1;
			odd  = (((5 * l0 + 4 * l1 - l2 + 4) >> 3) - hi) >> 1;
			even = sat16(even); odd = sat16(odd);
		} else if (r == h - 1) {
			int l0 = low[(size_t)r * low_pitch + c], l1 = low[(size_t)(r - 1) * low_pitch + c], l2 = low[(size_t)(r - 2) * low_pitch + c];
			even = (((5 * l0 + 4 * l1 - l2 + 4) >> 3) + hi) >> 1;
			odd  = (((11 * l0 - 4 * l1 + l2 + 4) >> 3) - hi) >> 1;
			even = sat16(even); odd = sat16(odd);
		} else {
			int a = low[(size_t)(r - 1) * low_pitch + c], b = low[(size_t)r * low_pitch + c], d = low[(size_t)(r + 1) * low_pitch + c];
			even = subs(a, d); even = adds(even, 4); even >>= 3; even = adds(even, b); even = adds(even, hi); even >>= 1;
			odd = subs(0, a); odd = adds(odd, d); odd = adds(odd, 4); odd >>= 3; odd = adds(odd, b); odd = subs(odd, hi); odd >>= 1;
		}
		even_out[c] = (PIXEL16)even;
		odd_out[c] = (PIXEL16)odd;
	}
}

/* Horizontal synthesis of one row: w lowpass + w highpass -> 2w outputs.
 * Codec/InvertHorizontalStrip16s.c:459 InvertHorizontalStrip16s (borders :172-198, :409-438, interior
 * :371-402) and :1700 InvertHorizontalStripDescale16s (no final >>1, result doubled with saturation:
 * _mm_adds_epi16(out,out) :1934-1935, scalar "<<= descaleshift" :2067-2068). */
static void inv_horizontal_row(const PIXEL16 *low, const PIXEL16 *high, int w, int descale, PIXEL16 *out)
{
	int c;
	for (c = 0; c < w; c++) {
		int even, odd, hi = high[c];
		if (c == 0) {
			even = ((11 * low[0] - 4 * low[1] + low[2] + 4) >> 3) + hi;
			odd  = ((5 * low[0] + 4 * low[1] - low[2] + 4) >> 3) - hi;
		} else if (c == w - 1) {
			even = ((5 * low[c] + 4 * low[c - 1] - low[c - 2] + 4) >> 3) + hi;
			odd  = ((11 * low[c] - 4 * low[c - 1] + low[c - 2] + 4) >> 3) - hi;
		} else {
			even = subs(low[c - 1], low[c + 1]); even = adds(even, 4); even >>= 3; even = adds(even, low[c]); even = adds(even, hi);
			odd  = subs(low[c + 1], low[c - 1]); odd = adds(odd, 4); odd >>= 3; odd = adds(odd, low[c]); odd = subs(odd, hi);
		}
		if (descale) { even = sat16(even * 2); odd = sat16(odd * 2); }
		else { even >>= 1; odd >>= 1; }
		out[2 * c] = (PIXEL16)sat16(even);
		out[2 * c + 1] = (PIXEL16)sat16(odd);
	}
}

/* Codec/wavelet.c:5685 TransformInverseSpatialQuantLowpass -> spatial.c:21877 / :22414. */
void orc_inv_spatial(PIXEL16 *const bands[4], int band_pitch, int w, int h, int descale, PIXEL16 *out, int out_pitch)
{
	PIXEL16 *el = (PIXEL16 *)malloc((size_t)w * 2), *ol = (PIXEL16 *)malloc((size_t)w * 2);
	PIXEL16 *eh = (PIXEL16 *)malloc((size_t)w * 2), *oh = (PIXEL16 *)malloc((size_t)w * 2);
	int r;
	for (r = 0; r < h; r++) {
		/* horizontal-lowpass rows from (LL, HL); horizontal-highpass rows from (LH, HH) */
		inv_vertical_row(bands[0], band_pitch, bands[2] + (size_t)r * band_pitch, r, h, w, el, ol);
		inv_vertical_row(bands[1], band_pitch, bands[3] + (size_t)r * band_pitch, r, h, w, eh, oh);
		inv_horizontal_row(el, eh, w, descale, out + (size_t)(2 * r) * out_pitch);
		inv_horizontal_row(ol, oh, w, descale, out + (size_t)(2 * r + 1) * out_pitch);
	}
	free(el); free(ol); free(eh); free(oh);
}

/* Codec/spatial.c:21114 InvertSpatialQuantOverflowProtected16s -- what the reference's GROUP decoder runs for the wavelets that were not prescaled: the top
 * wavelet of the temporal lowpass (level 4: wavelet.c:5759 `input->level >= 4`) and the wavelet of the temporal highpass (wavelet.c:5886).  Vertical pass in 32
 * bits with one saturation at the end (:21658-21700), the ordinary horizontal pass (InvertHorizontalStrip16s).  It carries a defect that a decoder which wants the
 * reference's pictures has to reproduce: the lowpass row pointer is not advanced behind the loop over the middle rows (the advance at :21770-21776 sits inside
 * `#if (0 && XMMOPT)`), so the border filter of the LAST coefficient row (:21799-21830) reads the lowpass rows h-2, h-3, h-4 where rows h-1, h-2, h-3 are meant.
 * The highpass-side bands (LH with HH) go through a ring of three row pointers that is right (:21762-21774).  Pinned on the reference decoder's group output. */
void orc_inv_spatial_overflow_protected(PIXEL16 *const bands[4], int band_pitch, int w, int h, PIXEL16 *out, int out_pitch)
{
	PIXEL16 *el = (PIXEL16 *)malloc((size_t)w * 2), *ol = (PIXEL16 *)malloc((size_t)w * 2);
	PIXEL16 *eh = (PIXEL16 *)malloc((size_t)w * 2), *oh = (PIXEL16 *)malloc((size_t)w * 2);
	int r, c, side;
	for (r = 0; r < h; r++) {
		for (side = 0; side < 2; side++) {
			const PIXEL16 *low = bands[side], *high = bands[2 + side] + (size_t)r * band_pitch;
			PIXEL16 *e = side ? eh : el, *o = side ? oh : ol;
			for (c = 0; c < w; c++) {
				int even, odd, hi = high[c];
				if (r == 0) {
					int l0 = low[c], l1 = low[band_pitch + c], l2 = low[2 * band_pitch + c];
					even = (((11 * l0 - 4 * l1 + l2 + 4) >> 3) + hi) >> 1;
					odd  = (((5 * l0 + 4 * l1 - l2 + 4) >> 3) - hi) >> 1;
				} else if (r == h - 1) {
					const int last = (side == 0 && h >= 4) ? r - 1 : r;      /* the defect: the LL band is read one row too high */
					int l0 = low[(size_t)last * band_pitch + c], l1 = low[(size_t)(last - 1) * band_pitch + c], l2 = low[(size_t)(last - 2) * band_pitch + c];
					even = (((5 * l0 + 4 * l1 - l2 + 4) >> 3) + hi) >> 1;
					odd  = (((11 * l0 - 4 * l1 + l2 + 4) >> 3) - hi) >> 1;
				} else {
					int a = low[(size_t)(r - 1) * band_pitch + c], b = low[(size_t)r * band_pitch + c], d = low[(size_t)(r + 1) * band_pitch + c];
					even = (((a - d + 4) >> 3) + b + hi) >> 1;
					odd  = (((d - a + 4) >> 3) + b - hi) >> 1;
				}
				e[c] = (PIXEL16)sat16(even); o[c] = (PIXEL16)sat16(odd);
			}
		}
		inv_horizontal_row(el, eh, w, 0, out + (size_t)(2 * r) * out_pitch);
		inv_horizontal_row(ol, oh, w, 0, out + (size_t)(2 * r + 1) * out_pitch);
	}
	free(el); free(ol); free(eh); free(oh);
}

/* One reconstructed 4:2:2 sample before the 10->8 bit reduction:
 * v = lowfilter +/- high (before the >>1), clamped at zero as the SIMD body does with the
 * +2048 / subs_epu16 pair (InvertHorizontalStrip16s.c:4086-4089); then (v>>1 + dither) >> shift,
 * clamped to 8 bits (packus :4620 / SATURATE_8U :4880).  dither is rand()&mask per SIMD lane in the
 * reference (:3869-3893); the oracle takes it as an explicit 0/1 input so both extremes can be checked. */
static int g_debug_raw = 0;            /* test hook: deliver ((v>>1) & 0xff) instead of the 8-bit pixel, see orc_debug_raw */
void orc_debug_raw(int mode) { g_debug_raw = mode; }
static inline int to8(int v, int shift, int dither)
{
	int x;
	if (g_debug_raw == 1) return (v >> 1) & 0xff;          /* low byte of the 10-bit value */
	if (g_debug_raw == 2) return ((v >> 1) >> 8) & 0xff;   /* high byte of the 10-bit value */
	if (v < 0) v = 0;
	x = ((v >> 1) + dither) >> shift;
	return x < 0 ? 0 : (x > 255 ? 255 : x);
}

static void inv_horizontal_row_prepack(const PIXEL16 *low, const PIXEL16 *high, int w, int *out /* 2w values, before >>1 */)
{
	int c;
	for (c = 0; c < w; c++) {
		int even, odd, hi = high[c];
		if (c == 0) {
			even = ((11 * low[0] - 4 * low[1] + low[2] + 4) >> 3) + hi;
			odd  = ((5 * low[0] + 4 * low[1] - low[2] + 4) >> 3) - hi;
		} else if (c == w - 1) {
			even = ((5 * low[c] + 4 * low[c - 1] - low[c - 2] + 4) >> 3) + hi;
			odd  = ((11 * low[c] - 4 * low[c - 1] + low[c - 2] + 4) >> 3) - hi;
		} else {
			even = subs(low[c - 1], low[c + 1]); even = adds(even, 4); even >>= 3; even = adds(even, low[c]); even += hi;
			odd  = subs(low[c + 1], low[c - 1]); odd = adds(odd, 4); odd >>= 3; odd = adds(odd, low[c]); odd -= hi;
		}
		out[2 * c] = even; out[2 * c + 1] = odd;
	}
}

/* Codec/decoder.c:27323 TransformInverseSpatialSectionToOutput -> spatial.c:31341/31511/31975
 * InvertSpatial{Top,Middle,Bottom}Row16sToOutput -> InvertHorizontalStrip16s.c:3770
 * InvertHorizontalStrip16sToYUYV (:5025 ToUYVY).  Channel order Y,V,U (:3785-3791). */
void orc_inv_spatial_to_yuv422(PIXEL16 *const bands[3][4], const int band_pitch[3], int luma_w, int h,
                               int precision, int uyvy, int dither, uint8_t *out, int out_pitch)
{
	int shift = precision - 8;
	int ch, r, k;
	PIXEL16 *el[3], *ol[3], *eh[3], *oh[3];
	int *even_px[3], *odd_px[3];
	for (ch = 0; ch < 3; ch++) {
		int w = ch ? luma_w / 2 : luma_w;
		el[ch] = (PIXEL16 *)malloc((size_t)w * 2); ol[ch] = (PIXEL16 *)malloc((size_t)w * 2);
		eh[ch] = (PIXEL16 *)malloc((size_t)w * 2); oh[ch] = (PIXEL16 *)malloc((size_t)w * 2);
		even_px[ch] = (int *)malloc((size_t)w * 2 * sizeof(int)); odd_px[ch] = (int *)malloc((size_t)w * 2 * sizeof(int));
	}
	for (r = 0; r < h; r++) {
		for (ch = 0; ch < 3; ch++) {
			int w = ch ? luma_w / 2 : luma_w;
			int bp = band_pitch[ch];
			inv_vertical_row(bands[ch][0], bp, bands[ch][2] + (size_t)r * bp, r, h, w, el[ch], ol[ch]);
			inv_vertical_row(bands[ch][1], bp, bands[ch][3] + (size_t)r * bp, r, h, w, eh[ch], oh[ch]);
			inv_horizontal_row_prepack(el[ch], eh[ch], w, even_px[ch]);
			inv_horizontal_row_prepack(ol[ch], oh[ch], w, odd_px[ch]);
		}
		for (k = 0; k < 2; k++) {
			uint8_t *o = out + (size_t)(2 * r + k) * out_pitch;
			int *const *px = k ? odd_px : even_px;
			int x;
			for (x = 0; x < luma_w; x++) {           /* luma_w = band width; 2*luma_w output luma samples */
				int y0 = to8(px[0][2 * x], shift, dither), y1 = to8(px[0][2 * x + 1], shift, dither);
				int u = to8(px[2][x], shift, dither), v = to8(px[1][x], shift, dither);
				if (uyvy) { o[4 * x] = (uint8_t)u; o[4 * x + 1] = (uint8_t)y0; o[4 * x + 2] = (uint8_t)v; o[4 * x + 3] = (uint8_t)y1; }
				else      { o[4 * x] = (uint8_t)y0; o[4 * x + 1] = (uint8_t)u; o[4 * x + 2] = (uint8_t)y1; o[4 * x + 3] = (uint8_t)v; }
			}
		}
	}
	for (ch = 0; ch < 3; ch++) { free(el[ch]); free(ol[ch]); free(eh[ch]); free(oh[ch]); free(even_px[ch]); free(odd_px[ch]); }
}

/* Interlaced ("frame" transform) last level -> packed 8-bit 4:2:2.
 * Codec/decoder.c:21493 TransformInverseFrameToYUV (:24304 threaded): per band row r of the level-1 wavelet, per channel,
 * spatial.c:19302 InvertHorizontalRow16s8sTo16sBuffered turns (LL, LH) into the temporal lowpass row and (HL, HH) into the temporal
 * highpass row (the same horizontal synthesis as everywhere, >> 1, saturated); HL arrives as running sums (the decoder undoes the
 * difference coding, decoder.c:20822).  temporal.c:5961 InvertInterlacedRow16s10bitToYUV (:6498 ToUYVY) then gives picture row 2r =
 * low - high and row 2r + 1 = low + high (saturating 16-bit), clamped to [0, 2047] by the adds / subs_epu16 pair (:6071-6078),
 * halved, + dither, >> 2, packed with unsigned saturation.  The columns behind the last group of 8 chroma columns take the scalar
 * loop (:6276-6390): (low -+ high) / 2 >> 2 without dither -- inside the same interval.  Channel order Y, V, U as everywhere. */
void orc_inv_frame_to_yuv422(PIXEL16 *const bands[3][4], const int band_pitch[3], int luma_w, int h,
                             int precision, int uyvy, int dither, uint8_t *out, int out_pitch)
{
	const int shift = precision - 8;
	int ch, r, k, x;
	PIXEL16 *low[3], *high[3];
	for (ch = 0; ch < 3; ch++) {
		const int w = ch ? luma_w / 2 : luma_w;
		low[ch] = (PIXEL16 *)malloc((size_t)w * 4); high[ch] = (PIXEL16 *)malloc((size_t)w * 4);
	}
	for (r = 0; r < h; r++) {
		for (ch = 0; ch < 3; ch++) {
			const int w = ch ? luma_w / 2 : luma_w;
			const size_t o = (size_t)r * band_pitch[ch];
			inv_horizontal_row(bands[ch][0] + o, bands[ch][1] + o, w, 0, low[ch]);
			inv_horizontal_row(bands[ch][2] + o, bands[ch][3] + o, w, 0, high[ch]);
		}
		for (k = 0; k < 2; k++) {
			uint8_t *o = out + (size_t)(2 * r + k) * out_pitch;
			for (x = 0; x < luma_w; x++) {           /* 2 * luma_w luma samples, luma_w samples of each chroma channel */
				int px[4];                               /* y0, y1, v, u */
				const int l[4] = { low[0][2 * x], low[0][2 * x + 1], low[1][x], low[2][x] };
				const int g[4] = { high[0][2 * x], high[0][2 * x + 1], high[1][x], high[2][x] };
				int i;
				for (i = 0; i < 4; i++) px[i] = to8(k ? adds(l[i], g[i]) : subs(l[i], g[i]), shift, dither);
				if (uyvy) { o[4 * x] = (uint8_t)px[3]; o[4 * x + 1] = (uint8_t)px[0]; o[4 * x + 2] = (uint8_t)px[2]; o[4 * x + 3] = (uint8_t)px[1]; }
				else      { o[4 * x] = (uint8_t)px[0]; o[4 * x + 1] = (uint8_t)px[3]; o[4 * x + 2] = (uint8_t)px[1]; o[4 * x + 3] = (uint8_t)px[2]; }
			}
		}
	}
	for (ch = 0; ch < 3; ch++) { free(low[ch]); free(high[ch]); }
}

/* 12-bit sample of an RGB 4:4:4 plane -> 16-bit output word.  v = lowfilter +/- high before the >>1.
 * InvertHorizontalStrip16s.c:16571 InvertHorizontalStrip16sToRow16u: interior columns clamp v to
 * [0, 2^(precision+1) - 1] with the adds_epi16 / subs_epu16 "protection" pair (:16596, :16724-16726), halve, and shift left by
 * 16 - precision (:16749); the border columns halve first and clamp through SATURATE_16U of the shifted value (:16650-16671).
 * Both give clamp(v >> 1, 0, 2^precision - 1) << (16 - precision) on the values a decoder meets. */
static inline unsigned to16(int v, int precision)
{
	int x = v >> 1, top = (1 << precision) - 1;
	if (x < 0) x = 0;
	if (x > top) x = top;
	return (unsigned)x << (16 - precision);
}

/* The columns behind the SIMD loops -- band columns w - w%8 - 9 .. w-1 for w >= 16: the scalar loop after the last vector group
 * and the right border (InvertHorizontalStrip16s.c:16876-16990) -- shift first and saturate the shifted value with SATURATE_16U,
 * so a clipped highlight reads 65535 there and 4095 << 4 = 65520 in the vector columns. */
static inline unsigned to16_tail(int v, int precision)
{
	int x = (v >> 1) << (16 - precision);
	return (unsigned)(x < 0 ? 0 : (x > 65535 ? 65535 : x));
}

/* Codec/decoder.c:26887 (RGB 4:4:4 sample, RG48 output, no active metadata) -> wavelet.c:4947 TransformInverseRGB444ToRGB48:
 * per band row and channel spatial.c:16985 InvertSpatial{Top,Middle,Bottom}Row16sToYUV16 (vertical synthesis in 32 bits with a
 * final SATURATE, :17183-17230) + InvertHorizontalStrip16sToRow16u, then convert.c:6747 ConvertPlanarRGB16uToPackedRGB48
 * (planes are G, R, B; output words R, G, B).
 * The word order, the first scalar column and the alpha plane are parameters.  b64a output of an RGBA 4:4:4:4 sample takes another
 * route in the reference (bayer.c:7147 forces its "active metadata" decoder for alpha output): decoder.c:26805
 * TransformInverseSpatialUniversalThreadedToRow16u -> the same InvertHorizontalStrip16sToRow16u per plane, then per row
 * bayer.c:11916 Row16uFull2OutputFormat -> convert.c:6031 ConvertPlanarGRBAToPlanarRGBA -> bayer.c:15966 Convert4444LinesToOutput,
 * which expands the companded alpha (codec.h:164-165, scalar loop bayer.c:16212-16226) and interleaves A, R, G, B (:17098-17106).
 * Both instances are pinned against the reference decoder (tests/test_oracle_vs_ref.py).
 * word_of_channel[c] = position of plane c's word in the pixel, tail_start = first band column of the scalar code,
 * alpha_channel = plane to expand (-1: none). */
void orc_inv_spatial_to_packed16(PIXEL16 *const bands[4][4], int band_pitch, int w, int h, int precision, int num_channels,
                                 const int *word_of_channel, int tail_start, int alpha_channel, uint16_t *out, int out_pitch_words)
{
	int ch, r, k, x;
	PIXEL16 *el = (PIXEL16 *)malloc((size_t)w * 2), *ol = (PIXEL16 *)malloc((size_t)w * 2);
	PIXEL16 *eh = (PIXEL16 *)malloc((size_t)w * 2), *oh = (PIXEL16 *)malloc((size_t)w * 2);
	int *px[2]; px[0] = (int *)malloc((size_t)w * 2 * sizeof(int)); px[1] = (int *)malloc((size_t)w * 2 * sizeof(int));
	const int top = (1 << precision) - 1;
	for (r = 0; r < h; r++)
		for (ch = 0; ch < num_channels; ch++) {
			inv_vertical_row(bands[ch][0], band_pitch, bands[ch][2] + (size_t)r * band_pitch, r, h, w, el, ol);
			inv_vertical_row(bands[ch][1], band_pitch, bands[ch][3] + (size_t)r * band_pitch, r, h, w, eh, oh);
			inv_horizontal_row_prepack(el, eh, w, px[0]);
			inv_horizontal_row_prepack(ol, oh, w, px[1]);
			for (k = 0; k < 2; k++) {
				uint16_t *o = out + (size_t)(2 * r + k) * out_pitch_words + word_of_channel[ch];
				for (x = 0; x < 2 * w; x++) {
					const int tail = (x >> 1) >= tail_start, v = px[k][x];
					unsigned word;
					word = tail ? to16_tail(v, precision) : to16(v, precision);
					if (ch == alpha_channel) {
						/* bayer.c:16212-16226 (16-bit planar rows; the vector loop above it never runs because of its
						 * `(width*3) & ~15` guard): undo the encoder's alpha companding on the finished 16-bit word */
						int a = (int)(word >> 4);
						a -= 256; a <<= 3; a *= 9400; a >>= 12;
						word = (unsigned)(a < 0 ? 0 : (a > 65535 ? 65535 : a));
					}
					o[(size_t)x * num_channels] = (uint16_t)word;
				}
			}
		}
	free(el); free(ol); free(eh); free(oh); free(px[0]); free(px[1]);
}

void orc_inv_spatial_to_b64a(PIXEL16 *const bands[4][4], int band_pitch, int w, int h, int precision, uint16_t *out, int out_pitch_words)
{
	static const int word_of_channel[4] = { 2, 1, 3, 0 };      /* planes G, R, B, A -> words A, R, G, B */
	orc_inv_spatial_to_packed16(bands, band_pitch, w, h, precision, 4, word_of_channel, w - (w % 8) - 9, 3, out, out_pitch_words);
}

void orc_inv_spatial_to_rgb48(PIXEL16 *const bands[4][4], int band_pitch, int w, int h, int precision, int num_channels,
                              uint16_t *out, int out_pitch_words)
{
	static const int word_of_channel[4] = { 1, 0, 2, 3 };      /* plane G -> word 1, R -> 0, B -> 2 */
	orc_inv_spatial_to_packed16(bands, band_pitch, w, h, precision, num_channels, word_of_channel, w - (w % 8) - 9, -1, out, out_pitch_words);
}

/* ---- 4:2:2 samples decoded to YU64 (16-bit words Y0 C1 Y1 C2) -------------------------------------------------------------------
 * The reference library does NOT take the row-pair route its source suggests at first sight (decoder.c:14446 -> wavelet.c:5403
 * TransformInverseSpatialToV210, whose "10 bit limit" horizontal pass clamps every column but the first to 1023): probing the built
 * library shows the signature of the planar 16-bit row route instead -- the one RG48 / b64a output takes (decoder.c:26500-26545
 * TransformInverseSpatialUniversalThreadedToRow16u + ConvertRow16uToOutput): per plane InvertHorizontalStrip16sToRow16u, i.e. clamp to
 * `precision` bits and shift up in the vector columns, shift first and saturate to 65535 in the columns of its scalar loop (band columns
 * >= w - w % 8 - 9, per plane: the chroma planes are half as wide), then the planes interleaved as Y | channel 1, Y | channel 2
 * (convert.c:14139-14162).  Pinned against the reference decoder in tests/test_oracle_vs_ref.py, highlights included. */
void orc_inv_spatial_to_yu64(PIXEL16 *const bands[3][4], const int band_pitch[3], int luma_w, int h, int precision, uint16_t *out, int out_pitch_words)
{
	int ch, r, k, x;
	PIXEL16 *el = (PIXEL16 *)malloc((size_t)luma_w * 2), *ol = (PIXEL16 *)malloc((size_t)luma_w * 2);
	PIXEL16 *eh = (PIXEL16 *)malloc((size_t)luma_w * 2), *oh = (PIXEL16 *)malloc((size_t)luma_w * 2);
	int *px[2]; px[0] = (int *)malloc((size_t)luma_w * 2 * sizeof(int)); px[1] = (int *)malloc((size_t)luma_w * 2 * sizeof(int));
	for (r = 0; r < h; r++)
		for (ch = 0; ch < 3; ch++) {
			const int w = ch ? luma_w / 2 : luma_w, bp = band_pitch[ch], tail_start = w - (w % 8) - 9;
			inv_vertical_row(bands[ch][0], bp, bands[ch][2] + (size_t)r * bp, r, h, w, el, ol);
			inv_vertical_row(bands[ch][1], bp, bands[ch][3] + (size_t)r * bp, r, h, w, eh, oh);
			inv_horizontal_row_prepack(el, eh, w, px[0]);
			inv_horizontal_row_prepack(ol, oh, w, px[1]);
			for (k = 0; k < 2; k++) {
				uint16_t *o = out + (size_t)(2 * r + k) * out_pitch_words;
				for (x = 0; x < 2 * w; x++) {
					const int tail = (x >> 1) >= tail_start, v = px[k][x];
					const unsigned word = tail ? to16_tail(v, precision) : to16(v, precision);
					if (ch == 0) o[2 * x] = (uint16_t)word;                 /* luma sample x -> word 2x */
					else o[4 * x + (ch == 1 ? 1 : 3)] = (uint16_t)word;     /* chroma sample x sits in pixel pair x: channel 1 behind Y0, channel 2 behind Y1 */
				}
			}
		}
	free(el); free(ol); free(eh); free(oh); free(px[0]); free(px[1]);
}

/* ---- RGB 4:4:4 samples decoded to the 8-bit RGB formats (RG24, BGRA: bottom row first; BGRa: top row first) ------------------------
 * Codec/wavelet.c:4700-4950 TransformInverseRGB444ToRGB32: the planes as 16-bit rows (InvertSpatial*Row16sToYUV16: orc_inv_spatial_to_rgb48), packed two rows at
 * a time by Codec/convert.c:6151 ConvertPlanarRGB16uToPackedRGB32 / :6475 ...RGB24 with shift 8: every word takes an unsigned saturating add of
 * (rand() & 127) + 10 * 127 / 32 -- the same eight-lane rounding vector for the three components -- and loses its low byte.  The pyramid this runs on carries the
 * lowpass bias of these output formats, 8 (Codec/decoder.c:12290-12296): the caller adds it.  The reference draws with rand(); the oracle takes r in 0..127 as an
 * input so that both ends of the interval can be computed (pinned: every byte of the reference inside, both ends reached).  Bytes B, G, R (, A = 255). */
void orc_inv_spatial_to_rgb8(PIXEL16 *const bands[4][4], int band_pitch, int w, int h, int precision, int display_height, int bytes_per_pixel, int bottom_up,
                             int r, uint8_t *out, int out_pitch_bytes)
{
	const int W = 2 * w;
	uint16_t *tmp = (uint16_t *)malloc((size_t)2 * h * W * 3 * sizeof(uint16_t));
	int y, x, c;
	orc_inv_spatial_to_rgb48(bands, band_pitch, w, h, precision, 3, tmp, W * 3);
	for (y = 0; y < display_height; y++) {
		uint8_t *o = out + (size_t)(bottom_up ? display_height - 1 - y : y) * out_pitch_bytes;
		for (x = 0; x < W; x++) {
			for (c = 0; c < 3; c++) {                      /* RG48 words R, G, B -> bytes B, G, R */
				const int a = tmp[((size_t)y * W + x) * 3 + c];
				int v = a + 39 + r;                          /* _mm_adds_epu16(word, (rand() & 127) + 10 * 127 / 32) */
				v = (v > 65535 ? 65535 : v) >> 8;
				o[(size_t)x * bytes_per_pixel + (2 - c)] = (uint8_t)v;
			}
			if (bytes_per_pixel == 4) o[(size_t)x * 4 + 3] = 255;
		}
	}
	free(tmp);
}

/* ---- 4:2:2 samples decoded to RG48 / b64a (TestCFHD's RG48 -> 4:2:2 and b64a -> 4:2:2 rows at full resolution) -----------------------------------------
 * Traced on the instrumented reference (tools/trace_reference.md): Codec/decoder.c ReconstructSampleFrameYUV422ToBuffer -> TransformInverseSpatialUniversalThreadedToRow16u
 * (the planes as 16-bit rows, InvertHorizontalStrip16sToRow16u: the YU64 route, orc_inv_spatial_to_yu64, on a pyramid with the DEFAULT lowpass bias 24 / 5 --
 * decoder.c:12268-12276 names only YU64 / YR16 / V210 for the small one) -> Codec/bayer.c:11916 Row16uFull2OutputFormat, ENCODED_FORMAT_YUV_422 without active
 * metadata (:12168-12176): RGB2YUV.c:1308 ChannelYUYV16toPlanarYUV16 (every chroma word serves two pixels; plane 1 is V, plane 2 is U) -> RGB2YUV.c:1760
 * PlanarYUV16toPlanarRGB16 (15-bit samples, 13-bit coefficients with the "tweak" offsets of :55-58, _mm_mulhi_epi16 products, saturating sums, << 2, a clamp to
 * [0, 16383] by the 0x7fff - 0x3fff add / unsigned subtract pair, << 2) -> bayer.c:478 ConvertLinesToOutput with white point 16: the words as they are, R G B (RG48,
 * :1358-1373) or 0xffff R G B (b64a, :1090-1106).  Widths are multiples of 16 here, so the vector body serves every column.  color_space: 1 = 601, 2 = 709
 * (computer-systems range: what a sample without other colour tags says). */
static int mulhi16(int a, int b) { return (a * b) >> 16; }
static int wrap16(int x) { return (int)(int16_t)(uint16_t)x; }
void orc_yu64_to_rgb16(const uint16_t *yu, int yu_pitch_words, int width, int rows, int color_space, int b64a, uint16_t *out, int out_pitch_words)
{
	/* PlanarYUV16toPlanarRGB16: fprecision = 8192; (int)(8192 * 1.164f) ... + tweakYUV2RGB_CG601 / _CG709 */
	const int is601 = color_space == 1;
	const int y_offset = 2048 + (is601 ? -28 : -32), ymult = 9535 + (is601 ? 14 : 11);
	const int r_vmult = (is601 ? 13074 : 14688) + 6, g_vmult = (is601 ? 6660 + 1 : 4374 - 17), g_umult = (is601 ? 3203 + 7 : 1744 - 6), b_umult = (is601 ? 16531 + 3 : 17326 + 0);
	const int u_offset = (1 << 14) + (is601 ? 23 : 22), v_offset = (1 << 14) + (is601 ? 23 : 22);
	int y, x, c;
	for (y = 0; y < rows; y++) {
		const uint16_t *row = yu + (size_t)y * yu_pitch_words;
		uint16_t *o = out + (size_t)y * out_pitch_words;
		for (x = 0; x < width; x++) {
			const int Y = row[2 * x], V = row[4 * (x >> 1) + 1], U = row[4 * (x >> 1) + 3];      /* words Y0 C1 Y1 C2: channel 1 = V, channel 2 = U */
			int yy = sat16((Y >> 1) - y_offset), uu = sat16((U >> 1) - u_offset), vv = sat16((V >> 1) - v_offset);
			int comp[3];
			yy = mulhi16(yy, ymult);
			comp[0] = sat16(mulhi16(vv, r_vmult) + yy);
			comp[1] = sat16(sat16(yy + mulhi16(uu, -g_umult)) + mulhi16(vv, -g_vmult));
			comp[2] = sat16(mulhi16(uu, b_umult) + yy);
			for (c = 0; c < 3; c++) {
				int v = wrap16(comp[c] << 2);                                /* 12 -> 14 bits (a 16-bit shift) */
				v = sat16(v + (0x7fff - 0x3fff));                            /* adds_epi16 */
				v = (v & 0xffff) - (0x7fff - 0x3fff); if (v < 0) v = 0;      /* subs_epu16 */
				comp[c] = (v << 2) & 0xffff;                                 /* 14 -> 16 bits */
			}
			if (b64a) { o[4 * x] = 0xffff; o[4 * x + 1] = (uint16_t)comp[0]; o[4 * x + 2] = (uint16_t)comp[1]; o[4 * x + 3] = (uint16_t)comp[2]; }
			else { o[3 * x] = (uint16_t)comp[0]; o[3 * x + 1] = (uint16_t)comp[1]; o[3 * x + 2] = (uint16_t)comp[2]; }
		}
	}
}

void orc_inv_spatial_to_rgb16_of_yuv422(PIXEL16 *const bands[3][4], const int band_pitch[3], int luma_w, int h, int precision, int display_height, int color_space, int b64a,
                                        uint16_t *out, int out_pitch_words)
{
	const int W = 2 * luma_w;
	uint16_t *yu = (uint16_t *)malloc((size_t)2 * h * W * 2 * sizeof(uint16_t));
	orc_inv_spatial_to_yu64(bands, band_pitch, luma_w, h, precision, yu, W * 2);
	orc_yu64_to_rgb16(yu, W * 2, W, display_height, color_space, b64a, out, out_pitch_words);
	free(yu);
}

/* ---- 4:2:2 samples decoded to BGRA (bottom row first) / BGRa (top row first) (TestCFHD's BGRA -> 4:2:2 and BGRa -> 4:2:2 rows at full resolution) ----------------
 * Traced: ReconstructSampleFrameYUV422ToBuffer -> TransformInverseSpatialThreadedYUV422ToBuffer -> InvertSpatial{Top,Middle,Bottom}Row16sToOutput (the vertical
 * pass of every other last level) -> Codec/spatial.c:29577 InvertHorizontalStripYUV16sToPackedRGB32: the horizontal pass of the 8-bit 4:2:2 route WITHOUT dither (its
 * rand() block is compiled out, :29737 `#if 0`) fused with an 8-bit colour conversion whose coefficients come from dither.c:264 ComputeColorCoefficientsYUVToRGB.
 *   vector columns (band columns below post_column, :29640-29644): samples clamped to [0, 255] (luma after its offset), Y << 7 mulhi 128 * 149 << 1, chroma
 *     products in wrapping 16-bit arithmetic shifted to six fraction bits, + 32 >> 6, packus (:30440-30520);
 *   scalar columns (the rest, :30839-31180): no clamp of the samples, Y * ymult >> 7, seven / eight fraction bits, SATURATE_8U.
 * Every chroma sample serves two pixels.  Bytes B, G, R, 255.  The pyramid carries the bias of decoder.c:12268 / :12479 (24; odd lowpass widths: 5, and for the
 * bottom-up format -3 / +1 / +1 by channel, :12500-12508): the caller applies it. */
void orc_inv_spatial_to_rgb32_of_yuv422(PIXEL16 *const bands[3][4], const int band_pitch[3], int luma_w, int h, int precision, int display_height, int color_space,
                                        int bottom_up, uint8_t *out, int out_pitch_bytes)
{
	const int is601 = color_space == 1, shift = precision - 8;
	const int ymult = 128 * 149, r_vmult = is601 ? 204 : 230, g_vmult = is601 ? 208 : 137, g_umult = is601 ? 100 : 55, b_umult = is601 ? 129 : 135, luma_offset = 16;
	int post_column = luma_w - (luma_w % 16), ch, r, k, x;
	PIXEL16 *el = (PIXEL16 *)malloc((size_t)luma_w * 2), *ol = (PIXEL16 *)malloc((size_t)luma_w * 2);
	PIXEL16 *eh = (PIXEL16 *)malloc((size_t)luma_w * 2), *oh = (PIXEL16 *)malloc((size_t)luma_w * 2);
	int *px[3][2];
	while (post_column > luma_w - 2 - 2) post_column -= 16;
	for (ch = 0; ch < 3; ch++) for (k = 0; k < 2; k++) px[ch][k] = (int *)malloc((size_t)luma_w * 2 * sizeof(int));
	for (r = 0; r < h; r++) {
		for (ch = 0; ch < 3; ch++) {
			const int w = ch ? luma_w / 2 : luma_w, bp = band_pitch[ch];
			inv_vertical_row(bands[ch][0], bp, bands[ch][2] + (size_t)r * bp, r, h, w, el, ol);
			inv_vertical_row(bands[ch][1], bp, bands[ch][3] + (size_t)r * bp, r, h, w, eh, oh);
			inv_horizontal_row_prepack(el, eh, w, px[ch][0]);
			inv_horizontal_row_prepack(ol, oh, w, px[ch][1]);
		}
		for (k = 0; k < 2; k++) {
			const int row = 2 * r + k;
			uint8_t *o;
			if (row >= display_height) continue;
			o = out + (size_t)(bottom_up ? display_height - 1 - row : row) * out_pitch_bytes;
			for (x = 0; x < 2 * luma_w; x++) {
				/* samples before the colour conversion: (value before the last >> 1) >> 1 >> shift; channel 1 = V, channel 2 = U; chroma sample x / 2 */
				const int ys = (px[0][k][x] >> 1) >> shift, vs = (px[1][k][x >> 1] >> 1) >> shift, us = (px[2][k][x >> 1] >> 1) >> shift;
				int rr, gg, bb;
				if ((x >> 1) < post_column) {
					int yy = ys - luma_offset, uu = us, vv = vs, t;
					yy = yy < 0 ? 0 : (yy > 255 ? 255 : yy); uu = (uu < 0 ? 0 : (uu > 255 ? 255 : uu)) - 128; vv = (vv < 0 ? 0 : (vv > 255 ? 255 : vv)) - 128;
					yy = wrap16(mulhi16(wrap16(yy << 7), ymult) << 1);
					t = wrap16(vv * r_vmult) >> 1; rr = sat16(sat16(yy + t) + 32) >> 6;
					t = wrap16(vv * g_vmult) >> 2; gg = sat16(yy - t); t = wrap16(uu * g_umult) >> 2; gg = sat16(gg - t); gg = sat16(gg + 32) >> 6;
					t = wrap16(uu * b_umult); bb = sat16(sat16(yy + t) + 32) >> 6;
				} else {
					const int yy = ((ys - luma_offset) * ymult) >> 7, uu = us - 128, vv = vs - 128;
					rr = (yy + r_vmult * vv + 64) >> 7;
					gg = (yy * 2 - g_umult * uu - g_vmult * vv + 128) >> 8;
					bb = (yy + 2 * b_umult * uu + 64) >> 7;
				}
				o[4 * x] = (uint8_t)(bb < 0 ? 0 : (bb > 255 ? 255 : bb)); o[4 * x + 1] = (uint8_t)(gg < 0 ? 0 : (gg > 255 ? 255 : gg));
				o[4 * x + 2] = (uint8_t)(rr < 0 ? 0 : (rr > 255 ? 255 : rr)); o[4 * x + 3] = 255;
			}
		}
	}
	for (ch = 0; ch < 3; ch++) for (k = 0; k < 2; k++) free(px[ch][k]);
	free(el); free(ol); free(eh); free(oh);
}

/* One plane's last-level reconstruction BEFORE the final >> 1 (v = lowfilter +/- high): the value the reference's output routines start
 * from; probes of further output formats are fitted on it (tests only). */
void orc_inv_spatial_prepack(PIXEL16 *const bands[4], int band_pitch, int w, int h, int32_t *out, int out_pitch)
{
	PIXEL16 *el = (PIXEL16 *)malloc((size_t)w * 2), *ol = (PIXEL16 *)malloc((size_t)w * 2);
	PIXEL16 *eh = (PIXEL16 *)malloc((size_t)w * 2), *oh = (PIXEL16 *)malloc((size_t)w * 2);
	int r;
	for (r = 0; r < h; r++) {
		inv_vertical_row(bands[0], band_pitch, bands[2] + (size_t)r * band_pitch, r, h, w, el, ol);
		inv_vertical_row(bands[1], band_pitch, bands[3] + (size_t)r * band_pitch, r, h, w, eh, oh);
		inv_horizontal_row_prepack(el, eh, w, out + (size_t)(2 * r) * out_pitch);
		inv_horizontal_row_prepack(ol, oh, w, out + (size_t)(2 * r + 1) * out_pitch);
	}
	free(el); free(ol); free(eh); free(oh);
}

/* ---- RGB 4:4:4 samples decoded to the 10-bit RGB words r210 / DPX0 (big-endian) / AB10 / AR10 (little-endian) -------------------
 * First fitted by probing (round 2: "(v + 3) >> 3"), now explained: the reference adds a lowpass bias of 6 for these output formats (Codec/decoder.c:12304-12310),
 * which reaches the last-level reconstruction before its final >> 1 as + 3 (an even bias passes the descaling levels unchanged), and its output stage
 * truncates that 13-bit value to 10 bits.  So: the caller's pyramid carries the bias, every component is v >> 3 clamped to [0, 1023], at the bit positions the
 * encoder reads them from (r210: R 20-29, G 10-19, B 0-9; DPX0: 22 / 12 / 2; AB10: R 0-9, G 10-19, B 20-29; AR10: R 20-29, G 10-19, B 0-9).  Pinned word for word
 * in tests/test_oracle_vs_ref.py.  Planes are G, R, B.  shift_r/g/b: bit positions; big_endian: words stored byte-swapped. */
void orc_inv_spatial_to_rgb10(PIXEL16 *const bands[4][4], int band_pitch, int w, int h, int display_height, int shift_r, int shift_g, int shift_b, int big_endian,
                              uint32_t *out, int out_pitch_words)
{
	const int W = 2 * w, shifts[3] = { shift_g, shift_r, shift_b };
	int32_t *v = (int32_t *)malloc((size_t)2 * h * W * sizeof(int32_t));
	int c, y, x;
	for (y = 0; y < display_height; y++) for (x = 0; x < W; x++) out[(size_t)y * out_pitch_words + x] = 0;
	for (c = 0; c < 3; c++) {
		orc_inv_spatial_prepack(bands[c], band_pitch, w, h, v, W);
		for (y = 0; y < display_height; y++)
			for (x = 0; x < W; x++) {
				int s = v[(size_t)y * W + x] >> 3;          /* 13 bits -> 10; rounding comes from the lowpass bias (6) the caller's pyramid carries */
				s = s < 0 ? 0 : (s > 1023 ? 1023 : s);
				out[(size_t)y * out_pitch_words + x] |= (uint32_t)s << shifts[c];
			}
	}
	if (big_endian)
		for (y = 0; y < display_height; y++) for (x = 0; x < W; x++) {
			const uint32_t u = out[(size_t)y * out_pitch_words + x];
			out[(size_t)y * out_pitch_words + x] = (u >> 24) | ((u >> 8) & 0xff00u) | ((u << 8) & 0xff0000u) | (u << 24);
		}
	free(v);
}

/* ---- 4:2:2 samples decoded to v210 (three 10-bit samples per 32-bit word; groups of six pixels in four words: Cb0 Y0 Cr0 | Y1 Cb1 Y2 |
 * Cr1 Y3 Cb2 | Y4 Cr2 Y5, low bits first) ------------------------------------------------------------------------------------------------
 * Probed on the built reference (tests/test_oracle_vs_ref.py): the 10-bit samples are the YU64 words >> 6 -- the same planar 16-bit row
 * route --, Cb = channel 2, Cr = channel 1.  Whole groups of six pixels only (the pin covers widths that are multiples of 6). */
void orc_inv_spatial_to_v210(PIXEL16 *const bands[3][4], const int band_pitch[3], int luma_w, int h, int precision, uint32_t *out, int out_pitch_words)
{
	const int W = 2 * luma_w;
	uint16_t *yu = (uint16_t *)malloc((size_t)2 * h * W * 2 * sizeof(uint16_t));
	int y, g;
	orc_inv_spatial_to_yu64(bands, band_pitch, luma_w, h, precision, yu, W * 2);
	for (y = 0; y < 2 * h; y++) {
		const uint16_t *r = yu + (size_t)y * W * 2;           /* words Y0 C1 Y1 C2 per pixel pair */
		uint32_t *o = out + (size_t)y * out_pitch_words;
		for (g = 0; g + 6 <= W; g += 6) {
			uint32_t Y[6], Cb[3], Cr[3];
			int k;
			for (k = 0; k < 6; k++) Y[k] = r[2 * (g + k)] >> 6;
			for (k = 0; k < 3; k++) { Cr[k] = r[2 * (g + 2 * k) + 1] >> 6; Cb[k] = r[2 * (g + 2 * k + 1) + 1] >> 6; }
			o[4 * (g / 6) + 0] = Cb[0] | (Y[0] << 10) | (Cr[0] << 20);
			o[4 * (g / 6) + 1] = Y[1] | (Cb[1] << 10) | (Y[2] << 20);
			o[4 * (g / 6) + 2] = Cr[1] | (Y[3] << 10) | (Cb[2] << 20);
			o[4 * (g / 6) + 3] = Y[4] | (Cr[2] << 10) | (Y[5] << 20);
		}
	}
	free(yu);
}

/* ---- RGBA 4:4:4:4 samples decoded to BGRA (bottom row first) / BGRa (top row first) ------------------------------------------------------
 * Probed on the built reference and pinned in tests/test_oracle_vs_ref.py (this route draws no dither: the reference's output is the same from
 * call to call): every colour byte is the 12-bit component of the 16-bit reconstruction (orc_inv_spatial_to_rgb48 with four planes; the caller's pyramid
 * carries the lowpass bias 8 of the 8-bit RGB outputs, Codec/decoder.c:12294, which arrives here as + 2 -- the "rounding" the first fit of this model found)
 * >> 4, saturated to 255; the alpha byte takes the same 12-bit value through the reference's alpha expansion -- minus
 * alphacompandDCoffset 256, << 3, times alphacompandGain 9400 >> 16, >> 4 (Codec/codec.h:164-165; the arithmetic of the scalar code at
 * Codec/convert.c:6391-6396) --, clamped to [0, 255].  Bytes B, G, R, A; planes are G, R, B, A. */
void orc_inv_spatial_to_rgba8(PIXEL16 *const bands[4][4], int band_pitch, int w, int h, int precision, int display_height, int bottom_up, uint8_t *out, int out_pitch_bytes)
{
	const int W = 2 * w;
	uint16_t *tmp = (uint16_t *)malloc((size_t)2 * h * W * 4 * sizeof(uint16_t));
	int y, x, c;
	orc_inv_spatial_to_rgb48(bands, band_pitch, w, h, precision, 4, tmp, W * 4);
	for (y = 0; y < display_height; y++) {
		uint8_t *o = out + (size_t)(bottom_up ? display_height - 1 - y : y) * out_pitch_bytes;
		for (x = 0; x < W; x++) {
			int a;
			for (c = 0; c < 3; c++) {                      /* words R, G, B -> bytes B, G, R */
				const int v = (tmp[((size_t)y * W + x) * 4 + c] >> 4) >> 4;
				o[(size_t)x * 4 + (2 - c)] = (uint8_t)(v > 255 ? 255 : v);
			}
			a = (tmp[((size_t)y * W + x) * 4 + 3] >> 4) - 256;
			a = a < 0 ? 0 : (((a << 3) * 9400) >> 16) >> 4;
			o[(size_t)x * 4 + 3] = (uint8_t)(a > 255 ? 255 : a);
		}
	}
	free(tmp);
}

/* ---- RGB 4:4:4 samples decoded to b64a (words A, R, G, B) ---------------------------------------------------------------------------------
 * Probed on the built reference and pinned in tests/test_oracle_vs_ref.py: the colour words are those of the 16-bit planar-row route
 * (InvertHorizontalStrip16sToRow16u per plane, as for RG48 output) with ONE difference -- only the last band column (the last two pixels of a
 * row) behaves like that routine's scalar tail and may reach 65535; the RG48 route starts its tail at band column w - w % 8 - 9 -- and the alpha
 * word is the constant 0xfff0 (4095 << 4). */
void orc_inv_spatial_to_b64a_of_rgb444(PIXEL16 *const bands[4][4], int band_pitch, int w, int h, int precision, uint16_t *out, int out_pitch_words)
{
	static const int word_of_channel[4] = { 1, 0, 2, 3 };      /* plane G -> word 1, R -> 0, B -> 2 of the temporary R, G, B pixels */
	const int W = 2 * w;
	uint16_t *tmp = (uint16_t *)malloc((size_t)2 * h * W * 3 * sizeof(uint16_t));
	int y, x;
	orc_inv_spatial_to_packed16(bands, band_pitch, w, h, precision, 3, word_of_channel, w - 1, -1, tmp, W * 3);
	for (y = 0; y < 2 * h; y++)
		for (x = 0; x < W; x++) {
			uint16_t *o = out + (size_t)y * out_pitch_words + (size_t)x * 4;
			const uint16_t *t = tmp + ((size_t)y * W + x) * 3;
			o[0] = 0xfff0; o[1] = t[0]; o[2] = t[1]; o[3] = t[2];
		}
	free(tmp);
}

/* ---- 4:2:2 samples decoded to RG24 (8-bit B, G, R, bottom row first) ---------------------------------------------------------------------
 * decoder.c:16486 -> InvertHorizontalStrip16s.c:17508 InvertHorizontalStrip16sYUVtoRGB: the three planes as 16-bit rows (InvertHorizontalStrip16sToRow16u,
 * the rows orc_inv_spatial_to_yu64 restates) -> decoder.c:23151 ConvertRow16uToDitheredBuffer -> convert.c:10677 ConvertRow16uToDitheredRGB, whose vector
 * code is compiled out (`#if (0 && XMMOPT)`): the scalar loop at convert.c:11392-11448 does all columns.  Per pixel, with the pair's chroma (U = channel
 * 2, V = channel 1; :10684-10686): Y' = ((Y - (y_offset << 8)) * ymult) >> 7, U -= 32768, V -= 32768,
 *   R = (Y' + r_vmult * V + d) >> 15, G = (Y' - g_umult * (U >> 1) - g_vmult * (V >> 1) + d) >> 15, B = (Y' + 2 * b_umult * U + d) >> 15, saturated to 8 bits,
 * d = rand() & 0x7fff, drawn once per pixel and shared by its three components.  The oracle takes d as an input (0 and 32767 are the ends of the
 * interval every reference byte lies in).  Matrices :10707-10750 by the sample's colour space (1 = 601, 2 = 709, +4 = video range); STRICT_SATURATE is 0
 * (color.h:30), so SATURATE_Y / Cb / Cr are the identity. */
/* The conversion alone: `rows` rows of YU64 words (Y0 C1 Y1 C2 per pixel pair, `width` pixels) -> RG24 rows, bottom row first. */
void orc_yu64_to_rgb24(const uint16_t *yu, int yu_pitch_words, int width, int rows, int color_space, int d, uint8_t *out, int out_pitch_bytes)
{
	int y_offset = 16, ymult = 128 * 149, r_vmult = 230, g_vmult = 137, g_umult = 55, b_umult = 135;      /* COLOR_SPACE_CG_709 and the default */
	int y, x;
	switch (color_space & 7) {
	case 1: r_vmult = 204; g_vmult = 208; g_umult = 100; b_umult = 129; break;                                             /* CG_601 */
	case 5: y_offset = 0; ymult = 128 * 128; r_vmult = 175; g_vmult = 179; g_umult = 86; b_umult = 111; break;             /* VS_601 */
	case 6: y_offset = 0; ymult = 128 * 128; r_vmult = 197; g_vmult = 118; g_umult = 47; b_umult = 116; break;             /* VS_709 */
	default: break;
	}
	for (y = 0; y < rows; y++) {
		const uint16_t *r = yu + (size_t)y * yu_pitch_words;
		uint8_t *o = out + (size_t)(rows - 1 - y) * out_pitch_bytes;          /* DECODED_FORMAT_RGB24 is "inverted": bottom row first (decoder.c:23166) */
		for (x = 0; x < width; x += 2) {
			const int V = (int)r[2 * x + 1] - 32768, U = (int)r[2 * x + 3] - 32768;
			int k;
			for (k = 0; k < 2; k++) {
				const int Y1 = (((int)r[2 * x + 2 * k] - (y_offset << 8)) * ymult) >> 7;
				int R = (Y1 + r_vmult * V + d) >> 15, G = (Y1 - g_umult * (U >> 1) - g_vmult * (V >> 1) + d) >> 15, B = (Y1 + 2 * b_umult * U + d) >> 15;
				o[3 * (x + k) + 0] = (uint8_t)(B < 0 ? 0 : (B > 255 ? 255 : B));
				o[3 * (x + k) + 1] = (uint8_t)(G < 0 ? 0 : (G > 255 ? 255 : G));
				o[3 * (x + k) + 2] = (uint8_t)(R < 0 ? 0 : (R > 255 ? 255 : R));
			}
		}
	}
}

void orc_inv_spatial_to_rgb24_of_yuv422(PIXEL16 *const bands[3][4], const int band_pitch[3], int luma_w, int h, int precision, int display_height, int color_space,
                                        int d, uint8_t *out, int out_pitch_bytes)
{
	const int W = 2 * luma_w;
	uint16_t *yu = (uint16_t *)malloc((size_t)2 * h * W * 2 * sizeof(uint16_t));
	orc_inv_spatial_to_yu64(bands, band_pitch, luma_w, h, precision, yu, W * 2);
	orc_yu64_to_rgb24(yu, W * 2, W, display_height, color_space, d, out, out_pitch_bytes);
	free(yu);
}

/* ---- Bayer samples decoded to BYR4 (the raw mosaic, no demosaic) ------------------------------------------------------------------------
 * Codec/decoder.c:14738 (full resolution, BYR2 / BYR4 output): the four planes G, R-G, B-G, G1-G2 are reconstructed as 16-bit rows -- the planar
 * row route of RG48 output, orc_inv_spatial_to_rgb48 with four planes -- into decoder->RawBayer16, then Codec/bayer.c:13233 GenerateBYR2 turns every
 * quad back into its four samples: r = ((rg - 32768) << 1) + g, b likewise, g1 = g + (gd - 32768), g2 = g - (gd - 32768), each clamped to 16 bits,
 * and -- for BYR4 output of a sample without an encode-curve preset -- sent through BYR4LinearRestore[x >> 2], the table decoder.c:10714-10783
 * builds: (int)(log2lin(j / 16384, 90) * 65535) in float, with log2lin (Common/AVIExtendedHeader.h:148) = (float)((pow(b, i) - 1) / (b - 1)) evaluated
 * in double.  Red-green phase (BAYER_FORMAT_RED_GRN, the default without Bayer metadata): rows r g1 / g2 b.  Pinned word for word against the reference
 * decoder in tests/test_oracle_vs_ref.py. */
void orc_byr4_linear_restore_curve(uint16_t curve[16384])
{
	int j;
	const float base = 90.0f;                           /* encode_curve == 0: CURVE_TYPE_LOG, base 90 (decoder.c:10729-10733) */
	for (j = 0; j < 16384; j++) {
		const float i = (float)j / 16384.0f;
		const float lin = (float)((pow(base, i) - 1.0) / (base - 1.0));
		int val = (int)(lin * 65535.0f);
		if (val < 0) val = 0;
		if (val > 65535) val = 65535;
		curve[j] = (uint16_t)val;
	}
}

void orc_inv_spatial_to_byr4(PIXEL16 *const bands[4][4], int band_pitch, int w, int h, int precision, int display_quad_rows, const uint16_t *curve,
                             uint16_t *out, int out_pitch_words)
{
	const int W = 2 * w;                                /* quads per row */
	uint16_t *raw = (uint16_t *)malloc((size_t)2 * h * W * 4 * sizeof(uint16_t));
	int y, x;
	orc_inv_spatial_to_rgb48(bands, band_pitch, w, h, precision, 4, raw, W * 4);      /* words R-G, G, B-G, G1-G2 per quad (planes G, R-G, B-G, G1-G2 -> words 1, 0, 2, 3) */
	for (y = 0; y < display_quad_rows; y++) {
		uint16_t *a = out + (size_t)(2 * y) * out_pitch_words, *b = a + out_pitch_words;
		for (x = 0; x < W; x++) {
			const uint16_t *q = raw + ((size_t)y * W + x) * 4;
			const int g = q[1], rg = q[0], bg = q[2], gd = (int)q[3] - 32768;
			int r = ((rg - 32768) << 1) + g, bl = ((bg - 32768) << 1) + g, g1 = g + gd, g2 = g - gd;
			r = r < 0 ? 0 : (r > 0xffff ? 0xffff : r); bl = bl < 0 ? 0 : (bl > 0xffff ? 0xffff : bl);
			g1 = g1 < 0 ? 0 : (g1 > 0xffff ? 0xffff : g1); g2 = g2 < 0 ? 0 : (g2 > 0xffff ? 0xffff : g2);
			if (curve) { r = curve[r >> 2]; g1 = curve[g1 >> 2]; g2 = curve[g2 >> 2]; bl = curve[bl >> 2]; }
			else { r &= 0xfffe; g1 &= 0xfffe; g2 &= 0xfffe; bl &= 0xfffe; }      /* (BYR2 output, or a curve preset: bayer.c:13302-13308) */
			a[2 * x] = (uint16_t)r; a[2 * x + 1] = (uint16_t)g1;
			b[2 * x] = (uint16_t)g2; b[2 * x + 1] = (uint16_t)bl;
		}
	}
	free(raw);
}
