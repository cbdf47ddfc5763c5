/* oracle/cfhd_oracle_fwd.c -- TEST INFRASTRUCTURE ONLY (see cfhd_oracle.h).
 *
 * Forward path: pixel unpack, 2/6 biorthogonal analysis (horizontal then vertical), quantizer.
 *
 * Arithmetic model.  The reference's SSE2 bodies use saturating 16-bit adds in a fixed association
 * order; its scalar tails/borders use 32-bit sums followed by SATURATE.  The two only differ when
 * an intermediate leaves the int16 range, which no pixel unpacker can produce (inputs are <= 12
 * bits; see DESIGN.md "value ranges").  We restate the SIMD association order with saturation for
 * interior taps and the scalar form for the border taps, exactly where the reference uses them.
 */
#include "cfhd_oracle.h"
#include <stdlib.h>
#include <string.h>

static inline int sat16(int x) { return x < -32768 ? -32768 : (x > 32767 ? 32767 : x); }
static inline int adds(int a, int b) { return sat16(a + b); }
static inline int subs(int a, int b) { return sat16(a - b); }

/* Codec/convert.c:4667 UnpackRowYUV16s (default branch: value << shift; ch0=Y, ch1=V, ch2=U;
 * :4796-4800 luma, :4876-4878 / :5060-5066 chroma).  uyvy selects COLOR_FORMAT_UYVY byte order. */
void orc_unpack_yuyv_row(const uint8_t *in, PIXEL16 *out, int width, int channel, int shift, int uyvy)
{
	int i;
	if (channel == 0) {
		int off = uyvy ? 1 : 0;
		for (i = 0; i < width; i++) out[i] = (PIXEL16)(in[2 * i + off] << shift);
	} else {
		/* YUYV: Y0 U Y1 V ; UYVY: U Y0 V Y1 */
		int off = (channel == 2) ? (uyvy ? 0 : 1) : (uyvy ? 2 : 3);
		for (i = 0; i < width; i++) out[i] = (PIXEL16)(in[4 * i + off] << shift);
	}
}

/* Codec/spatial.c:253 FilterHorizontalRow16s (prescale 0) and :3669 FilterHorizontalRow10bit16s
 * (prescale 2: every tap sees (x+3)>>2, the lowpass is (x0+x1+3)>>2, :3774-3776,:3960).
 * Left border :277-286, right border :559-569, interior SIMD order :326-397. */
void orc_fwd_horizontal(const PIXEL16 *x, int width, int prescale, PIXEL16 *low, PIXEL16 *high)
{
	int half = width / 2;
	int k;
#define P(i) (prescale ? ((x[i] + 3) >> 2) : x[i])
	for (k = 0; k < half; k++) {
		if (prescale) low[k] = (PIXEL16)sat16((x[2 * k] + x[2 * k + 1] + 3) >> 2);
		else          low[k] = (PIXEL16)adds(x[2 * k], x[2 * k + 1]);
	}
	{
		int sum = 5 * P(0) - 11 * P(1) + 4 * P(2) + 4 * P(3) - P(4) - P(5) + 4;
		high[0] = (PIXEL16)sat16(sum >> 3);
	}
	for (k = 1; k < half - 1; k++) {
		int c = 2 * k;
		int sum = subs(0, P(c - 2));
		sum = subs(sum, P(c - 1));
		sum = adds(sum, P(c + 2));
		sum = adds(sum, P(c + 3));
		sum = adds(sum, 4);
		sum >>= 3;
		high[k] = (PIXEL16)adds(sum, subs(P(c), P(c + 1)));
	}
	{
		int c = width - 2;
		int sum = 11 * P(c) - 5 * P(c + 1) - 4 * P(c - 1) - 4 * P(c - 2) + P(c - 3) + P(c - 4) + 4;
		high[half - 1] = (PIXEL16)sat16(sum >> 3);
	}
#undef P
}

/* Codec/quantize.c:1395 QuantizeRow16sTo16s: sign * (((|x| + mid) * floor(65536/div)) >> 16)
 * with 16-bit wrap of (|x|+mid) and of the multiplier (_mm_set1_epi16), mid per :1415-1427. */
void orc_quantize_row(const PIXEL16 *in, PIXEL16 *out, int length, int divisor, int midpoint_prequant)
{
	int i, mid = 0;
	unsigned mult;
	if (midpoint_prequant >= 2 && midpoint_prequant < 9) {
		mid = divisor / midpoint_prequant;
		if (midpoint_prequant == 2 && mid) mid--;
	}
	if (divisor <= 1) { memmove(out, in, (size_t)length * sizeof(PIXEL16)); return; }
	mult = ((1u << 16) / (unsigned)divisor) & 0xffffu;
	for (i = 0; i < length; i++) {
		int v = in[i];
		int neg = v < 0;
		unsigned a = (unsigned)(neg ? -v : v) & 0xffffu;
		unsigned q;
		a = (a + (unsigned)mid) & 0xffffu;
		q = (a * mult) >> 16;
		out[i] = (PIXEL16)(neg ? -(int)q : (int)q);
	}
}

/* Vertical analysis of six consecutive rows of horizontal results producing one output row.
 * Codec/spatial.c:10026 FilterSpatialQuant16s / :12942 FilterSpatialV210Quant16s / :14726
 * FilterSpatialYUVQuant16s share it: top :10178-10185 (14880-14915), middle SIMD order
 * :10301-10351 (15065-15180), bottom :10539-10546 (15290-15330). */
static void vertical_row(PIXEL16 *const rows[6], int n, int position /*0 top,1 middle,2 bottom*/,
                         PIXEL16 *lowout, PIXEL16 *highout)
{
	int c;
	for (c = 0; c < n; c++) {
		int a0 = rows[0][c], a1 = rows[1][c], a2 = rows[2][c], a3 = rows[3][c], a4 = rows[4][c], a5 = rows[5][c];
		if (position == 0) {
			lowout[c] = (PIXEL16)sat16(a0 + a1);
			highout[c] = (PIXEL16)sat16((5 * a0 - 11 * a1 + 4 * a2 + 4 * a3 - a4 - a5 + 4) >> 3);
		} else if (position == 2) {
			lowout[c] = (PIXEL16)sat16(a4 + a5);
			highout[c] = (PIXEL16)sat16((11 * a4 - 5 * a5 - 4 * a3 - 4 * a2 + a1 + a0 + 4) >> 3);
		} else {
			int sum = subs(0, a0);
			sum = subs(sum, a1);
			sum = adds(sum, a4);
			sum = adds(sum, a5);
			sum = adds(sum, 4);
			sum >>= 3;
			lowout[c] = (PIXEL16)adds(a2, a3);
			highout[c] = (PIXEL16)adds(sum, subs(adds(0, a2), a3));
		}
	}
}

typedef void (*row_source_fn)(void *ctx, int row, PIXEL16 *dst);

static void fwd_spatial_generic(row_source_fn src, void *ctx, int width, int height, int prescale,
                                const int quant[4], int mpq, PIXEL16 *bands[4], int band_pitch)
{
	int half = width / 2, hh = height / 2;
	PIXEL16 *L = (PIXEL16 *)malloc((size_t)height * half * sizeof(PIXEL16));
	PIXEL16 *H = (PIXEL16 *)malloc((size_t)height * half * sizeof(PIXEL16));
	PIXEL16 *rowbuf = (PIXEL16 *)malloc((size_t)width * sizeof(PIXEL16));
	PIXEL16 *t1 = (PIXEL16 *)malloc((size_t)half * sizeof(PIXEL16));
	PIXEL16 *t2 = (PIXEL16 *)malloc((size_t)half * sizeof(PIXEL16));
	int r;
	for (r = 0; r < height; r++) {
		src(ctx, r, rowbuf);
		orc_fwd_horizontal(rowbuf, width, prescale, L + (size_t)r * half, H + (size_t)r * half);
	}
	for (r = 0; r < hh; r++) {
		int pos = (r == 0) ? 0 : (r == hh - 1 ? 2 : 1);
		int first = (r == 0) ? 0 : (r == hh - 1 ? height - 6 : 2 * r - 2);
		PIXEL16 *lr[6], *hr[6];
		int k;
		for (k = 0; k < 6; k++) { lr[k] = L + (size_t)(first + k) * half; hr[k] = H + (size_t)(first + k) * half; }
		/* LL and HL (vertical high of horizontal low) */
		vertical_row(lr, half, pos, bands[0] + (size_t)r * band_pitch, t1);
		orc_quantize_row(t1, bands[2] + (size_t)r * band_pitch, half, quant[2], mpq);
		/* LH (vertical low of horizontal high) and HH */
		vertical_row(hr, half, pos, t2, t1);
		orc_quantize_row(t2, bands[1] + (size_t)r * band_pitch, half, quant[1], mpq);
		orc_quantize_row(t1, bands[3] + (size_t)r * band_pitch, half, quant[3], mpq);
	}
	free(L); free(H); free(rowbuf); free(t1); free(t2);
}

struct plane_src { const PIXEL16 *in; int pitch; int width; };
static void plane_row(void *ctx, int row, PIXEL16 *dst)
{
	struct plane_src *s = (struct plane_src *)ctx;
	memcpy(dst, s->in + (size_t)row * s->pitch, (size_t)s->width * sizeof(PIXEL16));
}

void orc_fwd_spatial(const PIXEL16 *in, int in_pitch, int width, int height, int prescale,
                     const int quant[4], int midpoint_prequant, PIXEL16 *bands[4], int band_pitch)
{
	struct plane_src s = { in, in_pitch, width };
	fwd_spatial_generic(plane_row, &s, width, height, prescale, quant, midpoint_prequant, bands, band_pitch);
}

struct yuv_src { const uint8_t *in; int pitch; int width; int channel; int shift; int uyvy; };
static void yuv_row(void *ctx, int row, PIXEL16 *dst)
{
	struct yuv_src *s = (struct yuv_src *)ctx;
	orc_unpack_yuyv_row(s->in + (size_t)row * s->pitch, dst, s->width, s->channel, s->shift, s->uyvy);
}

/* Codec/wavelet.c:2823 TransformForwardSpatialYUV -> spatial.c:14726 FilterSpatialYUVQuant16s
 * (FilterHorizontalRowYUV16s :4005 = UnpackRowYUV16s + FilterHorizontalRow16s). */
void orc_fwd_spatial_yuv422(const uint8_t *in, int in_pitch_bytes, int width, int height,
                            int channel, int shift, int uyvy, const int quant[4], int midpoint_prequant,
                            PIXEL16 *bands[4], int band_pitch)
{
	struct yuv_src s = { in, in_pitch_bytes, width, channel, shift, uyvy };
	fwd_spatial_generic(yuv_row, &s, width, height, 0, quant, midpoint_prequant, bands, band_pitch);
}

/* Default encode curve of the Bayer input path: log base 90 over 14-bit linear input (MAX_INPUT_PRECISION, frame.c:4843), scaled to the codec
 * precision.  Codec/frame.c:5219-5235 (BYR4_LOGTABLE with LOGBASE 90) and Common/AVIExtendedHeader.h:124,153 CURVE_LIN2LOG = lin2log(); the
 * expression keeps the reference's float/double mix so that the truncation to int falls on the same side. */
#include <math.h>
void orc_byr4_log90_curve(int precision, int input_bits, uint16_t *curve)
{
	const int max_value = 1 << input_bits;
	int i;
	curve[0] = 0;
	for (i = 1; i < max_value; i++) {
		const float x = (float)i / (float)max_value;
		const float b = 90.0f;
		const float y = (float)(log10(x * (b - 1.0) + 1.0) / log10(b));       /* lin2log() returns float (AVIExtendedHeader.h:153-156) */
		curve[i] = (uint16_t)(int)(y * (float)((1 << precision) - 1));
	}
}

/* One row of 2x2 Bayer quads (red-green order: R G1 / G2 B) -> the four component planes G, R-G, B-G, G1-G2 of
 * ConvertBYR4ToFrame16s (frame.c:5380-5393): curve first, then g = (g1+g2)>>1, rg = ((r-g)>>1) + mid, bg likewise,
 * gd = (g1-g2+2*mid)>>1 with mid = 2^(precision-1). */
void orc_byr4_unpack_row(const uint16_t *line1, const uint16_t *line2, int width, int precision, int input_bits, const uint16_t *curve,
                         PIXEL16 *g_out, PIXEL16 *rg_out, PIXEL16 *bg_out, PIXEL16 *gd_out)
{
	const int mid = 1 << (precision - 1), sh = 16 - input_bits;
	int x;
	for (x = 0; x < width; x++) {
		const int r = curve[line1[2 * x] >> sh], g1 = curve[line1[2 * x + 1] >> sh], g2 = curve[line2[2 * x] >> sh], b = curve[line2[2 * x + 1] >> sh];
		const int g = (g1 + g2) >> 1;
		g_out[x] = (PIXEL16)g; rg_out[x] = (PIXEL16)(((r - g) >> 1) + mid); bg_out[x] = (PIXEL16)(((b - g) >> 1) + mid); gd_out[x] = (PIXEL16)((g1 - g2 + 2 * mid) >> 1);
	}
}

/* Interlaced ("frame") level 1 of a packed 4:2:2 picture, one channel.  Codec/wavelet.c:6076 TransformForwardFrameYUV: for every
 * pair of rows (2k, 2k+1) the temporal pair low = r0 + r1, high = r1 - r0 of the samples << (precision - 8) (temporal.c:1915
 * FilterTemporalRowYUYVChannelTo16s, saturating adds/subs :2160-2163), then
 *   LL, LH = FilterHorizontalRow16s(low)             (LL stored as is, LH through QuantizeRow16sTo16s),
 *   HL, HH = FilterHorizontalRowScaled16sDifferenceFiltered(high) (spatial.c:5327): the horizontal lowpass is quantized inside with
 *            midpoint = divisor / midpoint_prequant (no "-1 for prequant 2" here, :5360-5363) and then difference coded along the
 *            row, q[x] - q[x-1] (:5608-5611); HH through QuantizeRow16sTo16s.
 * (The raw store behind the vector loop, :5621-5622, sits under !_PREROLL and is compiled out: rows whose width is not a multiple
 * of 16 samples are regular too -- checked against reference samples of 720-wide pictures, chroma width 360.) */
static int quant_inside(int v, int divisor, int mpq)
{
	int mid = (mpq >= 2 && mpq < 9) ? divisor / mpq : 0;
	int mult = (int)((1u << 16) / (unsigned)divisor);
	int a = v < 0 ? -v : v;
	if (divisor <= 1) return v;
	a = (int)(((long long)(a + mid) * mult) >> 16);
	return v < 0 ? -a : a;
}

void orc_fwd_frame_yuv422(const uint8_t *in, int in_pitch_bytes, int width, int height, int channel, int shift, int uyvy,
                          const int quant[4], int midpoint_prequant, PIXEL16 *bands[4], int band_pitch)
{
	const int half = width / 2;
	PIXEL16 *r0 = (PIXEL16 *)malloc((size_t)width * 2), *r1 = (PIXEL16 *)malloc((size_t)width * 2);
	PIXEL16 *tl = (PIXEL16 *)malloc((size_t)width * 2), *th = (PIXEL16 *)malloc((size_t)width * 2);
	PIXEL16 *lo = (PIXEL16 *)malloc((size_t)half * 2), *hi = (PIXEL16 *)malloc((size_t)half * 2);
	int k, x;
	for (k = 0; k < height / 2; k++) {
		orc_unpack_yuyv_row(in + (size_t)(2 * k) * in_pitch_bytes, r0, width, channel, shift, uyvy);
		orc_unpack_yuyv_row(in + (size_t)(2 * k + 1) * in_pitch_bytes, r1, width, channel, shift, uyvy);
		for (x = 0; x < width; x++) { tl[x] = (PIXEL16)adds(r0[x], r1[x]); th[x] = (PIXEL16)subs(r1[x], r0[x]); }
		orc_fwd_horizontal(tl, width, 0, bands[0] + (size_t)k * band_pitch, hi);
		orc_quantize_row(hi, bands[1] + (size_t)k * band_pitch, half, quant[1], midpoint_prequant);
		orc_fwd_horizontal(th, width, 0, lo, hi);
		{
			PIXEL16 *hl = bands[2] + (size_t)k * band_pitch;
			int prev = 0;
			for (x = 0; x < half; x++) {
				const int q = quant_inside(lo[x], quant[2], midpoint_prequant);
				hl[x] = (PIXEL16)sat16(q - prev);
				prev = q;
			}
		}
		orc_quantize_row(hi, bands[3] + (size_t)k * band_pitch, half, quant[3], midpoint_prequant);
	}
	free(r0); free(r1); free(tl); free(th); free(lo); free(hi);
}

/* ---- deep RGB -> YUV 4:2:2 10-bit planes (RG48 / b64a encoded as CFHD_ENCODED_FORMAT_YUV_422) ------------------------------------
 * Codec/frame.c:6731 ConvertAnyDeep444to422: per pixel y = ((yr r + yg g + yb b) >> 20) + y_offset, clamped to [0, 1023]; per pixel pair
 * u = ((u0 + u1) >> 1) + 512 with u_i = (-ur r - ug g + ub b) >> 20 (arithmetic shifts of negative sums), v likewise, clamped; planes Y,
 * channel 1 = v, channel 2 = u (:6795-6797); picture rows beyond the display height repeat the last row (:7176).  16-bit words r, g, b at
 * `words_per_pixel` apart, r at in[0].  color_space: 0 = computer-systems 709 (what the SDK passes by default), 1 = video-systems 709,
 * 2 = computer-systems 601, 3 = video-systems 601 (:6803-6870). */
static const int k_rgb2yuv[4][10] = {
	{ 2998, 10060, 1016, 64, 1655, 5538, 7193, 7193, 6537, 655 },
	{ 3490, 11715, 1180, 0, 1917, 6455, 8372, 8372, 7602, 770 },
	{ 4211, 8258, 1606, 64, 2425, 4768, 7193, 7193, 6029, 1163 },
	{ 4899, 9617, 1868, 0, 2818, 5554, 8372, 8372, 7012, 1360 },
};
void orc_rgb16_to_yuv422(const uint16_t *in, int in_pitch_words, int words_per_pixel, int width, int display_height, int height, int color_space,
                         PIXEL16 *y_plane, int y_pitch, PIXEL16 *c1_plane, PIXEL16 *c2_plane, int c_pitch)
{
	const int *m = k_rgb2yuv[color_space & 3];
	int row, x;
	for (row = 0; row < height; row++) {
		const uint16_t *p = in + (size_t)(row < display_height ? row : display_height - 1) * in_pitch_words;
		for (x = 0; x < width; x += 2) {
			int u = 0, v = 0, k;
			for (k = 0; k < 2; k++) {
				const int r = p[(size_t)(x + k) * words_per_pixel], g = p[(size_t)(x + k) * words_per_pixel + 1], b = p[(size_t)(x + k) * words_per_pixel + 2];
				int y = ((m[0] * r + m[1] * g + m[2] * b) >> 20) + m[3];
				u += (-m[4] * r - m[5] * g + m[6] * b) >> 20;
				v += (m[7] * r - m[8] * g - m[9] * b) >> 20;
				y_plane[(size_t)row * y_pitch + x + k] = (PIXEL16)(y < 0 ? 0 : (y > 1023 ? 1023 : y));
			}
			u = (u >> 1) + 512; v = (v >> 1) + 512;
			c2_plane[(size_t)row * c_pitch + x / 2] = (PIXEL16)(u < 0 ? 0 : (u > 1023 ? 1023 : u));
			c1_plane[(size_t)row * c_pitch + x / 2] = (PIXEL16)(v < 0 ? 0 : (v > 1023 ? 1023 : v));
		}
	}
}

/* ---- 8-bit RGB(A) -> YUV 4:2:2 10-bit planes (RG24 / BGRA / BGRa encoded as CFHD_ENCODED_FORMAT_YUV_422) ---------------------------------
 * Codec/frame.c:378 ConvertRGB32to10bitYUVFrame, row by row: bytes B, G, R(, A) become 16-bit planes (byte << 8, RGB2YUV.c:326, :365),
 * RGB2YUV.c:1404 PlanarRGB16toPlanarYUV16 turns them into 16-bit Y, U, V (13-bit coefficients from the float constants of the matrix, each
 * product of a 15-bit sample shifted down 16 on its own, sum << 2 plus the 14-bit offset, clamped to 14 bits, << 2), and RGB2YUV.c:736
 * PlanarYUV16toChannelYUYV16 stores Y >> 6 and one chroma sample per pixel pair: the EVEN pixel's chroma >> 6 in the 16-pixel blocks its vector
 * loop covers (:846-873), the pair's average (a + b) >> 7 in the columns after them (:877-883).  U goes to channel 2, V to channel 1 (frame.c:424-428).
 * Source rows are read bottom row first (frame.c:419-420; top row first when `top_down`: the caller flipped RGB32_INVERTED before, encoder.c:2398-2402);
 * picture rows beyond the display height hold Y = 64, U = V = 512 (frame.c:466-500).  color_space as in orc_rgb16_to_yuv422. */
void orc_rgb8_to_yuv422(const uint8_t *in, int in_pitch, int bytes_per_pixel, int top_down, int width, int display_height, int height, int color_space,
                        PIXEL16 *y_plane, int y_pitch, PIXEL16 *c1_plane, PIXEL16 *c2_plane, int c_pitch)
{
	const float fp = (float)(1 << 13);
	static const float k[4][9] = {
		{ 0.183f, 0.614f, 0.062f, 0.101f, 0.338f, 0.439f, 0.439f, 0.399f, 0.040f },      /* computer-systems 709 (the default) */
		{ 0.213f, 0.715f, 0.072f, 0.117f, 0.394f, 0.511f, 0.511f, 0.464f, 0.047f },      /* video-systems 709 */
		{ 0.257f, 0.504f, 0.098f, 0.148f, 0.291f, 0.439f, 0.439f, 0.368f, 0.071f },      /* computer-systems 601 */
		{ 0.299f, 0.587f, 0.114f, 0.172f, 0.339f, 0.511f, 0.511f, 0.428f, 0.083f },      /* video-systems 601 */
	};
	const int cs = color_space & 3;
	const int y_offset = (cs == 0 || cs == 2) ? (((65536 * 16) >> 8) >> 2) : 0, c_offset = 32768 >> 2;
	const int width16 = width & 0xfff0;
	int m[9], i, row, x;
	uint16_t *yuv = (uint16_t *)malloc((size_t)width * 3 * sizeof(uint16_t));
	for (i = 0; i < 9; i++) m[i] = (int)(fp * k[cs][i]);
	for (row = 0; row < height; row++) {
		PIXEL16 *yo = y_plane + (size_t)row * y_pitch, *c1 = c1_plane + (size_t)row * c_pitch, *c2 = c2_plane + (size_t)row * c_pitch;
		if (row >= display_height) {
			for (x = 0; x < width; x++) yo[x] = 64;
			for (x = 0; x < width / 2; x++) c1[x] = c2[x] = 512;
			continue;
		}
		{
			const uint8_t *p = in + (size_t)(top_down ? row : display_height - 1 - row) * in_pitch;
			for (x = 0; x < width; x++) {
				const int B = (p[(size_t)x * bytes_per_pixel] << 8) >> 1, G = (p[(size_t)x * bytes_per_pixel + 1] << 8) >> 1, R = (p[(size_t)x * bytes_per_pixel + 2] << 8) >> 1;
				int Y = ((((m[0] * R) >> 16) + ((m[1] * G) >> 16) + ((m[2] * B) >> 16)) << 2) + y_offset;
				int U = ((((-m[3] * R) >> 16) + ((-m[4] * G) >> 16) + ((m[5] * B) >> 16)) * 4) + c_offset;
				int V = ((((m[6] * R) >> 16) + ((-m[7] * G) >> 16) + ((-m[8] * B) >> 16)) * 4) + c_offset;
				if (Y < 0) Y = 0; if (Y > 16383) Y = 16383;
				if (U < 0) U = 0; if (U > 16383) U = 16383;
				if (V < 0) V = 0; if (V > 16383) V = 16383;
				yuv[x] = (uint16_t)(Y << 2); yuv[width + x] = (uint16_t)(U << 2); yuv[2 * width + x] = (uint16_t)(V << 2);
			}
		}
		for (x = 0; x < width; x += 2) {
			yo[x] = (PIXEL16)(yuv[x] >> 6); yo[x + 1] = (PIXEL16)(yuv[x + 1] >> 6);
			if (x < width16) {
				c2[x / 2] = (PIXEL16)((((yuv[width + x] >> 1) & 0xffff) << 1) >> 6);
				c1[x / 2] = (PIXEL16)((((yuv[2 * width + x] >> 1) & 0xffff) << 1) >> 6);
			} else {
				c2[x / 2] = (PIXEL16)((yuv[width + x] + yuv[width + x + 1]) >> 7);
				c1[x / 2] = (PIXEL16)((yuv[2 * width + x] + yuv[2 * width + x + 1]) >> 7);
			}
		}
	}
	free(yuv);
}

/* ---- BYR5 (12-bit Bayer, "packed line of 8-bit then line of 4-bit remainder") -> the four component planes ---------------------------------
 * Codec/frame.c:5473 ConvertBYR5ToFrame16s, one row pair of the mosaic = one row of the planes: 4 * width samples lie as 4 * width high bytes
 * (four runs of `width`: for the red-green order R, G1, G2, B -- :5591-5596) followed by 2 * width bytes of low nibbles, the even sample's in the low
 * half of the byte (:5540-5563: value = high << 4 | nibble).  No encode curve: G = (G1 + G2) >> 1, then (R - G + 4096) >> 1, (B - G + 4096) >> 1,
 * (G1 - G2 + 4096) >> 1 (:5647-5663).  `row` points at the 6 * width bytes of the row pair. */
void orc_byr5_unpack_row(const uint8_t *row, int width, PIXEL16 *g, PIXEL16 *rg, PIXEL16 *bg, PIXEL16 *dg)
{
	const uint8_t *nib = row + (size_t)4 * width;
	int x, k;
	for (x = 0; x < width; x++) {
		int v[4];
		for (k = 0; k < 4; k++) { const int s = k * width + x; v[k] = (row[s] << 4) | ((nib[s >> 1] >> (4 * (s & 1))) & 15); }
		{
			const int gg = (v[1] + v[2]) >> 1;
			g[x] = (PIXEL16)gg;
			rg[x] = (PIXEL16)((v[0] - gg + 4096) >> 1);
			bg[x] = (PIXEL16)((v[3] - gg + 4096) >> 1);
			dg[x] = (PIXEL16)((v[1] - v[2] + 4096) >> 1);
		}
	}
}
