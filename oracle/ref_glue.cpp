/* oracle/ref_glue.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin C entry points (prefix ref_) around the unmodified reference so that tests and the
 * cpu_baseline leg of bench.py can drive it through ctypes:
 *   - Qbist synthetic frames exactly as Example/TestCFHD.cpp:1149-1150,1208 generates them
 *   - PSNR exactly as Example/utils.cpp:471 computes it
 * The CFHD_* C ABI of the reference itself is exported by the same .so and is called directly.
 */
#include <stdint.h>
#include <string.h>
#include "CFHDTypes.h"
#include "qbist.h"
#include "utils.h"

extern "C" {

void ref_qbist_reset(unsigned int seed)
{
	GetRand(seed);
	initBaseTransform();
}

/* buf must hold width*height*8 bytes: Qbist renders RGBA16 in place (TestCFHD.cpp:1178). */
void ref_qbist_frame(int width, int height, int pitch, unsigned int pixelFormat, int alpha, unsigned char *buf)
{
	RunQBist(width, height, pitch, (CFHD_PixelFormat)pixelFormat, alpha, buf);
}

int ref_frame_pitch(unsigned int pixelFormat, int width)
{
	return FramePitch4PixelFormat((CFHD_PixelFormat)pixelFormat, width);
}

float ref_psnr(void *a, void *b, int width, int height, unsigned int pixelFormat, int scale)
{
	return PSNR(a, b, width, height, (CFHD_PixelFormat)pixelFormat, scale);
}

} // extern "C"

/* ------------------------------------------------------------------------------------------
 * Kernel-level known-answer entry points: call the reference's own (SSE2) row/plane routines
 * on caller-supplied planes.  All planes are copied into 16-byte aligned, 16-byte-pitched
 * scratch (the reference uses _mm_load_si128) and copied back tightly packed
 * (pitch == width) so the Python side can use plain numpy arrays.
 * ------------------------------------------------------------------------------------------ */
#include <stdlib.h>
extern "C" {
#include "config.h"
#include "image.h"
#include "spatial.h"
#include "quantize.h"
extern int g_midpoint_prequant;   /* Codec/quantize.c:183 */
}

namespace {
struct APlane {
	PIXEL *p; int pitch_bytes; int w, h;
	APlane(int w_, int h_) : w(w_), h(h_) {
		pitch_bytes = ((w * 2 + 15) / 16) * 16;
		void *m = 0; if (posix_memalign(&m, 64, (size_t)pitch_bytes * h + 64)) m = 0;
		p = (PIXEL *)m; memset(p, 0, (size_t)pitch_bytes * h);
	}
	~APlane() { free(p); }
	void load(const int16_t *src) { for (int r = 0; r < h; r++) memcpy((char *)p + (size_t)r * pitch_bytes, src + (size_t)r * w, w * 2); }
	void store(int16_t *dst) const { for (int r = 0; r < h; r++) memcpy(dst + (size_t)r * w, (char *)p + (size_t)r * pitch_bytes, w * 2); }
};
void *scratch(size_t n) { void *m = 0; if (posix_memalign(&m, 64, n)) m = 0; memset(m, 0, n); return m; }
}

extern "C" {

/* Forward level from a 16-bit plane. prescale 0 -> FilterSpatialQuant16s (spatial.c:10026),
 * prescale 2 -> FilterSpatialV210Quant16s (spatial.c:12942), as wavelet.c:2506-2528 selects. */
void ref_fwd_spatial(const int16_t *in, int width, int height, int prescale, const int quant[4], int midpoint_prequant,
                     int16_t *ll, int16_t *lh, int16_t *hl, int16_t *hh)
{
	APlane src(width, height), b0(width / 2, height / 2), b1(width / 2, height / 2), b2(width / 2, height / 2), b3(width / 2, height / 2);
	size_t bsz = (size_t)(width * 2 + 64) * 40; PIXEL *buf = (PIXEL *)scratch(bsz);
	ROI roi = { width, height }; int q[4] = { quant[0], quant[1], quant[2], quant[3] };
	g_midpoint_prequant = midpoint_prequant;
	src.load(in);
	if (prescale == 2)
		FilterSpatialV210Quant16s(src.p, src.pitch_bytes, b0.p, b0.pitch_bytes, b1.p, b1.pitch_bytes, b2.p, b2.pitch_bytes, b3.p, b3.pitch_bytes, buf, bsz, roi, q);
	else
		FilterSpatialQuant16s(src.p, src.pitch_bytes, b0.p, b0.pitch_bytes, b1.p, b1.pitch_bytes, b2.p, b2.pitch_bytes, b3.p, b3.pitch_bytes, buf, bsz, roi, q);
	b0.store(ll); b1.store(lh); b2.store(hl); b3.store(hh);
	free(buf);
}

/* Forward level 1 from packed YUYV/UYVY: FilterSpatialYUVQuant16s (spatial.c:14726). width = channel width. */
void ref_fwd_spatial_yuv(const uint8_t *in, int pitch_bytes, int width, int height, int channel, int color_format, int precision,
                         const int quant[4], int midpoint_prequant, int16_t *ll, int16_t *lh, int16_t *hl, int16_t *hh)
{
	int frame_width = (channel == 0) ? width : width * 2;
	APlane b0(width / 2, height / 2), b1(width / 2, height / 2), b2(width / 2, height / 2), b3(width / 2, height / 2);
	size_t bsz = (size_t)(frame_width * 2 + 64) * 40; PIXEL *buf = (PIXEL *)scratch(bsz);
	size_t isz = (size_t)pitch_bytes * height; uint8_t *ain = (uint8_t *)scratch(isz + 64); memcpy(ain, in, isz);
	ROI roi = { width, height }; int q[4] = { quant[0], quant[1], quant[2], quant[3] };
	FRAME_INFO info; InitFrameInfo(&info, frame_width, height, color_format);
	g_midpoint_prequant = midpoint_prequant;
	FilterSpatialYUVQuant16s(ain, pitch_bytes, b0.p, b0.pitch_bytes, b1.p, b1.pitch_bytes, b2.p, b2.pitch_bytes, b3.p, b3.pitch_bytes,
	                         buf, bsz, roi, channel, q, &info, precision, 0, 0);
	b0.store(ll); b1.store(lh); b2.store(hl); b3.store(hh);
	free(buf); free(ain);
}

/* Inverse level into a 16-bit plane: InvertSpatialQuant16s (spatial.c:21877) or, when the level was
 * prescaled, InvertSpatialQuantDescale16s (spatial.c:22414). Bands are w x h, output 2w x 2h. */
void ref_inv_spatial(const int16_t *ll, const int16_t *lh, const int16_t *hl, const int16_t *hh, int w, int h, int descale, int16_t *out)
{
	APlane b0(w, h), b1(w, h), b2(w, h), b3(w, h), dst(2 * w, 2 * h);
	size_t bsz = (size_t)(w * 2 + 64) * 40; PIXEL *buf = (PIXEL *)scratch(bsz);
	ROI roi = { w, h }; int q[4] = { 1, 1, 1, 1 };
	b0.load(ll); b1.load(lh); b2.load(hl); b3.load(hh);
	if (descale)
		InvertSpatialQuantDescale16s(b0.p, b0.pitch_bytes, b1.p, b1.pitch_bytes, b2.p, b2.pitch_bytes, b3.p, b3.pitch_bytes, dst.p, dst.pitch_bytes, roi, buf, bsz, descale, q);
	else
		InvertSpatialQuant16s(b0.p, b0.pitch_bytes, b1.p, b1.pitch_bytes, b2.p, b2.pitch_bytes, b3.p, b3.pitch_bytes, dst.p, dst.pitch_bytes, roi, buf, bsz, q);
	dst.store(out);
	free(buf);
}

void ref_quantize_row(const int16_t *in, int16_t *out, int length, int divisor, int midpoint_prequant)
{
	int n = ((length + 7) / 8) * 8 + 8;
	PIXEL *a = (PIXEL *)scratch((size_t)n * 2), *b = (PIXEL *)scratch((size_t)n * 2);
	memcpy(a, in, (size_t)length * 2);
	g_midpoint_prequant = midpoint_prequant;
	QuantizeRow16sTo16s(a, b, length, divisor);
	memcpy(out, b, (size_t)length * 2);
	free(a); free(b);
}

} // extern "C"
