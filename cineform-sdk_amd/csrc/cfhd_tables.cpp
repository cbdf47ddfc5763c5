// cfhd_tables.cpp -- entropy-code tables, quantizer derivation and pyramid geometry (host side).
//
// Replaces, for the hot path: Codec/codebooks.c InitCodebooks (:202) / FillVleTable (:1032) /
// ComputeRunLengthCodeTable (:401) / FillRunLengthCodeTable (:499), Codec/quantize.c
// QuantizationSetQuality (:186) / SetTransformQuantization (:2865), Codec/wavelet.c
// SetTransformScale (:7022) / SetTransformPrescale (:1710) / AllocWaveletStack (:427).
#include "cfhd_core.h"
#include "cfhd_codebook_data.h"
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <mutex>
#include <algorithm>
#include <vector>

namespace cfhd {

// ------------------------------------------------------------------------------------------
// Entropy tables
// ------------------------------------------------------------------------------------------
namespace {

struct BaseCode { uint32_t bits; int size; int run; int mag; };

void build_entropy_tables(EntropyTables *t, const uint8_t *mag_len, const uint32_t *mag_code, int num_mags,
                          const uint32_t (*runs)[3], int num_runs, const uint32_t *band_end, bool cubic)
{
	// Companding: the encoder maps a quantized magnitude m (0..1023) to the largest code index i with
	// expand(i) <= m, where expand(i) = i + floor(i^3 * 768 / 2^24)  (codebooks.c:1053-1078).
	int inverse[1025];
	for (int i = 0; i < 1025; i++) inverse[i] = 0;
	for (int i = 0; i < 256; i++) {
		int e = cubic ? i + (int)(((int64_t)i * i * i * 768) >> 24) : i;
		t->mag_expand[i] = (uint16_t)e;
		if (cubic && i > 0) inverse[e > 1023 ? 1023 : e] = i;
	}
	if (cubic) {
		int last = 0;
		for (int i = 0; i < 1025; i++) { if (inverse[i]) last = inverse[i]; else inverse[i] = last; }
	}
	for (int idx = 0; idx < 2048; idx++) {
		int value = (idx & 1024) ? (idx & 1023) - 1024 : idx;
		int mag = abs(value);
		if (cubic) mag = inverse[mag];
		if (mag > num_mags - 1) mag = num_mags - 1;
		uint32_t code = mag_code[mag];
		int size = mag_len[mag];
		if (value > 0) { code = (code << 1); size++; }            // sign bit 0 = positive (vlc.h:81-84)
		else if (value < 0) { code = (code << 1) | 1u; size++; }
		t->value_code[idx] = ((uint32_t)size << 27) | code;
	}

	// Composite run-length table: greedy concatenation of the base run codes (longest first) while the
	// composite stays within 31 bits; a single zero is the magnitude-0 code.
	RunCode book[16]; int n = 0;
	bool has_one = false;
	for (int i = 0; i < num_runs; i++) {
		book[n].bits = runs[i][0]; book[n].size = (uint8_t)runs[i][1]; book[n].count = (uint16_t)runs[i][2];
		if (book[n].count == 1) has_one = true;
		n++;
	}
	if (!has_one) { book[n].bits = mag_code[0]; book[n].size = mag_len[0]; book[n].count = 1; n++; }
	std::sort(book, book + n, [](const RunCode &a, const RunCode &b) { return a.count > b.count; });
	for (int len = 0; len < 3072; len++) {
		uint32_t word = 0; int size = 0, remaining = len;
		bool full = false;
		for (int j = 0; j < n && remaining > 0 && !full; j++) {
			int rep = remaining / book[j].count, k = 0;
			for (; k < rep; k++) {
				if (book[j].size > 31 - size) { if (size) full = true; break; }
				word = (word << book[j].size) | book[j].bits;
				size += book[j].size;
			}
			remaining -= k * book[j].count;
		}
		t->run_bits[len] = word; t->run_size[len] = (uint8_t)size; t->run_count[len] = (uint16_t)(len - remaining);
	}
	t->band_end_bits = band_end[0]; t->band_end_size = (int)band_end[1];

	// Decoder LUT on the next kDecBits bits.
	const int K = EntropyTables::kDecBits;
	memset(t->dec_lut, 0, sizeof(t->dec_lut));
	auto fill = [&](uint32_t code, int size, uint32_t payload) {
		if (size > K) return;
		uint32_t base = code << (K - size);
		for (uint32_t s = 0; s < (1u << (K - size)); s++) t->dec_lut[base + s] = payload | (uint32_t)size;
	};
	for (int m = 0; m < num_mags; m++) {
		if (m == 0) fill(mag_code[0], mag_len[0], (1u << 5));                       // one zero
		else fill(mag_code[m], mag_len[m], ((uint32_t)t->mag_expand[m] << 16));    // magnitude, sign bit follows
	}
	for (int i = 0; i < num_runs; i++) fill(runs[i][0], (int)runs[i][1], (runs[i][2] << 5));
}

EntropyTables g_tables[3];
std::once_flag g_tables_once;

} // namespace

const EntropyTables *entropy_tables(int codebook)
{
	std::call_once(g_tables_once, [] {
		build_entropy_tables(&g_tables[1], cfhd_cs17_mag_len, cfhd_cs17_mag_code, CFHD_CS17_NUM_MAGS,
		                     cfhd_cs17_run, CFHD_CS17_NUM_RUNS, cfhd_cs17_band_end, true);
		build_entropy_tables(&g_tables[2], cfhd_cs18_mag_len, cfhd_cs18_mag_code, CFHD_CS18_NUM_MAGS,
		                     cfhd_cs18_run, CFHD_CS18_NUM_RUNS, cfhd_cs18_band_end, false);
	});
	if (codebook != 1 && codebook != 2) return nullptr;
	return &g_tables[codebook];
}

// Serialised dev::DecTables (cfhd_entropy_kernels.h): multi[4096] (8 bytes each), lut1[4096], lut2_size, lut2[].
std::vector<uint32_t> build_dec_tables(int codebook)
{
	const uint8_t *ml = codebook == 2 ? cfhd_cs18_mag_len : cfhd_cs17_mag_len;
	const uint32_t *mc = codebook == 2 ? cfhd_cs18_mag_code : cfhd_cs17_mag_code;
	const uint32_t (*rn)[3] = codebook == 2 ? cfhd_cs18_run : cfhd_cs17_run;
	const uint32_t *be = codebook == 2 ? cfhd_cs18_band_end : cfhd_cs17_band_end;
	const EntropyTables *t = entropy_tables(codebook);
	struct Code { uint32_t bits; int len; uint32_t payload; };
	std::vector<Code> codes;
	for (int m = 0; m < 256; m++) codes.push_back({ mc[m], ml[m], m == 0 ? (1u << 5) : ((uint32_t)t->mag_expand[m] << 16) });
	for (int i = 0; i < 7; i++) codes.push_back({ rn[i][0], (int)rn[i][1], rn[i][2] << 5 });
	codes.push_back({ be[0], (int)be[1], 0xffffu << 16 });
	const int K = 12;
	std::vector<uint32_t> lut1(1u << K, 0), lut2;
	// long codes: group by their 12-bit prefix; each group gets a table indexed by the following (maxlen - 12) bits
	std::vector<int> maxlen(1u << K, 0);
	for (const Code &c : codes) if (c.len > K) { uint32_t p = c.bits >> (c.len - K); if (c.len > maxlen[p]) maxlen[p] = c.len; }
	for (uint32_t p = 0; p < (1u << K); p++) if (maxlen[p]) {
		const int nb = maxlen[p] - K;
		lut1[p] = ((uint32_t)lut2.size() << 10) | ((uint32_t)nb << 5) | 31u;
		lut2.resize(lut2.size() + ((size_t)1 << nb), 0);
	}
	for (const Code &c : codes) {
		if (c.len <= K) {
			const uint32_t base = c.bits << (K - c.len);
			for (uint32_t s = 0; s < (1u << (K - c.len)); s++) lut1[base + s] = c.payload | (uint32_t)c.len;
		} else {
			const uint32_t p = c.bits >> (c.len - K);
			const uint32_t e = lut1[p];
			const int nb = (int)((e >> 5) & 31u);
			const uint32_t rest = c.bits & ((1u << (c.len - K)) - 1);
			const uint32_t base = (e >> 10) + (rest << (nb - (c.len - K)));
			for (uint32_t s = 0; s < (1u << (nb - (c.len - K))); s++) lut2[base + s] = c.payload | (uint32_t)c.len;
		}
	}
	// Multi-symbol table: greedily take complete symbols (with their sign bits) from each 12-bit window, at most two values.
	std::vector<uint32_t> out;
	for (uint32_t win = 0; win < (1u << K); win++) {
		int used = 0, pre = 0, mid = 0, post = 0, v1 = 0, v2 = 0;
		for (;;) {
			const uint32_t rest = (win << used) & ((1u << K) - 1);        // remaining bits, left aligned in K bits
			const uint32_t e = lut1[rest];
			const int len = (int)(e & 31u);
			if (len == 0 || len == 31 || len > K - used) break;           // incomplete / long / does not fit
			const uint32_t mag = e >> 16;
			if (mag == 0xffffu) break;
			if (mag) {
				if (len + 1 > K - used) break;                            // the sign bit must be inside the window too
				if (v2) break;
				const int negative = (int)((rest >> (K - len - 1)) & 1u);
				const int v = negative ? -(int)mag : (int)mag;
				if (!v1) v1 = v; else v2 = v;
				used += len + 1;
			} else {
				const int run = (int)((e >> 5) & 0x7ffu);
				int *slot = v2 ? &post : (v1 ? &mid : &pre);
				const int limit = slot == &pre ? 0xfff : 0xff;
				if (*slot + run > limit) break;
				*slot += run;
				used += len;
			}
		}
		// a trailing zero run after v1 (no v2) was accumulated in `mid`; keep the layout consistent: zeros after the last value = post
		if (v1 && !v2) { post = mid; mid = 0; }
		out.push_back((uint32_t)(used & 15) | ((uint32_t)pre << 4) | ((uint32_t)(uint16_t)(int16_t)v1 << 16));
		out.push_back((uint32_t)(mid & 0xff) | ((uint32_t)(post & 0xff) << 8) | ((uint32_t)(uint16_t)(int16_t)v2 << 16));
	}
	out.insert(out.end(), lut1.begin(), lut1.end());
	out.push_back((uint32_t)lut2.size());
	out.insert(out.end(), lut2.begin(), lut2.end());
	return out;
}

int raw_codes(int codebook, RawCode *out)
{
	const uint8_t *ml = codebook == 2 ? cfhd_cs18_mag_len : cfhd_cs17_mag_len;
	const uint32_t *mc = codebook == 2 ? cfhd_cs18_mag_code : cfhd_cs17_mag_code;
	const uint32_t (*rn)[3] = codebook == 2 ? cfhd_cs18_run : cfhd_cs17_run;
	const uint32_t *be = codebook == 2 ? cfhd_cs18_band_end : cfhd_cs17_band_end;
	int n = 0;
	out[n++] = RawCode{ mc[0], ml[0], 0, 1 };                          // magnitude 0 = a single zero
	for (int m = 1; m < 256; m++) out[n++] = RawCode{ mc[m], ml[m], 1, m };
	for (int i = 0; i < 7; i++) out[n++] = RawCode{ rn[i][0], (int)rn[i][1], 0, (int)rn[i][2] };
	out[n++] = RawCode{ be[0], (int)be[1], 2, 0 };
	return n;
}

// Raw base codes for slow-path decoding of code words longer than kDecBits.
int slow_decode_symbol(int codebook, uint32_t window /*next 32 bits, MSB first*/, int *size, int *run, int *mag, bool *band_end)
{
	const uint8_t *ml = codebook == 2 ? cfhd_cs18_mag_len : cfhd_cs17_mag_len;
	const uint32_t *mc = codebook == 2 ? cfhd_cs18_mag_code : cfhd_cs17_mag_code;
	const uint32_t (*rn)[3] = codebook == 2 ? cfhd_cs18_run : cfhd_cs17_run;
	const uint32_t *be = codebook == 2 ? cfhd_cs18_band_end : cfhd_cs17_band_end;
	const EntropyTables *t = entropy_tables(codebook);
	*band_end = false; *run = 0; *mag = 0;
	if ((window >> (32 - be[1])) == be[0]) { *size = (int)be[1]; *band_end = true; return 0; }
	for (int i = 0; i < 7; i++) if ((window >> (32 - rn[i][1])) == rn[i][0]) { *size = (int)rn[i][1]; *run = (int)rn[i][2]; return 0; }
	for (int m = 0; m < 256; m++) if ((window >> (32 - ml[m])) == mc[m]) {
		*size = ml[m];
		if (m == 0) *run = 1; else *mag = t->mag_expand[m];
		return 0;
	}
	return -1;
}

// ------------------------------------------------------------------------------------------
// Geometry
// ------------------------------------------------------------------------------------------
bool build_frame_plan(FramePlan *plan, int width, int height, int pixel_kind, int encoded_format)
{
	memset(plan, 0, sizeof(*plan));
	// Dimensions come from untrusted sample headers (16-bit fields): beyond kMaxFrameDim the 32-bit band offsets below could wrap and a
	// crafted sample would make the decoder write outside its pyramid.  16384 x 16384 x 4 channels x 21/16 = 1.4 G elements fits.
	if (width <= 0 || height <= 0 || width > kMaxFrameDim || height > kMaxFrameDim) return false;
	plan->display_height = height;
	plan->encoded_format = encoded_format;
	plan->pixel_kind = pixel_kind;
	int chroma_width;
	switch (encoded_format) {
	case ENC_YUV422:   plan->num_channels = 3; plan->precision = 10; chroma_width = width / 2; break;
	case ENC_RGB444:   plan->num_channels = 3; plan->precision = 12; chroma_width = width; break;
	case ENC_RGBA4444: plan->num_channels = 4; plan->precision = 12; chroma_width = width; break;
	case ENC_BAYER:    plan->num_channels = 4; plan->precision = 12; width /= 2; height /= 2; chroma_width = width;
	                   plan->display_height = height; break;
	default: return false;
	}
	// Height is rounded up to a multiple of 8 (Codec/encoder.c:1569-1571, :2236-2238).
	int enc_height = (height + 7) / 8 * 8;
	plan->width = width; plan->height = enc_height;
	// Every level must halve evenly: widths divisible by 8 per channel (IsFrameTransformable).
	if ((chroma_width % 8) != 0 || (width % 8) != 0) return false;

	// Final (entropy coded) bands first, then the two intermediate LL planes of every channel.
	uint64_t off = 0;
	auto place = [&](BandDesc &b, int w, int h) {
		b.width = w; b.height = h; b.pitch = align_up(w, 8); b.offset = (uint32_t)off; b.quant = 1; b.scale = 1;
		uint64_t elems = (uint64_t)b.pitch * (uint64_t)h;
		off += (elems + 63u) & ~(uint64_t)63;      // 128-byte aligned bands
	};
	for (int c = 0; c < plan->num_channels; c++) {
		ChannelPlan &cp = plan->ch[c];
		cp.width = c == 0 ? width : chroma_width; cp.height = enc_height;
		place(cp.band[2][0], cp.width >> 3, cp.height >> 3);                       // LL of level 3
		for (int lv = 2; lv >= 0; lv--)
			for (int b = 1; b < 4; b++) place(cp.band[lv][b], cp.width >> (lv + 1), cp.height >> (lv + 1));
	}
	plan->final_elems = (uint32_t)off;
	for (int c = 0; c < plan->num_channels; c++) {
		ChannelPlan &cp = plan->ch[c];
		place(cp.band[0][0], cp.width >> 1, cp.height >> 1);
		place(cp.band[1][0], cp.width >> 2, cp.height >> 2);
	}
	if (off >= ((uint64_t)1 << 31)) return false;      // element offsets and raster indices stay inside 31 bits everywhere
	plan->coeff_elems = (uint32_t)off;
	return true;
}

// BYR4_LOGTABLE (frame.c:5228) with CURVE_LIN2LOG = lin2log() (Common/AVIExtendedHeader.h:124,153): the float / double mix is the reference's, so
// the truncation to int falls on the same side for every entry.
void build_bayer_log90_curve(int precision, uint16_t *curve)
{
	// Every intermediate is rounded exactly where the reference (gcc, plain IEEE sequence) rounds it: no fused multiply-add, no
	// reciprocal, the quotient narrowed to float before the scaling.  `volatile` pins the sequence whatever the host compiler's
	// floating-point contraction default is (hipcc's differs from gcc's; two of the 16384 entries depend on it).
	const int max_value = 1 << kBayerCurveBits;
	const volatile double base_m1 = 89.0, log_base = log10(90.0);
	const volatile float top = (float)((1 << precision) - 1);
	curve[0] = 0;
	for (int i = 1; i < max_value; i++) {
		volatile float x = (float)i / (float)max_value;
		volatile double prod = (double)x * base_m1;
		volatile double arg = prod + 1.0;
		volatile double lg = log10(arg);
		volatile double quot = lg / log_base;
		volatile float y = (float)quot;                    // lin2log() returns float (AVIExtendedHeader.h:153-156)
		volatile float scaled = y * top;
		curve[i] = (uint16_t)(int)scaled;
	}
}

// BYR4LinearRestore (decoder.c:10714-10783) for a sample without encode-curve metadata: log base 90 undone, 14-bit index -> 16-bit linear value;
// log2lin() (Common/AVIExtendedHeader.h:148) evaluates pow() in double and returns float, the scaling by 65535 is a float product.
void build_bayer_linear_restore_curve(uint16_t *curve)
{
	const volatile float base = 90.0f;
	for (int j = 0; j < (1 << kBayerCurveBits); j++) {
		volatile float i = (float)j / 16384.0f;
		volatile double p = pow((double)base, (double)i);
		volatile double num = p - 1.0;
		volatile double den = (double)base - 1.0;
		volatile double quot = num / den;
		volatile float lin = (float)quot;
		volatile float scaled = lin * 65535.0f;
		int val = (int)scaled;
		curve[j] = (uint16_t)(val < 0 ? 0 : (val > 65535 ? 65535 : val));
	}
}

// ------------------------------------------------------------------------------------------
// Quantizer
// ------------------------------------------------------------------------------------------
namespace {
const int kLumaQ[4][17] = {
	{4, 4,5,5, 4,5,5, 9,8,8,8, 4,4,4, 4,4,4},
	{4, 8,8,12, 8,8,12, 9,12,12,16, 32,32,48, 32,32,48},
	{4, 6,6,8, 6,6,8, 5,8,8,12, 16,16,24, 16,16,24},
	{4, 4,4,6, 4,4,6, 5,8,8,8, 8,8,12, 8,8,12},
};
const int kChromaQ[4][17] = {
	{4, 4,5,5, 4,5,5, 9,8,8,8, 8,8,8, 8,8,8},
	{4, 8,8,12, 8,8,12, 9,12,12,16, 32,32,48, 32,32,48},
	{4, 6,6,8, 6,6,8, 5,8,8,12, 16,16,32, 16,16,32},
	{4, 6,6,8, 6,6,8, 5,8,8,8, 8,8,16, 8,8,16},
};
}

// The subband divisors before they are dealt to the bands of a wavelet tree (QuantizationSetQuality quantize.c:186 up to the intra remap at :548):
// out[0] luma, [1] chroma, [2] luma under the bit-rate limiter, [3] chroma under it -- 17 entries each, the numbering of the 2-frame group.
static void derive_subband_tables(FramePlan *plan, int quality, bool progressive, QuantState *st, int out[4][17], int *factor_out, int *new_quality_out);

// The bit-rate limiter of SetTransformQuantization (quantize.c:2896-2906 the rate of the previous key sample, :2994-3100 the limiter): active only for qualities
// <= HIGH on <= 1080p 3-channel YUV.  One call per channel; channel 0 moves the state.
static int bitrate_of_previous_sample(const QuantState *st, float framerate, int gop_length)
{
	const float fr = (framerate > 10.0f && framerate < 120.0f) ? framerate : 30.0f;
	return (int)((float)(int32_t)st->lastgopbitcount * fr / (float)gop_length);
}
static bool bitrate_limiter_applies(int fixedQuality, int newQuality, int width, int height, int num_channels, int encoded_format)
{
	return fixedQuality != 0 && !(width > 1920 || height > 1080 || num_channels > 3 || newQuality > 3 || encoded_format == ENC_RGB444);
}
static void limit_bitrate(int fixedQuality, int currentbitrate, bool progressive, int c, QuantState *st, int quant[17], const int quantMAX[17])
{
	const int BR_LIMIT = 130000000, BR_STEPS = 10000000;
	const int upper = fixedQuality == 1 ? BR_LIMIT - 2 * BR_STEPS : (fixedQuality == 3 ? BR_LIMIT + 2 * BR_STEPS : BR_LIMIT);
	if (currentbitrate > upper) {
		memcpy(quant, quantMAX, sizeof(int) * 17);
		if (c == 0) {
			if (st->overbitrate == 0) st->overbitrate = 1;
			if (currentbitrate > upper * 12 / 10) st->overbitrate++;
			if (st->overbitrate > 16) st->overbitrate = 16;
		}
	} else if (st->overbitrate > 0) {
		if (c == 0) {
			if (st->overbitrate > 1 && currentbitrate < upper) st->overbitrate--;
			else if (st->overbitrate == 1 && currentbitrate < upper * 8 / 10) st->overbitrate = 0;
		}
		if (st->overbitrate > 0) memcpy(quant, quantMAX, sizeof(int) * 17);
	}
	if (st->overbitrate > 1) {
		const int rc = st->overbitrate - 1;
		if (progressive) { for (int i = 11; i < 17; i++) quant[i] = (quant[i] * (rc + 4)) >> 2; }
		else {
			for (int i : {11, 14}) quant[i] = (quant[i] * (rc + 4)) >> 2;
			for (int i : {12, 15, 13, 16}) quant[i] = (quant[i] * (rc / 8 + 4)) >> 2;
		}
	}
}

void derive_quantization(FramePlan *plan, int quality, bool progressive, float framerate, QuantState *st)
{
	int tabs[4][17], factor, newQuality;
	derive_subband_tables(plan, quality, progressive, st, tabs, &factor, &newQuality);
	int *qL = tabs[0], *qC = tabs[1], *qLmax = tabs[2], *qCmax = tabs[3];
	const int precision = plan->precision;
	const int mpq = plan->midpoint_prequant;
	(void)precision;
	// an intra frame has no temporal wavelet: its level-2 bands take the divisors of the group's frame wavelets (quantize.c:548-565)
	for (int i = 0; i < 3; i++) { qL[7 + i] = qL[11 + i]; qC[7 + i] = qC[11 + i]; qLmax[7 + i] = qLmax[11 + i]; qCmax[7 + i] = qCmax[11 + i]; }
	const int fixedQuality = factor;     // 0 => bitrate mode (not supported: treated as quality 3 tables w/o VBR)

	plan->prescale[0] = 0; plan->prescale[1] = precision >= 10 ? 2 : 0; plan->prescale[2] = precision == 12 ? 2 : 0;

	const int currentbitrate = bitrate_of_previous_sample(st, framerate, 1);
	const bool limiter_on = bitrate_limiter_applies(fixedQuality, newQuality, plan->width, plan->height, plan->num_channels, plan->encoded_format);
	if (st->overbitrate < 0 || st->overbitrate > 16) st->overbitrate = 0;

	for (int c = 0; c < plan->num_channels; c++) {
		int quant[17], quantMAX[17];
		memcpy(quant, c ? qC : qL, sizeof(quant)); memcpy(quantMAX, c ? qCmax : qLmax, sizeof(quantMAX));
		if (limiter_on) limit_bitrate(fixedQuality, currentbitrate, progressive, c, st, quant, quantMAX);
		ChannelPlan &cp = plan->ch[c];
		int scale[3][4] = {{4, 2, 2, 1}};
		for (int k = 1; k < 3; k++) { int s = scale[k - 1][0]; scale[k][0] = 4 * s; scale[k][1] = 2 * s; scale[k][2] = 2 * s; scale[k][3] = s; }
		for (int k = 0; k < 3; k++) for (int b = 0; b < 4; b++) { cp.band[k][b].scale = scale[k][b]; }
		int subband = 1;
		auto midpoint = [&](int q) { if (mpq) { q *= mpq; q /= (mpq - 1) * 2; } else q /= 2; return q; };
		for (int index = 2; index >= 1; index--) {
			cp.band[index][0].quant = 1;
			for (int b = 1; b < 4; b++, subband++) {
				int vscale = (quantMAX[subband] - quant[subband]) * 256 - 256 * quantMAX[subband] + 512 * quant[subband];
				int q = ((vscale * scale[index][b]) >> 8) >> 2;
				if (!(quality & 0x10000000)) q = midpoint(q);
				cp.band[index][b].quant = q;
			}
		}
		cp.band[0][0].quant = 1;
		for (int b = 1; b < 4; b++, subband++) {
			int vscale = (quantMAX[subband] - quant[subband]) * 256 - 256 * quantMAX[subband] + 512 * quant[subband];
			cp.band[0][b].quant = midpoint(vscale >> 8);
		}
	}
}

static void derive_subband_tables(FramePlan *plan, int quality, bool progressive, QuantState *st, int out[4][17], int *factor_out, int *new_quality_out)
{
	int *qL = out[0], *qC = out[1], *qLmax = out[2], *qCmax = out[3];
	// Bayer input: the encoder pins the RGB quality bits before it derives the tables ("prevent increased quant on channels 1-3",
	// encoder.c:2638); the sample header keeps the caller's quality word
	if (plan->encoded_format == ENC_BAYER) quality |= 3 << 25;
	// encoder.c:1141 SetEncoderQuantization: ChromaFullRes = (input colour format >= COLOR_FORMAT_BAYER (100)): true for RG48 (120) and
	// BYR4 (104), false for the packed 4:2:2 formats and -- although it is a 4:4:4:4 format -- for b64a (COLOR_FORMAT_BGRA64 = 30),
	// whose R, B and A planes therefore get the chroma tables
	const bool chroma_full = plan->pixel_kind == PIX_RG48 || plan->pixel_kind == PIX_BYR4 || plan->pixel_kind == PIX_BYR5 || plan->pixel_kind == PIX_RG64 || plan->pixel_kind == PIX_RG24 || plan->pixel_kind == PIX_BGRA || plan->pixel_kind == PIX_BGRa || (plan->pixel_kind >= PIX_R210 && plan->pixel_kind <= PIX_AR10);
	// (deep RGB encoded as 4:2:2 is converted first and quantized as the 4:2:2 frame it has become: encoder.c:2339-2440 hand the converted format on)
	const bool chroma_full_res = chroma_full && plan->encoded_format != ENC_YUV422;
	const int precision = plan->precision;
	int factor = quality & 0xff;
	const int detail = (quality & 0x0e0000) >> 17;
	int rgb_quality = (quality & 0x06000000) >> 25;
	if (rgb_quality > 2) rgb_quality = 2;
	int mpq = detail + 2; if (mpq > 8) mpq = 0;
	plan->midpoint_prequant = mpq;
	if (quality & 0x1f00) factor = 5;
	const int newQuality = factor;
	// FSratelimiter is seeded once (frame == NULL in the reference's init call, quantize.c:228-238) and
	// afterwards only moves for FILMSCAN2/3 rate feedback, which needs the previous sample size.
	if (st->FSratelimiter < 0) st->FSratelimiter = (newQuality == 5) ? 8 : (newQuality == 6 ? 4 : 0);
	if (newQuality >= 5 && st->lastgopbitcount && !(quality & 0x1f00)) {
		float gop_size = (float)(int32_t)(st->lastgopbitcount >> 3);
		float compression = (float)(plan->width * plan->height * plan->num_channels * precision / 8) / gop_size;
		if (!chroma_full_res) compression /= 1.5f;
		int &r = st->FSratelimiter;
		if (newQuality == 5) {
			if (compression > 5.5f) { r--; if (compression > 6.5f) r--; if (compression > 7.5f) r -= 2; }
			else if (compression < 4.0f) { r++; if (compression < 3.5f) r++; if (compression < 3.0f) r++; if (compression < 2.5f) r++; if (compression < 2.0f) r++; if (compression < 1.5f) r += 2; }
		} else if (newQuality == 10) {
			if (compression > 2.5f) r--; else if (compression < 2.0f) { r++; if (compression < 1.5f) r += 2; }
		} else {
			if (compression > 4.5f) { r--; if (compression > 5.5f) r--; if (compression > 6.5f) r -= 2; }
			else if (compression < 3.0f) { r++; if (compression < 2.5f) r++; if (compression < 2.0f) r++; if (compression < 1.5f) r += 2; }
		}
		if (r < 0) r = 0; if (r > 20) r = 20;
	}
	if (factor < 1 || factor > 10) factor = 0;
	if (factor > 3) factor = 3;
	int overrate = factor; if (overrate >= 2) overrate--;
	for (int i = 0; i < 17; i++) {
		qL[i] = kLumaQ[factor][i]; qLmax[i] = kLumaQ[overrate][i];
		qC[i] = chroma_full_res ? kLumaQ[factor][i] : kChromaQ[factor][i];
		qCmax[i] = chroma_full_res ? kLumaQ[overrate][i] : kChromaQ[overrate][i];
	}
	for (int i = 0; i < 17; i++) { qLmax[i] = qL[i] + (qLmax[i] - qL[i]) / 2; qCmax[i] = qC[i] + (qCmax[i] - qC[i]) / 2; }
	int lowfreqquant = 4;
	if (precision >= 10) {
		int scale = 64, limiter = std::min(st->FSratelimiter, 16);
		if (newQuality == 4) { lowfreqquant = 3; scale = 48; }
		else if (newQuality >= 5 && newQuality <= 10) { lowfreqquant = 2; scale = 16 + limiter * 2; }
		if (newQuality >= 5 && scale >= 4) scale >>= 1;
		if (newQuality == 10 && scale >= 6) { scale *= 2; scale /= 3; }
		if (newQuality >= 4) for (int i = 1; i < 7; i++) qL[i] = qC[i] = qLmax[i] = qCmax[i] = lowfreqquant;
		for (int i = 8; i < 17; i++) {
			qL[i] = std::max((qL[i] * scale) >> 4, 2); qC[i] = std::max((qC[i] * scale) >> 4, 2);
			qLmax[i] = std::max((qLmax[i] * 64) >> 4, 2); qCmax[i] = std::max((qCmax[i] * 64) >> 4, 2);
		}
		qL[7] = qC[7] = qLmax[7] = qCmax[7] = 4;
	}
	if (precision == 12) {
		int chromagain = rgb_quality == 0 ? 8 : (rgb_quality == 1 ? 6 : 4);
		if (newQuality >= 4) for (int i = 1; i < 7; i++) qL[i] = qC[i] = qLmax[i] = qCmax[i] = lowfreqquant;
		for (int i = 4; i < 7; i++) { qL[i] *= 4; qC[i] *= 4; qLmax[i] *= 4; qCmax[i] *= 4; }
		if (st->FSratelimiter > 16) chromagain = std::min(chromagain + st->FSratelimiter - 16, 8);
		for (int i = 11; i < 17; i++) { qL[i] *= 4; qC[i] *= chromagain; qLmax[i] *= 4; qCmax[i] *= chromagain; }
	}
	if (!progressive) {
		if (factor == 2) for (int i : {12, 13, 15, 16}) { qLmax[i] = qL[i]; qCmax[i] = qC[i]; }
		for (int *a : {qL, qC, qLmax, qCmax}) { a[11] = a[11] * 3 / 2; a[12] = a[12] * 2 / 3; a[14] = a[14] * 3 / 2; a[15] = a[15] * 2 / 3; }
	}
	*factor_out = factor; *new_quality_out = newQuality;
}

// (cfhd_gop.cpp) the divisors of a progressive two-frame group, per channel and subband.  Every CFHD_EncodeSample call of a group runs QuantizationSetQuality
// (encoder.c:2880: the FILMSCAN2/3 limiter moves with the size of the last key sample); only the call that opens a group (deal) also runs SetTransformQuantization
// (encoder.c:2895-2905, group.count == 0: the bit-rate limiter, its rate = bits * fps / 2) and hands the divisors to the wavelets.
void derive_gop_subband_divisors(FramePlan *plan, int quality, bool progressive, float framerate, QuantState *st, bool deal, int out[3][17])
{
	int tabs[4][17], factor, newQuality;
	derive_subband_tables(plan, quality, progressive, st, tabs, &factor, &newQuality);
	if (!deal) return;
	const int currentbitrate = bitrate_of_previous_sample(st, framerate, 2);
	const bool limiter_on = bitrate_limiter_applies(factor, newQuality, plan->width, plan->height, plan->num_channels, plan->encoded_format);
	if (st->overbitrate < 0 || st->overbitrate > 16) st->overbitrate = 0;
	for (int c = 0; c < 3; c++) {
		memcpy(out[c], tabs[c ? 1 : 0], sizeof(int) * 17);
		if (limiter_on) limit_bitrate(factor, currentbitrate, progressive, c, st, out[c], tabs[c ? 3 : 2]);
	}
}

int unit_device(int i, int ndevices, const char *pinned_env, const char *list_env)
{
	if (ndevices < 1) ndevices = 1;
	if (i < 0) i = 0;
	if (list_env && *list_env) {
		int list[64], n = 0;
		for (const char *p = list_env; *p && n < 64;) {
			while (*p == ',' || *p == ' ') p++;
			if (*p < '0' || *p > '9') break;
			int v = 0;
			while (*p >= '0' && *p <= '9') v = v * 10 + (*p++ - '0');
			list[n++] = v % ndevices;
		}
		if (n) return list[i % n];
	}
	if (pinned_env && *pinned_env) return -1;
	return i % ndevices;
}

} // namespace cfhd
