// cfhd_gop.cpp -- see cfhd_gop.h: plan, quantizer tables and sample syntax of the two-frame group (host code).
#include "cfhd_gop.h"
#include <string.h>

namespace cfhd {

void derive_gop_subband_divisors(FramePlan *plan, int quality, bool progressive, float framerate, QuantState *st, bool deal, int out[3][17]);   // cfhd_tables.cpp

bool build_gop_plan(GopPlan *plan, int width, int height, int pixel_kind, bool interlaced)
{
	plan->interlaced = interlaced;
	if (width <= 0 || height <= 0 || width > kMaxFrameDim || height > kMaxFrameDim) return false;
	if (pixel_kind != PIX_YUY2 && pixel_kind != PIX_2VUY) return false;              // CFHD_ENCODING_FLAGS_YUV_2FRAME_GOP: "YUV 4:2:2 only" (CFHDTypes.h:254)
	const int enc_height = (height + 7) & ~7;                                        // encoder.c:1569-1571
	// the chroma planes halve once per level on whole pairs: level 1, the two levels on the temporal lowpass band
	if ((width / 2) % 8 || width % 16) return false;
	plan->width = width; plan->height = enc_height; plan->display_height = height;
	plan->num_channels = 3; plan->precision = 10; plan->pixel_kind = pixel_kind;
	size_t at = 0;
	auto pitch_of = [](int w) { return (w + 7) & ~7; };
	for (int c = 0; c < 3; c++) {
		GopChannel &ch = plan->ch[c];
		ch.width = c == 0 ? width : width / 2; ch.height = enc_height;
		// type, level, bands, divisor of the channel's dimensions
		static const int type[kGopWavelets] = { 5, 5, 4, 3, 3, 3 }, level[kGopWavelets] = { 1, 1, 2, 3, 3, 4 }, nb[kGopWavelets] = { 4, 4, 2, 4, 4, 4 }, div[kGopWavelets] = { 2, 2, 2, 4, 4, 8 };
		for (int k = 0; k < kGopWavelets; k++) {
			GopWavelet &w = ch.w[k];
			w.type = type[k]; w.level = level[k]; w.nbands = nb[k];
			if (ch.width % div[k] || ch.height % div[k]) return false;
			w.width = ch.width / div[k]; w.height = ch.height / div[k]; w.pitch = pitch_of(w.width);
			for (int b = 0; b < 4; b++) {
				w.offset[b] = at; w.quant[b] = 1; w.scale[b] = 0;
				if (b < w.nbands) at += ((size_t)w.pitch * w.height + 63) & ~(size_t)63;      // bands start on 128-byte boundaries
			}
			w.prescale = 0;
		}
		ch.w[4].prescale = 2;                                                           // 10-bit FIELDPLUS: {0, 0, 0, 0, 2, 0, 0, 0} (wavelet.c:1745)
		// wavelet.c:7142 SetTransformScale, TRANSFORM_TYPE_FIELDPLUS
		const int frame_scale[4] = { 4, 2, 2, 1 };
		for (int k = 0; k < 2; k++) for (int b = 0; b < 4; b++) ch.w[k].scale[b] = frame_scale[b];
		ch.w[2].scale[0] = 2 * frame_scale[0]; ch.w[2].scale[1] = frame_scale[0];
		auto spatial = [](GopWavelet &w, int s) { w.scale[0] = 4 * s; w.scale[1] = 2 * s; w.scale[2] = 2 * s; w.scale[3] = s; };
		spatial(ch.w[3], ch.w[2].scale[1]);
		spatial(ch.w[4], ch.w[2].scale[0]);
		spatial(ch.w[5], ch.w[4].scale[0]);
	}
	plan->coeff_elems = at;
	plan->sample_buffer_bytes = (size_t)width * height * 4 + 65536;      // SampleEncoder.cpp:387 (PixelSize of YUY2 / 2vuy: 4, :1232)
	return true;
}

bool derive_gop_quantization(GopPlan *plan, int quality, QuantState *st, float framerate, bool deal)
{
	FramePlan fp;
	if (!build_frame_plan(&fp, plan->width, plan->display_height, plan->pixel_kind, ENC_YUV422)) return false;
	int tabs[3][17];
	derive_gop_subband_divisors(&fp, quality, !plan->interlaced, framerate, st, deal, tabs);
	if (!deal) return true;
	plan->midpoint_prequant = fp.midpoint_prequant;
	const int mpq = plan->midpoint_prequant;
	auto midpoint = [&](int q) { if (mpq) { q *= mpq; q /= (mpq - 1) * 2; } else q /= 2; return q; };
	for (int c = 0; c < 3; c++) {
		const int *quant = tabs[c];
		GopChannel &ch = plan->ch[c];
		int subband = 1;
		// quantize.c:3480: the two spatial wavelets on the temporal lowpass band, top first (VSCALE(q, qmax, 256) = 256 q; quantScaleFactor 2)
		for (int k = 5; k >= 4; k--) {
			ch.w[k].quant[0] = 1;
			for (int b = 1; b < 4; b++, subband++) {
				int q = ((quant[subband] * 256 * ch.w[k].scale[b]) >> 8) >> 2;
				if (!(quality & 0x10000000)) q = midpoint(q);
				ch.w[k].quant[b] = q;
			}
		}
		subband++;                                                                    // subband 7: the lowpass band of w[3], kept at 1 for 10-bit sources (encoder.c:8538)
		ch.w[3].quant[0] = 1;
		for (int b = 1; b < 4; b++, subband++) {
			int q = ((quant[subband] * 256 * ch.w[3].scale[b]) >> 8) >> 2;
			if (!(quality & 0x10000000)) q = midpoint(q);
			ch.w[3].quant[b] = q;
		}
		ch.w[2].quant[0] = ch.w[2].quant[1] = 1;
		for (int k = 1; k >= 0; k--) {                                                // the frame wavelets: frame 1 first (subbands 11-13), then frame 0
			ch.w[k].quant[0] = 1;
			for (int b = 1; b < 4; b++, subband++) ch.w[k].quant[b] = midpoint((quant[subband] * 256) >> 8);
		}
	}
	return true;
}

// ------------------------------------------------------------------------------------------
// Writer
// ------------------------------------------------------------------------------------------
namespace {

// The walk over a group sample's syntax is written once against a sink (as cfhd_bitstream.cpp walk_sample): the host sink codes the payloads inline, the
// template sink records them as holes the GPU entropy stage fills.
struct GroupHostSink {
	BitWriter w; const GopPlan &plan; const int16_t *coeffs;
	size_t index_at = 0, channel_start = 0;
	std::vector<int16_t> zeros;
	GroupHostSink(uint8_t *out, size_t cap, const GopPlan &p, const int16_t *c) : w(out, cap), plan(p), coeffs(c) {}
	void tag(int t, int v) { w.put_tag(t, v); }
	void tag_opt(int t, int v) { w.put_tag_opt(t, v); }
	void bytes(const void *p, size_t n) { w.put_bytes(p, n); }
	void push(int t) { w.size_push(t); }
	void pop() { w.size_pop(); }
	void index_entries(int n) { index_at = w.bytes(); for (int i = 0; i < n; i++) w.put_tag(TAG_ENTRY, i); }
	void channel_begin(int) { channel_start = w.bytes(); }
	void channel_end(int c) { w.patch32(index_at + 4 * (size_t)c, (uint32_t)(w.bytes() - channel_start)); }      // the channel's size in the index (encoder.c:7828: bytes, big-endian)
	// 16-bit big-endian coefficients, row after row without the pitch padding (encoder.c:4423 lowpass, :5850 EncodeQuant16s)
	void raw16(int c, int k)
	{
		const GopWavelet &wv = plan.ch[c].w[k];
		const int16_t *band = coeffs + wv.offset[0];
		for (int r = 0; r < wv.height; r++)
			for (int x = 0; x < wv.width; x++) w.put_bits((uint16_t)band[(size_t)r * wv.pitch + x], 16);
		w.pad32();
	}
	void band(int c, int k, int b)
	{
		const GopWavelet &wv = plan.ch[c].w[k];
		const int codebook = gop_band_is_difference_coded(plan, k, b) ? 2 : 1;
		// "only compress up to 80% of the frame size" (encoder.c:8332): once the sample fills more than that of the caller's sample buffer, the remaining
		// bands of the frame wavelets are coded as zeros (EncodeZeroBand encoder.c:6220: the same header, one run over the whole band)
		if (k < 2 && zero_band) {
			if (zeros.size() < (size_t)wv.pitch * wv.height) zeros.assign((size_t)wv.pitch * wv.height, 0);
			vlc_encode_band(w, zeros.data(), wv.width, wv.height, wv.pitch, codebook, wv.quant[b]);
		} else vlc_encode_band(w, coeffs + wv.offset[b], wv.width, wv.height, wv.pitch, codebook, wv.quant[b], collect_peaks ? &peaks : nullptr);
	}
	// the three optional tags in front of a peak-coded band's size chunk and the table behind the band (cfhd_bitstream.cpp HostSink: codec.c:1804-1809, encoder.c:6543)
	std::vector<int16_t> peaks; size_t peak_tags_at = 0; bool collect_peaks = false;
	void peak_tags() { peak_tags_at = w.bytes(); w.put_tag_opt(TAG_PEAK_TABLE_OFFSET_L, 0); w.put_tag_opt(TAG_PEAK_TABLE_OFFSET_H, 0); w.put_tag_opt(TAG_PEAK_LEVEL, 0); peaks.clear(); collect_peaks = true; }
	void peak_table(int quant)
	{
		collect_peaks = false;
		if (peaks.empty()) return;
		if ((peaks.size() + 1) / 2 > 0xffff) return;        // more longwords than the chunk header can count (MAX_CHUNK_SIZE, codec.h:195): the reference writes no table and leaves the three tags zero (encoder.c:6557)
		const uint32_t offset = (uint32_t)(w.bytes() - peak_tags_at);
		auto tagword = [](int tag, uint32_t value) { return ((uint32_t)(uint16_t)(int16_t)(-tag) << 16) | (value & 0xffffu); };
		w.patch32(peak_tags_at, tagword(TAG_PEAK_TABLE_OFFSET_L, offset & 0xffffu));
		w.patch32(peak_tags_at + 4, tagword(TAG_PEAK_TABLE_OFFSET_H, offset >> 16));
		w.patch32(peak_tags_at + 8, tagword(TAG_PEAK_LEVEL, (uint32_t)(kPeakThreshold * quant)));
		if (peaks.size() & 1) peaks.push_back(0);
		w.put_tag_opt(TAG_PEAK_TABLE, (int)(peaks.size() / 2));
		w.put_bytes(peaks.data(), peaks.size() * 2);
	}
	bool zero_band = false;
	void frame_band_begins() { zero_band = (uint64_t)w.bytes() * 100 > (uint64_t)plan.sample_buffer_bytes * 80; }      // (asked in front of the band's header, as the reference does)
};

struct GroupTemplateSink : TemplateRecorder {
	const GopPlan &plan;
	GroupTemplateSink(SampleTemplate &tt, const GopPlan &p) : TemplateRecorder(tt), plan(p) {}
	// holes: `level` is the wavelet index (0 .. 5), band 0 of wavelets 5 and 3 are raw 16-bit words
	void raw16(int c, int k) { const GopWavelet &wv = plan.ch[c].w[k]; hole(0, c, k, 0, (((wv.width * wv.height * 2) + 3) / 4) * 4); }
	void band(int c, int k, int b) { hole(1, c, k, b, 0); }
	void frame_band_begins() {}          // (data dependent: the caller checks the finished sample against gop_sample_may_zero_bands())
};

template <typename Sink> void put_band_header(Sink &w, int band, const GopWavelet &wv, int subband, int encoding, bool difference = false)
{
	w.tag(TAG_MARKER, MARK_BAND_START);
	w.tag(TAG_BAND_NUMBER, band);
	w.tag(TAG_BAND_CODING_FLAGS, difference ? 2 + 16 : 1);      // code set 17, no difference coding (encoder.c:6120 SetCodingFlags for a progressive group); interlaced: subbands 12 / 15 in code set 18, difference coded
	w.tag(TAG_BAND_WIDTH, wv.width);
	w.tag(TAG_BAND_HEIGHT, wv.height);
	w.tag(TAG_BAND_SUBBAND, subband);
	w.tag(TAG_BAND_ENCODING, encoding);
	w.tag(TAG_BAND_QUANTIZATION, wv.quant[band]);
	w.tag(TAG_BAND_SCALE, wv.scale[band]);
	if (difference) w.peak_tags();
	w.push(TAG_SUBBAND_SIZE);
	w.tag(TAG_BAND_HEADER, 0);
}
template <typename Sink> void put_wavelet_header(Sink &w, const GopWavelet &wv, int number)
{
	w.tag(TAG_MARKER, MARK_HIGHPASS_START);
	w.tag(TAG_WAVELET_TYPE, wv.type);
	w.tag(TAG_WAVELET_NUMBER, number);
	w.tag(TAG_WAVELET_LEVEL, wv.level);
	w.tag(TAG_NUM_BANDS, wv.nbands);
	w.tag(TAG_HIGHPASS_WIDTH, wv.width);
	w.tag(TAG_HIGHPASS_HEIGHT, wv.height);
	w.tag(TAG_LOWPASS_BORDER, 0);
	w.tag(TAG_HIGHPASS_BORDER, 0);
	w.tag(TAG_LOWPASS_SCALE, wv.scale[0]);
	w.tag(TAG_LOWPASS_DIVISOR, 0);
	w.push(TAG_LEVEL_SIZE);
}

template <typename Sink> void walk_group_sample(Sink &w, const GopPlan &plan, const SampleHeaderInfo &hdr)
{
	const int nch = plan.num_channels;
	// --- PutVideoGroupHeader (codec.c:835) ---
	w.tag(TAG_SAMPLE, 2);                                // SAMPLE_TYPE_GROUP
	w.tag(TAG_INDEX, nch);
	w.index_entries(nch);
	w.tag(TAG_TRANSFORM_TYPE, 2);                        // TRANSFORM_TYPE_FIELDPLUS
	w.tag(TAG_NUM_FRAMES, 2);
	w.tag(TAG_NUM_CHANNELS, nch);
	w.tag_opt(TAG_INPUT_FORMAT, hdr.input_format);
	{ const int cs = hdr.color_space & ~4; if (cs) w.tag_opt(TAG_ENCODED_COLORSPACE, cs); }
	w.tag(TAG_NUM_WAVELETS, kGopWavelets);
	w.tag(TAG_NUM_SUBBANDS, kGopSubbands);
	w.tag(TAG_NUM_SPATIAL, 3);
	w.tag(TAG_FIRST_WAVELET, 3);
	w.tag(TAG_FRAME_WIDTH, plan.width);
	w.tag(TAG_FRAME_HEIGHT, plan.height);
	w.tag_opt(TAG_FRAME_NUMBER, (int)(hdr.frame_number & 0xffff));
	w.tag(TAG_PRECISION, plan.precision);
	w.tag_opt(TAG_FRAME_DISPLAY_HEIGHT, plan.display_height);
	w.tag_opt(TAG_VERSION, (10 << 12) | (1 << 8) | 0);
	w.tag_opt(TAG_QUALITY_L, hdr.encoder_quality & 0xffff);
	w.tag_opt(TAG_QUALITY_H, (hdr.encoder_quality >> 16) & 0xffff);
	{
		unsigned table = 0;
		for (int k = 0; k < kGopWavelets; k++) table += (unsigned)plan.ch[0].w[k].prescale << (14 - 2 * k);
		w.tag_opt(TAG_PRESCALE_TABLE, (int)table);         // the decoder's built-in FIELDPLUS default: optional (codec.c:1040)
	}
	if (hdr.channel_number_tag) w.tag_opt(TAG_ENCODED_CHANNEL_NUMBER, 0);
	// --- EncodeQuantizedGroup (encoder.c:7559-7620) ---
	w.push(TAG_SAMPLE_SIZE);
	auto put_metadata = [&](const uint8_t *block, size_t size) {
		if (!block || !size) return;
		w.tag_opt(TAG_METADATA, (int)(size >> 2));
		w.bytes(block, size);
	};
	put_metadata(hdr.meta_global, hdr.meta_global_size);
	put_metadata(hdr.meta_local, hdr.meta_local_size);
	{
		uint8_t freespace[512];
		memset(freespace, 0, sizeof(freespace));
		memcpy(freespace, "FREE", 4);
		freespace[4] = (uint8_t)(504 & 0xff); freespace[5] = (uint8_t)(504 >> 8);
		put_metadata(freespace, sizeof(freespace));
	}
	w.tag_opt(TAG_INTERLACED_FLAGS, 0);
	w.tag_opt(TAG_PROTECTION_FLAGS, 0);
	w.tag_opt(TAG_PICTURE_ASPECT_X, 16);
	w.tag_opt(TAG_PICTURE_ASPECT_Y, 9);
	if (hdr.progressive) w.tag(TAG_SAMPLE_FLAGS, 1);
	// the band end code of code set 17, padded to a whole word: what FinishEncodeBand leaves behind the raw words of a 16-bit band
	uint8_t band_end[8]; size_t band_end_bytes;
	{ BitWriter e(band_end, sizeof(band_end)); vlc_encode_band(e, nullptr, 0, 0, 0, 1, 1); band_end_bytes = e.bytes(); }

	for (int c = 0; c < nch; c++) {
		const GopChannel &ch = plan.ch[c];
		if (c > 0) { w.tag(TAG_SAMPLE, SAMPLE_TYPE_CHANNEL); w.tag(TAG_CHANNEL, c); }
		w.channel_begin(c);
		// --- the sample's lowpass band: w[5]'s (encoder.c:4251) ---
		const GopWavelet &top = ch.w[5];
		w.tag(TAG_MARKER, MARK_LOWPASS_START);
		w.tag(TAG_LOWPASS_SUBBAND, 0);
		w.tag(TAG_NUM_LEVELS, 4);
		w.tag(TAG_LOWPASS_WIDTH, top.width);
		w.tag(TAG_LOWPASS_HEIGHT, top.height);
		w.tag(TAG_MARGIN_LEFT, 0); w.tag(TAG_MARGIN_TOP, 0); w.tag(TAG_MARGIN_RIGHT, 0); w.tag(TAG_MARGIN_BOTTOM, 0);
		w.tag(TAG_PIXEL_OFFSET, 0);
		w.tag(TAG_QUANTIZATION, 1);
		w.tag(TAG_PIXEL_DEPTH, 16);
		w.push(TAG_SUBBAND_SIZE);
		w.tag(TAG_MARKER, MARK_COEFF_START);
		w.raw16(c, 5);
		w.tag(TAG_MARKER, MARK_LOWPASS_END);
		w.pop();
		// --- EncodeQuantizedFieldPlusTransform (encoder.c:8078) ---
		int subband = 1;
		for (int k = 5; k >= 4; k--) {                     // the spatial wavelets on the temporal lowpass band
			const GopWavelet &wv = ch.w[k];
			put_wavelet_header(w, wv, k + 1);
			for (int b = 1; b < 4; b++, subband++) {
				put_band_header(w, b, wv, subband, 3);         // BAND_ENCODING_RUNLENGTHS
				w.band(c, k, b);
				w.tag(TAG_BAND_TRAILER, 0);
				w.pop();
			}
			w.tag(TAG_MARKER, MARK_HIGHPASS_END);
			w.pop();
		}
		{                                                  // the spatial wavelet on the temporal highpass band: all four bands, the lowpass one as raw words
			const GopWavelet &wv = ch.w[3];
			put_wavelet_header(w, wv, 4);
			for (int b = 0; b < 4; b++, subband++) {
				put_band_header(w, b, wv, subband, b == 0 ? 4 : 3);      // BAND_ENCODING_16BIT for the lowpass band (encoder.c:8219, precision >= 10)
				if (b == 0) { w.raw16(c, 3); w.bytes(band_end, band_end_bytes); }
				else w.band(c, 3, b);
				w.tag(TAG_BAND_TRAILER, 0);
				w.pop();
			}
			w.tag(TAG_MARKER, MARK_HIGHPASS_END);
			w.pop();
		}
		{                                                  // the temporal wavelet: a header and one empty band (encoder.c:6607 EncodeEmptyQuantBand, subband 255)
			const GopWavelet &wv = ch.w[2];
			put_wavelet_header(w, wv, 3);
			put_band_header(w, 1, wv, 255, 3);
			w.tag(TAG_BAND_TRAILER, 0);
			w.pop();
			w.tag(TAG_MARKER, MARK_HIGHPASS_END);
			w.pop();
		}
		for (int k = 1; k >= 0; k--) {                     // the frame wavelets, the second frame first
			const GopWavelet &wv = ch.w[k];
			put_wavelet_header(w, wv, k + 1);
			for (int b = 1; b < 4; b++, subband++) {
				const bool diff = gop_band_is_difference_coded(plan, k, b);
				w.frame_band_begins();
				put_band_header(w, b, wv, subband, 3, diff);
				w.band(c, k, b);
				w.tag(TAG_BAND_TRAILER, 0);
				w.pop();
				if (diff) w.peak_table(wv.quant[b]);
			}
			w.tag(TAG_MARKER, MARK_HIGHPASS_END);
			w.pop();
		}
		w.channel_end(c);
	}
	// --- PutVideoGroupTrailer (codec.c:1075) ---
	w.tag(TAG_SAMPLE, 6);                                // SAMPLE_TYPE_GROUP_TRAILER
	w.tag(TAG_GROUP_TRAILER, 0);
	w.pop();
}
}

size_t write_group_sample(const GopPlan &plan, const SampleHeaderInfo &hdr, const int16_t *coeffs, uint8_t *out, size_t cap)
{
	GroupHostSink sink(out, cap, plan, coeffs);
	walk_group_sample(sink, plan, hdr);
	return sink.w.overflow() ? 0 : sink.w.bytes();
}

void build_group_template(const GopPlan &plan, const SampleHeaderInfo &hdr, SampleTemplate *t)
{
	t->bytes.clear(); t->holes.clear(); t->patches.clear();
	GroupTemplateSink sink(*t, plan);
	walk_group_sample(sink, plan, hdr);
}

// A finished group sample of `bytes` bytes whose last frame-wavelet band starts at or beyond 80% of the reference's sample buffer may have had bands zeroed by the
// reference (encoder.c:8332).  The GPU stage cannot know while it writes: a sample this large is written again by the host writer (conservative: the test is on the
// whole sample, the reference asks in front of every band of the two frame wavelets).
bool gop_sample_may_zero_bands(const GopPlan &plan, size_t bytes) { return (uint64_t)bytes * 100 > (uint64_t)plan.sample_buffer_bytes * 80; }

size_t write_sequence_header(const GopPlan &plan, int input_format, uint8_t *out, size_t cap)
{
	BitWriter w(out, cap);
	w.put_tag(TAG_SAMPLE, 7);                            // SAMPLE_TYPE_SEQUENCE_HEADER (codec.c:736)
	w.put_tag(5, 0); w.put_tag(6, 1); w.put_tag(7, 0); w.put_tag(8, 0);      // version 0.1.0.0 (encoder.c:2930)
	w.put_tag(9, 0);                                     // sequence flags
	w.put_tag(TAG_FRAME_WIDTH, plan.width);
	w.put_tag(TAG_FRAME_HEIGHT, plan.height);
	w.put_tag(TAG_FRAME_FORMAT, 2);                       // (the reference writes 2 for YUY2 and 2vuy alike: pinned on its samples)
	w.put_tag_opt(TAG_INPUT_FORMAT, input_format);
	return w.overflow() ? 0 : w.bytes();
}

size_t write_pframe_sample(const GopPlan &plan, uint32_t frame_number, uint8_t *out, size_t cap)
{
	BitWriter w(out, cap);
	w.put_tag(TAG_SAMPLE, 1);                            // SAMPLE_TYPE_FRAME (codec.c:1258)
	w.put_tag(TAG_FRAME_TYPE, 2);                        // FRAME_TYPE_PFRAME
	w.put_tag(TAG_FRAME_WIDTH, plan.width);
	w.put_tag(TAG_FRAME_HEIGHT, plan.height);
	w.put_tag_opt(TAG_FRAME_NUMBER, (int)(frame_number & 0xffff));
	w.put_tag(TAG_FRAME_INDEX, 1);
	return w.overflow() ? 0 : w.bytes();
}

// ------------------------------------------------------------------------------------------
// Parser
// ------------------------------------------------------------------------------------------
int parse_group_sample(const uint8_t *d, size_t size, ParsedGroup *pg)
{
	*pg = ParsedGroup();
	memset(pg->lowpass, 0, sizeof(pg->lowpass));
	memset(pg->band, 0, sizeof(pg->band));
	size_t pos = 0;
	int channel = 0, wavelet = -1, band = 0, bw = 0, bh = 0, bq = 1, bflags = 0, bsub = 0, benc = 3, lw = 0, lh = 0;
	size_t peak_base = 0; uint32_t peak_offset = 0; int peak_level = 0;
	uint32_t pending = 0; size_t pending_at = 0;
	bool first = true;
	auto rd = [&](size_t o) { return ((uint32_t)d[o] << 24) | ((uint32_t)d[o + 1] << 16) | ((uint32_t)d[o + 2] << 8) | d[o + 3]; };
	while (pos + 4 <= size) {
		const uint32_t word = rd(pos);
		int tag = (int16_t)(word >> 16);
		const int value = (int)(word & 0xffff);
		if (tag < 0) tag = -tag;
		pos += 4;
		if (first) { if (tag != TAG_SAMPLE) return -1; pg->sample_type = value; first = false; if (value != 2) { /* keep reading the few header tags */ } continue; }
		if (tag & 0x4000) {
			const uint32_t bytes = (tag & 0x2000) ? ((((uint32_t)(tag & 0xff) << 16) | (uint32_t)value) * 4) : (uint32_t)value * 4;
			if (pos + bytes > size) return -2;
			pos += bytes;
			continue;
		}
		if (tag & 0x2000) {
			const uint32_t longs = ((uint32_t)(tag & 0xff) << 16) | (uint32_t)value;
			if ((tag & 0xff00) == 0x2000) { pending = longs * 4; pending_at = pos; }
			continue;
		}
		switch (tag) {
		case TAG_SAMPLE: break;                            // channel headers, the group trailer
		case TAG_INDEX: pos += 4 * (size_t)value; break;
		case TAG_CHANNEL: channel = value; if (channel < 0 || channel >= 3) return -3; break;
		case TAG_NUM_CHANNELS: pg->num_channels = value; break;
		case TAG_INPUT_FORMAT: pg->input_format = value; break;
		case TAG_FRAME_WIDTH: pg->width = value; break;
		case TAG_FRAME_HEIGHT: pg->height = value; break;
		case TAG_FRAME_DISPLAY_HEIGHT: pg->display_height = value; break;
		case TAG_FRAME_NUMBER: pg->frame_number = value; break;
		case TAG_PRECISION: pg->precision = value; break;
		case TAG_SAMPLE_FLAGS: pg->progressive = value & 1; break;
		case TAG_LOWPASS_WIDTH: lw = value; break;
		case TAG_LOWPASS_HEIGHT: lh = value; break;
		case TAG_MARKER:
			if (value == MARK_COEFF_START) {
				ParsedBand &pb = pg->lowpass[channel];
				pb.offset = (uint32_t)pos; pb.width = lw; pb.height = lh; pb.quant = 1; pb.present = true; pb.bytes = (uint32_t)((size_t)lw * lh * 2);
				if (!pending) return -4;
				const size_t end = pending_at + pending;
				if (pos + pb.bytes > end || end > size) return -4;
				pos = end; pending = 0;
			}
			break;
		case TAG_WAVELET_NUMBER: wavelet = value - 1; if (wavelet < 0 || wavelet >= kGopWavelets) return -5; break;
		case TAG_BAND_NUMBER: band = value; if (band < 0 || band > 3) return -6; bflags = 0; benc = 3; break;
		case TAG_BAND_CODING_FLAGS: bflags = value; break;
		// peak table of the band that follows (interlaced groups: subbands 12 and 15): offset in bytes from the word behind the OFFSET_L tuple, level (decoder.c:23978-23993)
		case TAG_PEAK_TABLE_OFFSET_L: peak_offset = (peak_offset & ~0xffffu) | (uint32_t)value; peak_base = pos; peak_level = 0; break;
		case TAG_PEAK_TABLE_OFFSET_H: peak_offset = (peak_offset & 0xffffu) | ((uint32_t)value << 16); peak_level = 0; break;
		case TAG_PEAK_LEVEL: peak_level = value; break;
		case TAG_BAND_WIDTH: bw = value; break;
		case TAG_BAND_HEIGHT: bh = value; break;
		case TAG_BAND_SUBBAND: bsub = value; break;
		case TAG_BAND_ENCODING: benc = value; break;
		case TAG_BAND_QUANTIZATION: bq = value; break;
		case TAG_BAND_HEADER: {
			if (wavelet < 0 || !pending) return -7;
			ParsedBand &pb = pg->band[channel][wavelet][band];
			const size_t end = pending_at + pending;
			if (end < pos + 4 || end > size) return -7;
			pb.offset = (uint32_t)pos; pb.bytes = (uint32_t)(end - 4 - pos);
			pb.width = bw; pb.height = bh; pb.quant = bq; pb.subband = bsub; pb.present = true;
			pb.codebook = benc == 4 ? -1 : (bflags & 0xf);      // -1: raw 16-bit words (BAND_ENCODING_16BIT)
			pb.difference = (bflags & 0x10) != 0; pb.peak_level = peak_level; pb.peak_offset = peak_level ? (uint32_t)(peak_base + peak_offset) : 0u;      // (difference-coded bands: interlaced groups)
			if (pb.peak_level && (size_t)pb.peak_offset + 2 > size) return -8;
			peak_level = 0;
			pos = end; pending = 0;
			break; }
		default: break;
		}
	}
	if (pg->sample_type != 2) return 1;
	if (pg->display_height == 0) pg->display_height = pg->height;
	if (!(pg->width > 0 && pg->height > 0 && pg->num_channels == 3)) return -1;
	return 0;
}

} // namespace cfhd
