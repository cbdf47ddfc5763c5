// cfhd_bitstream.cpp -- sample writer / parser and the host-side run-length + VLC stage.
//
// Behaviour restated from the reference's syntax layer (host C, stays on the host per the design):
//   Codec/encoder.c:7461 EncodeQuantizedGroup, :7889 EncodeQuantizedFrameTransform, :6318 EncodeQuantizedBand,
//   :4251 EncodeLowPassBand, :5386 EncodeQuantLongRuns; Codec/codec.c:1364 PutVideoIntraFrameHeader,
//   :1547 PutVideoLowPassHeader, :1609 PutVideoHighPassHeader, :1778 PutVideoBandHeader, :1178 PutVideoGroupExtension;
//   Codec/bitstream.c:819 PutBits, :1389 PadBitsTag, :2206/:2220 SizeTagPush/Pop.
//   Decoder side: Codec/decoder.c:23334 UpdateCodecState (tag state machine), :19534 DecodeBandFSM16sNoGap.
#include "cfhd_bitstream.h"
#include <string.h>

namespace cfhd {

int slow_decode_symbol(int codebook, uint32_t window, int *size, int *run, int *mag, bool *band_end);

// ------------------------------------------------------------------------------------------
// BitWriter
// ------------------------------------------------------------------------------------------
void BitWriter::emit(uint32_t w)
{
	if (n_ + 4 <= cap_) { p_[n_] = (uint8_t)(w >> 24); p_[n_ + 1] = (uint8_t)(w >> 16); p_[n_ + 2] = (uint8_t)(w >> 8); p_[n_ + 3] = (uint8_t)w; }
	else overflow_ = true;
	n_ += 4;
}

void BitWriter::put_bits(uint32_t bits, int nbits)
{
	if (nbits <= 0) return;
	if (nbits < 32) bits &= (1u << nbits) - 1;
	if (nbits <= free_) {
		acc_ = (nbits == 32) ? bits : ((acc_ << nbits) | bits);
		free_ -= nbits;
	} else {
		int rest = nbits - free_;
		if (free_ > 0) acc_ = (acc_ << free_) | (bits >> rest);
		emit(acc_);
		acc_ = bits & ((1u << rest) - 1);
		free_ = 32 - rest;
	}
	if (free_ == 0) { emit(acc_); acc_ = 0; free_ = 32; }
}

void BitWriter::pad32() { if (free_ < 32) put_bits(0, free_); }
void BitWriter::put_long(uint32_t w) { pad32(); emit(w); }

void BitWriter::put_bytes(const void *src, size_t n)
{
	pad32();
	if (n_ + n <= cap_) memcpy(p_ + n_, src, n); else overflow_ = true;
	n_ += n;
}

void BitWriter::patch32(size_t offset, uint32_t v)
{
	if (offset + 4 <= cap_) { p_[offset] = (uint8_t)(v >> 24); p_[offset + 1] = (uint8_t)(v >> 16); p_[offset + 2] = (uint8_t)(v >> 8); p_[offset + 3] = (uint8_t)v; }
}

void BitWriter::size_push(int tag)
{
	pad32();
	if (depth_ < 8) stack_[depth_++] = n_;
	put_tag(tag, 0);
}

void BitWriter::size_pop()
{
	pad32();
	if (depth_ <= 0) return;
	size_t at = stack_[--depth_];
	if (at + 4 > cap_) return;
	int tag = (int16_t)((p_[at] << 8) | p_[at + 1]);
	uint32_t size = (uint32_t)((n_ - at) >> 2);
	size = size >= 1 ? size - 1 : 0;                 // longwords that follow the tag/value pair
	if (tag & 0x2000) { tag |= (int)(size >> 16) & 0xff; size &= 0xffff; }
	else size &= 0xffff;
	tag = -tag;                                       // size chunks are optional tags
	patch32(at, ((uint32_t)(uint16_t)tag << 16) | size);
}

// ------------------------------------------------------------------------------------------
// Host VLC (encoder.c:5386)
// ------------------------------------------------------------------------------------------
// peaks != nullptr: EncodeQuantLongRunsPlusPeaks (encoder.c:4802): a coefficient beyond +-PEAK_THRESHOLD (250, codec.h:155) is coded as
// +-251 and its value * quant goes to the peak table that follows the band.
void vlc_encode_band(BitWriter &w, const int16_t *band, int width, int height, int pitch, int codebook, int quant, std::vector<int16_t> *peaks)
{
	const EntropyTables *t = entropy_tables(codebook);
	const int gap = pitch - width;
	int count = 0;
	auto put_run = [&](int c) {
		while (c > 0) {
			int idx = c < 3072 ? c : 3071;
			w.put_bits(t->run_bits[idx], t->run_size[idx]);
			c -= t->run_count[idx];
		}
	};
	for (int row = 0; row < height; row++) {
		const int16_t *p = band + (size_t)row * pitch;
		for (int i = 0; i < width; i++) {
			int v = p[i];
			if (v == 0) { count++; continue; }
			if (count) { put_run(count); count = 0; }
			if (peaks && (v > kPeakThreshold || v < -kPeakThreshold)) { peaks->push_back((int16_t)(v * quant)); v = v > 0 ? kPeakThreshold + 1 : -kPeakThreshold - 1; }
			if (v < 0) { if (v <= -1024) v = -1023; v += 2048; } else if (v >= 1024) v = 1023;
			uint32_t e = t->value_code[v];
			w.put_bits(e & 0x7FFFFFFu, (int)(e >> 27));
		}
		count += gap;
	}
	if (count) put_run(count);
	w.put_bits(t->band_end_bits, t->band_end_size);
	w.pad32();
}

// ------------------------------------------------------------------------------------------
// Sample writer
// ------------------------------------------------------------------------------------------
// The syntax walk is written once against a "sink": BitWriter-backed (host entropy: the payloads are produced inline) or
// the template recorder (GPU entropy: payloads become holes that the device fills, size fields become patches).
namespace {

struct HostSink {
	BitWriter w; const FramePlan &plan; const BandSource &src;
	size_t index_at = 0, channel_start = 0;
	HostSink(uint8_t *out, size_t cap, const FramePlan &p, const BandSource &s) : w(out, cap), plan(p), src(s) {}
	void tag(int t, int v) { w.put_tag(t, v); }
	void tag_opt(int t, int v) { w.put_tag_opt(t, v); }
	void bytes(const void *p, size_t n) { w.put_bytes(p, n); }
	void push(int t) { w.size_push(t); }
	void pop() { w.size_pop(); }
	void index_entries(int n) { index_at = w.bytes(); for (int i = 0; i < n; i++) w.put_tag(TAG_ENTRY, i); }
	void channel_begin(int) { channel_start = w.bytes(); }
	void channel_end(int c) { w.patch32(index_at + 4 * (size_t)c, (uint32_t)(w.bytes() - channel_start)); }
	void lowpass(int c)
	{
		const BandDesc &ll = plan.ch[c].band[2][0];
		const int16_t *base = src.coeffs + ll.offset;
		for (int r = 0; r < ll.height; r++) {
			const int16_t *row = base + (size_t)r * ll.pitch;
			if (ll.width & 1) { for (int x = 0; x < ll.width; x++) w.put_bits((uint16_t)row[x], 16); }
			else for (int x = 0; x < ll.width; x += 2) w.put_long(((uint32_t)(uint16_t)row[x] << 16) | (uint16_t)row[x + 1]);
		}
		w.pad32();
	}
	std::vector<int16_t> peaks; size_t peak_tags_at = 0; bool collect_peaks = false;
	void band(int c, int lv, int b, int k, int codebook)
	{
		const BandDesc &bd = plan.ch[c].band[lv][b];
		if (src.packed) w.put_bytes(src.packed[c * 9 + k], src.packed_bytes[c * 9 + k]);
		else vlc_encode_band(w, src.coeffs + bd.offset, bd.width, bd.height, bd.pitch, codebook, bd.quant, collect_peaks ? &peaks : nullptr);
	}
	// the three optional tags in front of a peak-coded band's size chunk (codec.c:1804-1809), zero until a table follows
	void peak_tags() { peak_tags_at = w.bytes(); w.put_tag_opt(TAG_PEAK_TABLE_OFFSET_L, 0); w.put_tag_opt(TAG_PEAK_TABLE_OFFSET_H, 0); w.put_tag_opt(TAG_PEAK_LEVEL, 0); peaks.clear(); collect_peaks = true; }
	// behind the band trailer: offset and level patched into the tags, then the table chunk -- 16-bit values in host order, padded
	// to a whole longword (encoder.c:6543-6585)
	void peak_table(int quant)
	{
		collect_peaks = false;
		if (peaks.empty()) return;
		if ((peaks.size() + 1) / 2 > 0xffff) return;        // more longwords than the chunk header can count (MAX_CHUNK_SIZE, codec.h:195): the reference writes no table and leaves the three tags zero (encoder.c:6557)
		const uint32_t offset = (uint32_t)(w.bytes() - peak_tags_at);
		auto tagword = [](int t, uint32_t v) { return ((uint32_t)(uint16_t)(-t) << 16) | (v & 0xffffu); };
		w.patch32(peak_tags_at, tagword(TAG_PEAK_TABLE_OFFSET_L, offset & 0xffffu));
		w.patch32(peak_tags_at + 4, tagword(TAG_PEAK_TABLE_OFFSET_H, offset >> 16));
		w.patch32(peak_tags_at + 8, tagword(TAG_PEAK_LEVEL, (uint32_t)(kPeakThreshold * quant)));
		if (peaks.size() & 1) peaks.push_back(0);
		w.put_tag_opt(TAG_PEAK_TABLE, (int)(peaks.size() / 2));
		w.put_bytes(peaks.data(), peaks.size() * 2);
	}
};

struct TemplateSink : TemplateRecorder {
	explicit TemplateSink(SampleTemplate &tt) : TemplateRecorder(tt) {}
	void lowpass(int c) { const BandDesc &ll = t.plan.ch[c].band[2][0]; hole(0, c, 2, 0, (((ll.width * ll.height * 2) + 3) / 4) * 4); }
	void band(int c, int lv, int bnd, int, int) { hole(1, c, lv, bnd, 0); }
};

template <typename Sink>
void walk_sample(Sink &w, const FramePlan &plan, const SampleHeaderInfo &hdr)
{
	const int nch = plan.num_channels;

	// --- PutVideoIntraFrameHeader (codec.c:1364) ---
	w.tag(TAG_SAMPLE, SAMPLE_TYPE_IFRAME);
	w.tag(TAG_INDEX, nch);
	w.index_entries(nch);
	w.tag(TAG_TRANSFORM_TYPE, 0);
	w.tag(TAG_NUM_FRAMES, 1);
	w.tag(TAG_NUM_CHANNELS, nch);
	if (hdr.input_format >= 100) w.tag(TAG_INPUT_FORMAT, hdr.input_format);
	else w.tag_opt(TAG_INPUT_FORMAT, hdr.input_format);
	w.tag(TAG_ENCODED_FORMAT, plan.encoded_format);
	{
		int cs = hdr.color_space;
		if (plan.encoded_format == ENC_YUV422) cs &= ~4;
		else if (plan.encoded_format == ENC_BAYER) cs = 0;
		else cs &= ~3;
		if (cs) w.tag_opt(TAG_ENCODED_COLORSPACE, cs);
	}
	w.tag(TAG_NUM_WAVELETS, kNumLevels);
	w.tag(TAG_NUM_SUBBANDS, 10);
	w.tag(TAG_NUM_SPATIAL, 2);
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
		for (int i = 0; i < kNumLevels; i++) table += (unsigned)plan.prescale[i] << (14 - i * 2);
		// optional only while the table equals the decoder's built-in 10-bit spatial default {0,2,0} (codec.c:1515-1524,
		// TestTransformPrescaleMatch wavelet.c:1784): the 12-bit table {0,2,2} must be transmitted
		const bool is_default = plan.prescale[0] == 0 && plan.prescale[1] == 2 && plan.prescale[2] == 0;
		if (is_default) w.tag_opt(TAG_PRESCALE_TABLE, (int)table); else w.tag(TAG_PRESCALE_TABLE, (int)table);
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

	for (int c = 0; c < nch; c++) {
		const ChannelPlan &cp = plan.ch[c];
		if (c > 0) { w.tag(TAG_SAMPLE, SAMPLE_TYPE_CHANNEL); w.tag(TAG_CHANNEL, c); }
		w.channel_begin(c);

		// --- EncodeLowPassBand (encoder.c:4251): raw 16-bit big-endian coefficients ---
		const BandDesc &ll = cp.band[2][0];
		w.tag(TAG_MARKER, MARK_LOWPASS_START);
		w.tag(TAG_LOWPASS_SUBBAND, 0);
		w.tag(TAG_NUM_LEVELS, 3);
		w.tag(TAG_LOWPASS_WIDTH, ll.width);
		w.tag(TAG_LOWPASS_HEIGHT, ll.height);
		w.tag(TAG_MARGIN_LEFT, 0); w.tag(TAG_MARGIN_TOP, 0); w.tag(TAG_MARGIN_RIGHT, 0); w.tag(TAG_MARGIN_BOTTOM, 0);
		w.tag(TAG_PIXEL_OFFSET, 0);
		w.tag(TAG_QUANTIZATION, 1);
		w.tag(TAG_PIXEL_DEPTH, 16);
		w.push(TAG_SUBBAND_SIZE);
		w.tag(TAG_MARKER, MARK_COEFF_START);
		w.lowpass(c);
		w.tag(TAG_MARKER, MARK_LOWPASS_END);
		w.pop();

		// --- EncodeQuantizedFrameTransform (encoder.c:7889) ---
		int subband = 1, k = 0;
		for (int lv = kNumLevels - 1; lv >= 0; lv--) {
			const BandDesc &b1 = cp.band[lv][1];
			w.tag(TAG_MARKER, MARK_HIGHPASS_START);
			w.tag(TAG_WAVELET_TYPE, lv == 0 ? 5 : 3);       // level 1 is the "frame" wavelet (horizontal-temporal type)
			w.tag(TAG_WAVELET_NUMBER, lv + 1);
			w.tag(TAG_WAVELET_LEVEL, lv + 1);
			w.tag(TAG_NUM_BANDS, 4);
			w.tag(TAG_HIGHPASS_WIDTH, b1.width);
			w.tag(TAG_HIGHPASS_HEIGHT, b1.height);
			w.tag(TAG_LOWPASS_BORDER, 0);
			w.tag(TAG_HIGHPASS_BORDER, 0);
			w.tag(TAG_LOWPASS_SCALE, cp.band[lv][0].scale);
			w.tag(TAG_LOWPASS_DIVISOR, 0);
			w.push(TAG_LEVEL_SIZE);
			for (int b = 1; b < 4; b++, subband++, k++) {
				const BandDesc &bd = cp.band[lv][b];
				// SetCodingFlags (encoder.c:6120): code set 17 for every band of a progressive intra frame; an interlaced intra frame
				// codes subband 8 (the temporal-highpass, horizontal-lowpass band of the frame wavelet) with code set 18, difference
				// coding (flag 16) and a peak table
				const bool diffband = !hdr.progressive && subband == 8;
				const int codebook = diffband ? 2 : 1;
				w.tag(TAG_MARKER, MARK_BAND_START);
				w.tag(TAG_BAND_NUMBER, b);
				w.tag(TAG_BAND_CODING_FLAGS, codebook + (diffband ? 16 : 0));
				w.tag(TAG_BAND_WIDTH, bd.width);
				w.tag(TAG_BAND_HEIGHT, bd.height);
				w.tag(TAG_BAND_SUBBAND, subband);
				w.tag(TAG_BAND_ENCODING, 3);                 // BAND_ENCODING_RUNLENGTHS
				w.tag(TAG_BAND_QUANTIZATION, bd.quant);
				w.tag(TAG_BAND_SCALE, bd.scale);
				if (diffband) w.peak_tags();
				w.push(TAG_SUBBAND_SIZE);
				w.tag(TAG_BAND_HEADER, 0);
				w.band(c, lv, b, k, codebook);
				w.tag(TAG_BAND_TRAILER, 0);
				w.pop();
				if (diffband) w.peak_table(bd.quant);
			}
			w.tag(TAG_MARKER, MARK_HIGHPASS_END);
			w.pop();
		}
		w.channel_end(c);
	}
	w.tag(TAG_FRAME_TRAILER, 0);
	w.pop();
}

} // namespace

size_t write_sample(const FramePlan &plan, const SampleHeaderInfo &hdr, const BandSource &src, uint8_t *out, size_t cap)
{
	HostSink sink(out, cap, plan, src);
	walk_sample(sink, plan, hdr);
	return sink.w.overflow() ? 0 : sink.w.bytes();
}

void build_sample_template(const FramePlan &plan, const SampleHeaderInfo &hdr, SampleTemplate *t)
{
	t->plan = plan;
	t->bytes.clear(); t->holes.clear(); t->patches.clear();
	TemplateSink sink(*t);
	walk_sample(sink, plan, hdr);
}

// ------------------------------------------------------------------------------------------
// Parser
// ------------------------------------------------------------------------------------------
int parse_sample(const uint8_t *d, size_t size, ParsedSample *ps)
{
	*ps = ParsedSample();
	ps->size = size;
	memset(ps->lowpass, 0, sizeof(ps->lowpass));
	memset(ps->high, 0, sizeof(ps->high));
	size_t pos = 0;
	int channel = 0, lv = -1, band = 0, bw = 0, bh = 0, bq = 1, bflags = 0, bsub = 0;
	int lw = 0, lh = 0;
	size_t peak_base = 0; uint32_t peak_offset = 0; int peak_level = 0;
	uint32_t pending_chunk = 0;   // bytes of the SUBBAND_SIZE chunk that was just opened
	bool truncated = false;       // ran off the end of the supplied bytes (callers that only need the header pass 512 bytes)
	size_t pending_at = 0;
	auto rd = [&](size_t o) { return ((uint32_t)d[o] << 24) | ((uint32_t)d[o + 1] << 16) | ((uint32_t)d[o + 2] << 8) | d[o + 3]; };
	while (pos + 4 <= size && !truncated) {
		uint32_t word = rd(pos);
		int tag = (int16_t)(word >> 16);
		int value = (int)(word & 0xffff);
		bool optional = tag < 0;
		if (optional) tag = -tag;
		pos += 4;
		if (tag & 0x4000) {                       // chunks with payload (metadata, peak tables): skip the payload
			uint32_t bytes = (tag & 0x2000) ? ((((uint32_t)(tag & 0xff) << 16) | (uint32_t)value) * 4) : (uint32_t)value * 4;
			if ((tag == TAG_METADATA || (tag & 0xff00) == 0x6000) && ps->metadata_bytes == 0) { ps->metadata_offset = (uint32_t)pos; ps->metadata_bytes = bytes; }
			if (pos + bytes > size) { truncated = true; break; }
			pos += bytes;
			continue;
		}
		if (tag & 0x2000) {                       // 24-bit size chunks: SUBBAND/LEVEL/SAMPLE size (contents are parsed)
			uint32_t longs = ((uint32_t)(tag & 0xff) << 16) | (uint32_t)value;
			if ((tag & 0xff00) == 0x2000) { pending_chunk = longs * 4; pending_at = pos; }
			continue;
		}
		switch (tag) {
		case TAG_SAMPLE: if (value == SAMPLE_TYPE_CHANNEL) { /* channel header follows */ } break;
		case TAG_INDEX: pos += 4 * (size_t)value; break;
		case TAG_CHANNEL: channel = value; if (channel < 0 || channel >= kMaxChannels) return -3; break;
		case TAG_TRANSFORM_TYPE: ps->transform_type = value; break;
		case TAG_NUM_CHANNELS: ps->num_channels = value; break;
		case TAG_NUM_WAVELETS: ps->num_wavelets = value; break;
		case TAG_NUM_SPATIAL: ps->num_spatial = value; break;
		case TAG_INPUT_FORMAT: ps->input_format = value; break;
		case TAG_ENCODED_FORMAT: ps->encoded_format = value; break;
		case TAG_ENCODED_COLORSPACE: ps->color_space = value; break;
		case TAG_FRAME_WIDTH: ps->width = value; break;
		case TAG_FRAME_HEIGHT: ps->height = value; break;
		case TAG_FRAME_DISPLAY_HEIGHT: ps->display_height = value; break;
		case TAG_FRAME_NUMBER: ps->frame_number = value; break;
		case TAG_PRECISION: ps->precision = value; break;
		case TAG_VERSION: ps->version = value; break;
		case TAG_QUALITY_L: ps->quality = (ps->quality & ~0xffff) | value; break;
		case TAG_QUALITY_H: ps->quality = (ps->quality & 0xffff) | (value << 16); break;
		case TAG_PRESCALE_TABLE: ps->prescale_table = value; break;
		case TAG_SAMPLE_FLAGS: ps->progressive = value & 1; break;
		case TAG_INTERLACED_FLAGS: ps->interlaced_flags = value; break;
		case TAG_LOWPASS_WIDTH: lw = value; break;
		case TAG_LOWPASS_HEIGHT: lh = value; break;
		case TAG_MARKER:
			if (value == MARK_COEFF_START) {       // raw lowpass coefficients follow, inside the pending SUBBAND_SIZE chunk
				ParsedBand &pb = ps->lowpass[channel];
				pb.offset = (uint32_t)pos; pb.width = lw; pb.height = lh; pb.quant = 1; pb.present = true;
				pb.bytes = (uint32_t)((size_t)lw * lh * 2);
				if (pending_chunk == 0) return -4;
				size_t end = pending_at + pending_chunk;
				if (pos + pb.bytes > end) return -4;
				if (end > size) { pb.present = false; truncated = true; break; }
				pos = end; pending_chunk = 0;
			}
			break;
		case TAG_WAVELET_NUMBER: lv = value - 1; if (lv < 0 || lv >= kNumLevels) return -5; break;
		case TAG_BAND_NUMBER: band = value; if (band < 1 || band > 3) return -6; bflags = 0; break;
		case TAG_BAND_CODING_FLAGS: bflags = value; break;
		// peak table of the band that follows: offset (bytes) from the word behind the OFFSET_L tuple, level (decoder.c:23978-23993)
		case TAG_PEAK_TABLE_OFFSET_L: peak_offset = (peak_offset & ~0xffffu) | (uint32_t)value; peak_base = pos; peak_level = 0; break;
		case TAG_PEAK_TABLE_OFFSET_H: peak_offset = (peak_offset & 0xffffu) | ((uint32_t)value << 16); peak_level = 0; break;
		case TAG_PEAK_LEVEL: peak_level = value; break;
		case TAG_BAND_WIDTH: bw = value; break;
		case TAG_BAND_HEIGHT: bh = value; break;
		case TAG_BAND_SUBBAND: bsub = value; break;
		case TAG_BAND_QUANTIZATION: bq = value; break;
		case TAG_BAND_HEADER: {
			if (lv < 0 || pending_chunk == 0) return -7;
			ParsedBand &pb = ps->high[channel][lv][band];
			size_t end = pending_at + pending_chunk;     // chunk covers BAND_HEADER .. BAND_TRAILER
			if (end < pos + 4) return -7;
			if (end > size) { truncated = true; break; }
			pb.offset = (uint32_t)pos; pb.bytes = (uint32_t)(end - 4 - pos);
			pb.width = bw; pb.height = bh; pb.quant = bq; pb.codebook = bflags & 0xf; pb.subband = bsub; pb.present = true;
			pb.difference = (bflags >> 4) & 1; pb.peak_level = peak_level; pb.peak_offset = peak_level ? (uint32_t)(peak_base + peak_offset) : 0u;
			if (pb.peak_level && (size_t)pb.peak_offset + 2 > size) return -8;
			peak_level = 0;
			pos = end; pending_chunk = 0;
			break; }
		default: break;
		}
	}
	if (ps->display_height == 0) ps->display_height = ps->height;
	if (ps->precision == 0) ps->precision = 8;
	if (!(ps->width > 0 && ps->height > 0 && ps->num_channels > 0)) return -1;
	return truncated ? 1 : 0;
}

int lowpass_bias(int precision, int lowpass_width, int out_pixel_kind, int channel)
{
	const bool even = (lowpass_width & 1) == 0;
	if (precision == 8) return 32;
	if (precision == 10) {
		// decoder.c:12265-12276: the 16-bit and 10-bit 4:2:2 outputs (YU64, YR16, V210) take 4 where the 8-bit ones take 24; odd widths: :12479
		if (out_pixel_kind == PIX_YU64 || out_pixel_kind == PIX_V210) return even ? 4 : 5;
		// RGB24 / RGB32 output (bottom row first: RG24, BGRA -- not the top-down BGRa) of a 10-bit sample, bit-serial path of odd widths only: the reference
		// takes 8 off the luma bias and 4 off the chroma bias ("fixed rounding error introduced by YUV->RGB", decoder.c:12500-12508)
		if (!even && (out_pixel_kind == PIX_RG24 || out_pixel_kind == PIX_BGRA)) return channel == 0 ? 5 - 8 : 5 - 4;
		return even ? 24 : 5;
	}
	if (precision == 12) {
		// decoder.c:12290-12312: the 8-bit RGB outputs take 8, the 10-bit RGB words 6, the 16-bit outputs (RG48, b64a, ...) none; Bayer samples never any (:12316)
		if (out_pixel_kind == PIX_RG24 || out_pixel_kind == PIX_BGRA || out_pixel_kind == PIX_BGRa) return 8;
		if (out_pixel_kind >= PIX_R210 && out_pixel_kind <= PIX_AR10) return 6;
	}
	return 0;
}

// ------------------------------------------------------------------------------------------
// Host VLC decode
// ------------------------------------------------------------------------------------------
int vlc_decode_band(const uint8_t *data, size_t bytes, int width, int height, int pitch, int quant, int codebook, int16_t *band)
{
	(void)width;
	const EntropyTables *t = entropy_tables(codebook ? codebook : 1);
	if (!t) return -1;
	const int K = EntropyTables::kDecBits;
	const size_t total = (size_t)height * pitch;
	size_t idx = 0;
	uint64_t acc = 0; int have = 0; size_t pos = 0;
	auto refill = [&]() { while (have <= 56 && pos < bytes) { acc |= (uint64_t)data[pos++] << (56 - have); have += 8; } };
	for (;;) {
		refill();
		if (have <= 0) return -2;
		uint32_t e = t->dec_lut[(uint32_t)(acc >> (64 - K))];
		int size = (int)(e & 31), run = (int)((e >> 5) & 0x7ff), mag = (int)(e >> 16);
		if (size == 0) {
			bool end;
			if (slow_decode_symbol(codebook ? codebook : 1, (uint32_t)(acc >> 32), &size, &run, &mag, &end) < 0) return -3;
			if (end) return 0;
		}
		acc <<= size; have -= size;
		if (mag) {
			int negative = (int)(acc >> 63);
			acc <<= 1; have -= 1;
			if (idx >= total) return -4;
			int v = mag * quant;
			band[idx++] = (int16_t)(negative ? -v : v);
		} else {
			idx += (size_t)run;
		}
		if (have < 0) return -2;
	}
}

void finish_difference_band(int16_t *band, int width, int height, int pitch, const uint8_t *peaks, size_t peak_bytes, int peak_level)
{
	size_t next = 0;
	for (int y = 0; y < height; y++) {
		int16_t *line = band + (size_t)y * pitch;
		if (peak_level && peaks)
			for (int x = 0; x < width; x++) {
				const int v = line[x];
				if ((v < 0 ? -v : v) > peak_level && next + 2 <= peak_bytes) { line[x] = (int16_t)(peaks[next] | (peaks[next + 1] << 8)); next += 2; }
			}
		for (int x = 1; x < width; x++) line[x] = (int16_t)(line[x] + line[x - 1]);
	}
}

} // namespace cfhd
