// cfhd_bitstream.h -- CineForm sample syntax: writer (encoder side) and parser (decoder side).
#pragma once
#include "cfhd_core.h"
#include <vector>

namespace cfhd {

// Tags of the tag/value syntax (16-bit tag, 16-bit value, big endian; a negated tag is optional).
// Numbering follows Codec/codec.h:201-416.
enum Tag : int {
	TAG_SAMPLE = 1, TAG_INDEX = 2, TAG_ENTRY = 3, TAG_MARKER = 4,
	TAG_TRANSFORM_TYPE = 10, TAG_NUM_FRAMES = 11, TAG_NUM_CHANNELS = 12, TAG_NUM_WAVELETS = 13, TAG_NUM_SUBBANDS = 14,
	TAG_NUM_SPATIAL = 15, TAG_FIRST_WAVELET = 16, TAG_GROUP_TRAILER = 18,
	TAG_FRAME_TYPE = 19, TAG_FRAME_WIDTH = 20, TAG_FRAME_HEIGHT = 21, TAG_FRAME_FORMAT = 22, TAG_FRAME_INDEX = 23, TAG_FRAME_TRAILER = 24,
	TAG_LOWPASS_SUBBAND = 25, TAG_NUM_LEVELS = 26, TAG_LOWPASS_WIDTH = 27, TAG_LOWPASS_HEIGHT = 28,
	TAG_MARGIN_TOP = 29, TAG_MARGIN_BOTTOM = 30, TAG_MARGIN_LEFT = 31, TAG_MARGIN_RIGHT = 32,
	TAG_PIXEL_OFFSET = 33, TAG_QUANTIZATION = 34, TAG_PIXEL_DEPTH = 35,
	TAG_WAVELET_TYPE = 37, TAG_WAVELET_NUMBER = 38, TAG_WAVELET_LEVEL = 39, TAG_NUM_BANDS = 40,
	TAG_HIGHPASS_WIDTH = 41, TAG_HIGHPASS_HEIGHT = 42, TAG_LOWPASS_BORDER = 43, TAG_HIGHPASS_BORDER = 44,
	TAG_LOWPASS_SCALE = 45, TAG_LOWPASS_DIVISOR = 46,
	TAG_BAND_NUMBER = 48, TAG_BAND_WIDTH = 49, TAG_BAND_HEIGHT = 50, TAG_BAND_SUBBAND = 51, TAG_BAND_ENCODING = 52,
	TAG_BAND_QUANTIZATION = 53, TAG_BAND_SCALE = 54, TAG_BAND_HEADER = 55, TAG_BAND_TRAILER = 56,
	TAG_CHANNEL = 62, TAG_INTERLACED_FLAGS = 63, TAG_PROTECTION_FLAGS = 64, TAG_PICTURE_ASPECT_X = 65, TAG_PICTURE_ASPECT_Y = 66,
	TAG_SAMPLE_FLAGS = 68, TAG_FRAME_NUMBER = 69, TAG_PRECISION = 70, TAG_INPUT_FORMAT = 71, TAG_BAND_CODING_FLAGS = 72,
	TAG_VERSION = 79, TAG_QUALITY_L = 80, TAG_QUALITY_H = 81, TAG_BAND_SECONDPASS = 82, TAG_PRESCALE_TABLE = 83,
	TAG_ENCODED_FORMAT = 84, TAG_FRAME_DISPLAY_HEIGHT = 85, TAG_ENCODED_COLORSPACE = 91, TAG_ENCODED_CHANNEL_NUMBER = 93,
	TAG_PEAK_LEVEL = 74, TAG_PEAK_TABLE_OFFSET_L = 75, TAG_PEAK_TABLE_OFFSET_H = 76, TAG_PEAK_TABLE = 0x4001,
	TAG_SUBBAND_SIZE = 0x2000, TAG_LEVEL_SIZE = 0x2100, TAG_SAMPLE_SIZE = 0x2200,
	TAG_METADATA = 0x4002,
};
enum { SAMPLE_TYPE_CHANNEL = 3, SAMPLE_TYPE_IFRAME = 9 };
enum { MARK_LOWPASS_START = 0x1A4A, MARK_LOWPASS_END = 0x1B4B, MARK_COEFF_START = 0x0F0F,
       MARK_HIGHPASS_START = 0x0D0D, MARK_HIGHPASS_END = 0x0C0C, MARK_BAND_START = 0x0E0E };

// MSB-first bit writer into 32-bit big-endian words (Codec/bitstream.c:819 PutBits semantics).
class BitWriter {
public:
	BitWriter(uint8_t *buf, size_t cap) : p_(buf), cap_(cap) {}
	size_t bytes() const { return n_; }                // whole words emitted so far
	bool overflow() const { return overflow_; }
	void put_bits(uint32_t bits, int nbits);
	void pad32();                                       // PadBits32 + flush
	void put_long(uint32_t w);                          // requires 32-bit alignment
	void put_tag(int tag, int value) { put_long(((uint32_t)(uint16_t)tag << 16) | (uint32_t)(value & 0xffff)); }
	void put_tag_opt(int tag, int value) { put_tag(-tag, value); }
	void put_bytes(const void *src, size_t n);          // requires alignment; n multiple of 4
	void size_push(int tag);                            // SizeTagPush (bitstream.c:2206)
	void size_pop();                                    // SizeTagPop  (bitstream.c:2220)
	uint8_t *cursor() { return p_ + n_; }
	void patch32(size_t offset, uint32_t value_be);
private:
	void emit(uint32_t w);
	uint8_t *p_; size_t cap_; size_t n_ = 0;
	uint32_t acc_ = 0; int free_ = 32; bool overflow_ = false;
	size_t stack_[8]; int depth_ = 0;
};

struct SampleHeaderInfo {
	uint32_t frame_number;
	int input_format;        // reference COLOR_FORMAT_* of the submitted frame
	int color_space;         // reference COLOR_SPACE_* flags
	int encoder_quality;     // 32-bit quality word as passed to the codec
	bool progressive;
	const uint8_t *meta_global; size_t meta_global_size;
	const uint8_t *meta_local; size_t meta_local_size;
	bool channel_number_tag = false;   // metadata carried TAG_VIDEO_CHANNELS: the reference then also writes -ENCODED_CHANNEL_NUMBER=0 (encoder.c:7553, :9043-9046)
};

// Payload provider for one highpass band: either host-side VLC from coefficients, or bytes already
// packed on the GPU.  Must append whole 32-bit words (band end code included, zero padded).
struct BandSource {
	const int16_t *coeffs = nullptr;     // frame coefficient buffer (FramePlan layout) for host VLC
	const uint8_t *const *packed = nullptr;  // [channel*9 + k] -> GPU packed payload (k = coding order index 0..8)
	const uint32_t *packed_bytes = nullptr;
};

// Sample template for the GPU entropy stage: every fixed byte of the sample in order, with the positions where the device
// inserts payloads (holes: raw lowpass words, entropy coded bands) and the size fields that depend on the payload sizes.
// A template byte at offset x preceded by h holes ends up at x + (sum of the sizes of the first h holes) in the sample.
struct SampleTemplate {
	struct Hole { int tmpl_offset; int kind /*0 = lowpass raw, 1 = coded band, 2 = peak table of the coded band in front of it (empty when the band has no peaks; fixed_bytes: the band's divisor)*/; int channel, level, band; int fixed_bytes; };
	struct Patch {          // kind 0: 24/16-bit chunk size at (at_tmpl, at_holes), chunk ends at (end_tmpl, end_holes)
		int kind;           // kind 1: 32-bit big-endian byte count end - start written at at_tmpl (channel index entry)
		                    // kind 2: the three peak tags at (at_tmpl, at_holes) of the table in hole start_holes, which begins at (end_tmpl, end_holes); tag: the band's divisor.  Left zero when the hole is empty.
		int at_tmpl, at_holes, start_tmpl, start_holes, end_tmpl, end_holes, tag;
	};
	FramePlan plan;
	std::vector<uint8_t> bytes;
	std::vector<Hole> holes;
	std::vector<Patch> patches;
};
void build_sample_template(const FramePlan &plan, const SampleHeaderInfo &hdr, SampleTemplate *t);

// The recorder behind a template: the tag / chunk interface of the syntax walks (cfhd_bitstream.cpp walk_sample, cfhd_gop.cpp walk_group_sample), writing fixed
// bytes, holes and patches instead of a bit stream.
struct TemplateRecorder {
	SampleTemplate &t;
	std::vector<uint8_t> &b;
	int holes = 0; int stack[8]; int depth = 0; int index_at = 0; int ch_start_tmpl = 0, ch_start_holes = 0;
	explicit TemplateRecorder(SampleTemplate &tt) : t(tt), b(tt.bytes) {}
	void word(uint32_t w) { b.push_back((uint8_t)(w >> 24)); b.push_back((uint8_t)(w >> 16)); b.push_back((uint8_t)(w >> 8)); b.push_back((uint8_t)w); }
	void tag(int tg, int v) { word(((uint32_t)(uint16_t)tg << 16) | (uint32_t)(v & 0xffff)); }
	void tag_opt(int tg, int v) { tag(-tg, v); }
	void bytes(const void *p, size_t n) { const uint8_t *q = (const uint8_t *)p; b.insert(b.end(), q, q + n); }
	void push(int tg)
	{
		SampleTemplate::Patch p; p.kind = 0; p.at_tmpl = (int)b.size(); p.at_holes = holes; p.tag = tg; p.start_tmpl = p.start_holes = p.end_tmpl = p.end_holes = 0;
		stack[depth++] = (int)t.patches.size(); t.patches.push_back(p);
		tag(tg, 0);
	}
	void pop() { SampleTemplate::Patch &p = t.patches[stack[--depth]]; p.end_tmpl = (int)b.size(); p.end_holes = holes; }
	void index_entries(int n) { index_at = (int)b.size(); for (int i = 0; i < n; i++) tag(TAG_ENTRY, i); }
	void channel_begin(int) { ch_start_tmpl = (int)b.size(); ch_start_holes = holes; }
	void channel_end(int c)
	{
		SampleTemplate::Patch p; p.kind = 1; p.at_tmpl = index_at + 4 * c; p.at_holes = 0; p.tag = 0;
		p.start_tmpl = ch_start_tmpl; p.start_holes = ch_start_holes; p.end_tmpl = (int)b.size(); p.end_holes = holes;
		t.patches.push_back(p);
	}
	void hole(int kind, int c, int lv, int bnd, int fixed) { SampleTemplate::Hole h = { (int)b.size(), kind, c, lv, bnd, fixed }; t.holes.push_back(h); holes++; }
	// The peak table of a difference-coded band (codec.c:1804-1809, encoder.c:6543-6585): three optional tags in front of the band's size chunk, zero while the band has
	// no values beyond the threshold, and behind the band trailer the table chunk -- a hole that stays empty for such a band.  The device fills both (k_ent_layout, k_ent_peaks).
	int peak_at_tmpl = 0, peak_at_holes = 0;
	void peak_tags() { peak_at_tmpl = (int)b.size(); peak_at_holes = holes; tag_opt(TAG_PEAK_TABLE_OFFSET_L, 0); tag_opt(TAG_PEAK_TABLE_OFFSET_H, 0); tag_opt(TAG_PEAK_LEVEL, 0); }
	void peak_table(int quant)
	{
		const SampleTemplate::Hole band = t.holes.back();       // (the coded band the table belongs to)
		SampleTemplate::Patch p; p.kind = 2; p.at_tmpl = peak_at_tmpl; p.at_holes = peak_at_holes; p.start_tmpl = 0; p.start_holes = holes; p.end_tmpl = (int)b.size(); p.end_holes = holes; p.tag = quant;
		t.patches.push_back(p);
		hole(2, band.channel, band.level, band.band, quant);
	}
};

// Writes a complete intra-frame sample.  Returns the sample size in bytes, or 0 on overflow.
size_t write_sample(const FramePlan &plan, const SampleHeaderInfo &hdr, const BandSource &src, uint8_t *out, size_t cap);

// Host VLC of one band (the reference's EncodeQuantLongRuns + band end code + pad), appended to w.
enum { kPeakThreshold = 250 };   // PEAK_THRESHOLD, Codec/codec.h:155
void vlc_encode_band(BitWriter &w, const int16_t *band, int width, int height, int pitch, int codebook, int quant = 1, std::vector<int16_t> *peaks = nullptr);

// ---- parser ----
struct ParsedBand {
	uint32_t offset, bytes; int width, height, quant, codebook, subband; bool present;
	// the difference-coded band of an interlaced frame (subband 8 of every channel): coefficients are coded as the difference to their left
	// neighbour (BAND_CODING_FLAGS bit 4, decoder.c:23974, :20822); peak_level != 0: values beyond it were coded as +-(level / quant + 1)
	// and the real (dequantized) values follow in raster order as 16-bit little-endian words at peak_offset (decoder.c:23978-23993, :19809)
	bool difference; int peak_level; uint32_t peak_offset;
};
struct ParsedSample {
	int width = 0, height = 0, display_height = 0, num_channels = 0, precision = 0, encoded_format = 0;
	int input_format = 0, color_space = 0, quality = 0, prescale_table = 0, frame_number = 0, progressive = 0, version = 0;
	int transform_type = 0, num_spatial = 0, num_wavelets = 0, interlaced_flags = 0;
	ParsedBand lowpass[kMaxChannels];                     // raw 16-bit big-endian pairs
	ParsedBand high[kMaxChannels][kNumLevels][kNumBands]; // [ch][wavelet index][band 1..3]
	uint32_t metadata_offset = 0, metadata_bytes = 0;     // first metadata chunk
	size_t size = 0;                                      // bytes handed to parse_sample
};
// Returns 0 on success, 1 when the header parsed but the data ends early (header sniffing), <0 on malformed input.
int parse_sample(const uint8_t *data, size_t size, ParsedSample *out);
// Bias the reference decoder adds to every lowpass coefficient while unpacking it (Codec/decoder.c:12240-12290 fast path for
// even widths, :12468-12545 bit-serial path for odd widths): depends on the sample precision and the output pixel format.
int lowpass_bias(int precision, int lowpass_width, int out_pixel_kind, int channel = 0);
// Host VLC decode of one band into a zeroed band. Returns 0 on success.
int vlc_decode_band(const uint8_t *data, size_t bytes, int width, int height, int pitch, int quant, int codebook, int16_t *band);
// What the reference does to a decoded difference band (DecodeBandFSM16sNoGapWithPeaks decoder.c:19809 + :20822): coefficients beyond the peak
// level take their values from the peak table, then every row becomes its running sum.  peaks may be NULL (level 0).
void finish_difference_band(int16_t *band, int width, int height, int pitch, const uint8_t *peaks, size_t peak_bytes, int peak_level);

} // namespace cfhd
