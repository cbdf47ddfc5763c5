// cfhd_gop.h -- the two-frame group of pictures: CFHD_ENCODING_FLAGS_YUV_2FRAME_GOP (Common/CFHDTypes.h:254), TRANSFORM_TYPE_FIELDPLUS
// in the reference (Codec/encoder.c:1587).  SURVEY.md section 8, row f3.
//
// Two consecutive frames share one sample.  Per channel the reference builds six wavelets (encoder.c:8431 FinishFieldPlusTransformQuant):
//   w[0], w[1]  level 1 of frame 0 / frame 1: the spatial 2/6 transform every intra frame gets (wavelet.c:2823), highpass bands quantized
//   w[2]        temporal: low = sat16(LL0 + LL1), high = sat16(LL1 - LL0) of the two level-1 lowpass bands (temporal.c:498); two bands, nothing coded
//   w[3]        spatial 2/6 of the temporal highpass band (its lowpass band is coded too: raw 16-bit words, subband 7)
//   w[4], w[5]  spatial 2/6 of the temporal lowpass band, and of w[4]'s lowpass band; w[5]'s lowpass band is the sample's lowpass (raw 16 bit)
// 17 subbands: 0 = LL of w[5]; 1-3 w[5]; 4-6 w[4]; 7-10 w[3]; 11-13 w[1]; 14-16 w[0] (quantize.c:3480).
// The first CFHD_EncodeSample call of a sequence returns a 40-byte sequence header, every second frame the group, the frames in between a
// 24-byte P-frame header (encoder.c:2927, :3282-3380); the decoder hands out frame 0 for the group and frame 1 for the P-frame sample
// (decoder.c:11180 DecodeSampleGroup, :11426 DecodeSampleFrame).
#pragma once
#include "cfhd_core.h"
#include "cfhd_bitstream.h"

namespace cfhd {

enum { kGopWavelets = 6, kGopSubbands = 17 };

struct GopWavelet {
	int type, level, nbands;                 // WAVELET_TYPE_* of the reference (5 frame, 4 temporal, 3 spatial), pyramid level, 2 or 4 bands
	int width, height, pitch;                // band dimensions; pitch in coefficients (rows padded to 8, wavelet.c:439-442)
	size_t offset[4];                        // band b inside the channel-interleaved group pyramid (int16 elements)
	int quant[4], scale[4];
	int prescale;                            // of the transform that PRODUCES this wavelet (wavelet.c:1710 SetTransformPrescale, FIELDPLUS)
};
struct GopChannel { int width, height; GopWavelet w[kGopWavelets]; };
struct GopPlan {
	int width = 0, height = 0, display_height = 0;      // height rounded up to a multiple of 8 (encoder.c:1569)
	int num_channels = 3, precision = 10, midpoint_prequant = 2;
	int pixel_kind = PIX_YUY2;
	// CFHD_ENCODING_FLAGS_YUV_INTERLACED on top of the group flag: level 1 of both frames is the frame transform of interlaced intra frames (Codec/encoder.c:2950-2979
	// TransformForwardFrameYUV into w[0] / w[1]); everything above it is the same.  The horizontal-lowpass / temporal-highpass band of both frame wavelets (band 2:
	// subbands 12 and 15) is difference coded in code set 18 with a peak table (encoder.c:6143-6154 SetCodingFlags), as subband 8 of an interlaced intra frame.
	bool interlaced = false;
	GopChannel ch[3];
	size_t coeff_elems = 0;                  // int16 elements of one group's pyramid
	size_t sample_buffer_bytes = 0;          // the reference's sample buffer (its BITSTREAM block length): the encoder stops coding the frame wavelets' bands at 80% of it
	// the level-1 transforms run through the kernels of the intra path: two ordinary frame plans whose level-1 bands alias w[0] / w[1]
};

// the band of a frame wavelet that an interlaced group codes as differences along the row, in code set 18, with a peak table (subbands 12 and 15)
inline bool gop_band_is_difference_coded(const GopPlan &plan, int wavelet, int band) { return plan.interlaced && wavelet < 2 && band == 2; }
// false: geometry the group transform does not serve (the same rule as build_frame_plan: chroma must halve on whole pairs four times here)
bool build_gop_plan(GopPlan *plan, int width, int height, int pixel_kind, bool interlaced = false);
// Quantizer tables of the group (QuantizationSetQuality quantize.c:186 + SetTransformQuantization :2865, :3480 + SetTransformScale wavelet.c:7142).
// The reference runs the first on every CFHD_EncodeSample call and the second only on the call that opens a group (encoder.c:2880-2905), both with the size of
// the last key sample -- the 40-byte sequence header counts as one (encoder.c:3414) -- in `st`: deal = true for the opening call (the tables are written into the
// plan), false for the call that completes the group (only the FILMSCAN2/3 limiter in `st` moves).
bool derive_gop_quantization(GopPlan *plan, int quality, QuantState *st, float framerate = 0.0f, bool deal = true);

// The group sample (codec.c:835 PutVideoGroupHeader + encoder.c:7461 EncodeQuantizedGroup + :8078 EncodeQuantizedFieldPlusTransform) from the
// group's coefficient pyramid; 0 on overflow.
size_t write_group_sample(const GopPlan &plan, const SampleHeaderInfo &hdr, const int16_t *coeffs, uint8_t *out, size_t cap);
// The same sample as a template for the GPU entropy stage: holes carry (channel, wavelet index as `level`, band); kind 0 = raw 16-bit band (w[5] / w[3] band 0).
void build_group_template(const GopPlan &plan, const SampleHeaderInfo &hdr, SampleTemplate *t);
bool gop_sample_may_zero_bands(const GopPlan &plan, size_t bytes);
// The 40-byte sequence header a sequence starts with (codec.c:736) and the 24-byte sample of a group's second frame (codec.c:1258).
size_t write_sequence_header(const GopPlan &plan, int input_format, uint8_t *out, size_t cap);
size_t write_pframe_sample(const GopPlan &plan, uint32_t frame_number, uint8_t *out, size_t cap);

// ---- decoder side ----
struct ParsedGroup {
	int sample_type = 0;                     // 2 group, 1 frame (the P-frame header), 7 sequence header
	int width = 0, height = 0, display_height = 0, num_channels = 0, precision = 0, input_format = 0, frame_number = 0;
	int progressive = 0;                     // the reference's default (codec.c:263); set only by TAG_SAMPLE_FLAGS (decoder.c:13397): an interlaced group carries no such tag
	ParsedBand lowpass[3];                   // w[5]'s lowpass band, raw 16-bit big-endian
	ParsedBand band[3][kGopWavelets][4];     // [channel][wavelet][band]: coded bands (w[3] band 0: raw 16-bit, codebook -1)
};
// 0: a group; 1: a sample of another kind (sample_type says which); < 0: malformed
int parse_group_sample(const uint8_t *data, size_t size, ParsedGroup *out);

} // namespace cfhd
