// cfhd_core.h -- shared host/device definitions of the MI355X CineForm core.
//
// Geometry ("FramePlan") of one frame's coefficient pyramid in HBM, the quantizer tables and the
// entropy-code tables.  Everything here is plain data that host code computes once per
// encoder/decoder and hands to the HIP kernels by value or through constant buffers.
#pragma once
#include <stdint.h>
#include <stddef.h>

#ifndef CFHD_HD
#if defined(__HIPCC__)
#define CFHD_HD __host__ __device__
#else
#define CFHD_HD
#endif
#endif

namespace cfhd {

enum { kMaxChannels = 4, kNumLevels = 3, kNumBands = 4, kMaxFrameDim = 16384 };

// Internal pixel formats handled by the unpack (encode) / pack (decode) kernels.
enum PixelKind : int {
	PIX_NONE = 0,
	PIX_YUY2,      // 8-bit 4:2:2  Y0 U Y1 V   (reference COLOR_FORMAT_YUYV = 2)
	PIX_2VUY,      // 8-bit 4:2:2  U Y0 V Y1   (reference COLOR_FORMAT_UYVY = 1)
	PIX_RG48,      // 16-bit RGB 4:4:4          (COLOR_FORMAT_RGB48 = 120)
	PIX_B64A,      // 16-bit ARGB 4:4:4:4       (COLOR_FORMAT_BGRA64 = 30)
	PIX_BYR4,      // 16-bit Bayer              (COLOR_FORMAT_BYR4 = 104)
	PIX_YU64,      // 16-bit 4:2:2  Y0 C1 Y1 C2  (COLOR_FORMAT_YU64 = 12; encoder input only)
	PIX_V210,      // 10-bit 4:2:2, six pixels in four 32-bit words (COLOR_FORMAT_V210 = 10; encoder input only)
	PIX_RG24,      // 8-bit B, G, R bytes, bottom row first (COLOR_FORMAT_RGB24 = 7; encoder input only, to RGB 4:4:4)
	PIX_BGRA,      // 8-bit B, G, R, A bytes, bottom row first ('BGRA', format code 32; encoder input only, to RGB 4:4:4: alpha dropped)
	PIX_BGRa,      // the same, top row first ('BGRa', format code 9)
	PIX_R210,      // 10-bit RGB in a big-endian 32-bit word, R bits 20-29, G 10-19, B 0-9 (COLOR_FORMAT_R210 = 123; encoder input only, to RGB 4:4:4)
	PIX_DPX0,      // big-endian, R 22-31, G 12-21, B 2-11 (128)
	PIX_AB10,      // little-endian, R 0-9, G 10-19, B 20-29 (125)
	PIX_AR10,      // little-endian, R 20-29, G 10-19, B 0-9 (124)
	PIX_RG64,      // 16-bit words R, G, B, A (COLOR_FORMAT_RG64 = 121; encoder input only: to RGBA 4:4:4:4, RGB 4:4:4 or YUV 4:2:2 like b64a, quantized like RG48)
	PIX_BYR5,      // 12-bit Bayer, per row pair the four components as runs of high bytes, then their low nibbles (COLOR_FORMAT_BYR5 = 105; encoder input only)
};

// ENCODED_FORMAT_* values as written into the bitstream (Codec/codec.h)
enum EncodedFormat : int { ENC_YUV422 = 1, ENC_BAYER = 2, ENC_RGB444 = 3, ENC_RGBA4444 = 4 };

struct BandDesc {
	int width, height;     // coefficients
	int pitch;             // row stride in int16 elements (width rounded up to 8: Codec/wavelet.c:439-442)
	uint32_t offset;       // element offset inside the frame's coefficient buffer
	int quant;             // divisor applied by the encoder (1 for every LL)
	int scale;             // informational scale written into the band header
};

struct ChannelPlan {
	int width, height;                       // channel plane at full resolution
	BandDesc band[kNumLevels][kNumBands];    // [wavelet index 0..2 = level 1..3][LL,LH,HL,HH]
};

struct FramePlan {
	int width, height;           // encoded dimensions (height rounded up to a multiple of 8)
	int display_height;
	int num_channels;
	int precision;               // 10 or 12 bits
	int encoded_format;          // EncodedFormat
	int pixel_kind;              // PixelKind of the packed frame
	int color_matrix;            // deep RGB input encoded as 4:2:2: the conversion matrix (0 computer-systems 709, 1 video 709, 2 computer 601, 3 video 601; frame.c:6803)
	int interlaced;              // level 1 is the field ("frame") transform: temporal 2-tap between the two fields, horizontal 2/6 (encoder.c:2093)
	int prescale[kNumLevels];    // per wavelet index (Codec/wavelet.c:1710)
	int midpoint_prequant;       // Codec/quantize.c:183,211-213
	ChannelPlan ch[kMaxChannels];
	uint32_t coeff_elems;        // int16 elements of the whole pyramid (all channels)
	uint32_t final_elems;        // leading part that holds the bands that get entropy coded
};

// ---- entropy tables (code set 17 = codebook 1, cubic companding; code set 18 = codebook 2, linear) ----
struct RunCode { uint32_t bits; uint8_t size; uint16_t count; };
struct EntropyTables {
	uint32_t value_code[2048];     // size<<27 | codeword, indexed by 11-bit two's complement value (Codec/vlc.h:71-74)
	uint32_t run_bits[3072];       // composite zero-run codewords (Codec/codebooks.c:499-582)
	uint8_t  run_size[3072];
	uint16_t run_count[3072];      // zeros actually covered by the composite code
	uint32_t band_end_bits;
	int      band_end_size;
	// decoder: direct lookup on the next kDecBits bits of the stream
	enum { kDecBits = 12 };
	// entry: bits 0..4 = code length (0 => longer than kDecBits, take the slow path),
	//        bits 5..15 = zero run (0 for a magnitude), bits 16..31 = expanded magnitude (0 for runs / zero)
	uint32_t dec_lut[1 << kDecBits];
	uint16_t mag_expand[256];      // magnitude after undoing the companding curve
};

const EntropyTables *entropy_tables(int codebook /*1 = cs17 cubic, 2 = cs18 linear*/);

// The base code words of a code set, one entry per symbol: kind 0 zero run (payload = zeros), 1 magnitude (payload = index 1..255 into
// EntropyTables::mag_expand; a sign bit follows the code word), 2 band end marker.  What the decoder tables are built from.
struct RawCode { uint32_t bits; int len; int kind; int payload; };
int raw_codes(int codebook, RawCode *out /* 300 entries */);

// ---- quantizer ----
struct QuantState {            // mirrors the reference's cross-frame quantizer state (Codec/quantize.h QUANTIZER)
	int overbitrate;
	int FSratelimiter;
	int64_t lastgopbitcount;
};
// Fills plan->prescale, midpoint_prequant and every band's quant/scale.
void derive_quantization(FramePlan *plan, int quality, bool progressive, float framerate, QuantState *state);

// Builds the pyramid geometry for the given encoded dimensions.
bool build_frame_plan(FramePlan *plan, int width, int height, int pixel_kind, int encoded_format);

// Which GPU serves unit i (worker i of an encoder pool, the i-th decoder handle of the process) on a node with `ndevices` GPUs.
//   pinned (CFHD_AMD_DEVICE or LOCAL_RANK is set: one process per GPU, the launcher dealt the devices) -> -1 for everybody: the process default
//   list   (CFHD_AMD_POOL_DEVICES, e.g. "0,1,2,3" or "0,0": devices may repeat)                      -> list[i % length], each taken modulo ndevices
//   else                                                                                             -> i % ndevices (round robin, as the reference's
//          pool deals frames to its encoder threads, EncoderSDK/EncoderPool.cpp:281-291)
// Frames keep their submission order whatever device encodes them (the pool's FIFO, EncoderPool.cpp:297-380).
int unit_device(int i, int ndevices, const char *pinned_env, const char *list_env);
// Default encode curve of the Bayer input path (log base 90 over 14-bit linear input, scaled to `precision` bits), frame.c:5219-5235.
enum { kBayerCurveBits = 14 };
void build_bayer_log90_curve(int precision, uint16_t *curve /* 1 << kBayerCurveBits entries */);
// The decoder's way back for BYR4 output (decoder.c:10714 BYR4LinearRestore, log base 90): index = 16-bit value >> 2.
void build_bayer_linear_restore_curve(uint16_t *curve /* 1 << kBayerCurveBits entries */);

static inline int align_up(int x, int a) { return (x + a - 1) / a * a; }

// Block lists of the quantized level-1 bands (dev::FwdBlockLists in cfhd_kernels.h; k_fwd_yuv422_strip_blocks writes them, k_ent_count_blocks reads them): every band
// row is cut into chunks of kBlockChunkCols coefficients, one 64-bit occupancy mask per chunk.  mask_base[c][b]: the first chunk of band (c, b) in a frame's mask
// array (chunks by band row, then by position in the row); returns the masks per frame.
enum { kBlockChunkCols = 496 };
static inline int block_list_layout(const FramePlan &plan, int mask_base[kMaxChannels][kNumBands])
{
	int at = 0;
	for (int c = 0; c < plan.num_channels; c++)
		for (int b = 0; b < kNumBands; b++) {
			const BandDesc &bd = plan.ch[c].band[0][b];
			mask_base[c][b] = b ? at : -1;
			if (b) at += bd.height * ((bd.pitch + kBlockChunkCols - 1) / kBlockChunkCols);
		}
	return at;
}

// The decode side of the same idea (round 4): the tile pass of the entropy decoder (k_dec_tiles) leaves the dequantized level-1 highpass bands as block lists and the
// inverse level-1 strip kernel gathers them.  Here a chunk is 64 consecutive blocks of the band's flat raster (512 coefficients, pitch padding included -- what a
// tile of the decoder is made of), the listed blocks of chunk k sit compacted at the band's own blocks 64 k .. in the pyramid, and there is one 64-bit mask per chunk.
enum { kDecChunkCoeffs = 512 };
static inline int dec_block_list_layout(const FramePlan &plan, int mask_base[kMaxChannels][kNumBands])
{
	int at = 0;
	for (int c = 0; c < plan.num_channels; c++)
		for (int b = 0; b < kNumBands; b++) {
			const BandDesc &bd = plan.ch[c].band[0][b];
			mask_base[c][b] = b ? at : -1;
			if (b) at += (bd.height * bd.pitch + kDecChunkCoeffs - 1) / kDecChunkCoeffs;
		}
	return at;
}

} // namespace cfhd
