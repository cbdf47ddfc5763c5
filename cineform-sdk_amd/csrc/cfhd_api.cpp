// cfhd_api.cpp -- the CFHD_* C ABI (include/cfhd_amd.h) on top of the GPU core.
//
// Mirrors the behaviour of the reference's SDK layer for the hot path:
//   EncoderSDK/CFHDEncoder.cpp + SampleEncoder.cpp (sync encode, metadata handling :744-939),
//   EncoderSDK/CFHDEncoderPool.cpp + EncoderPool.cpp + AsyncEncoder.cpp (async pool, FIFO completion),
//   DecoderSDK/CFHDDecoder.cpp + SampleDecoder.cpp (decode), DecoderSDK/CFHDMetadata.cpp (sample metadata access).
// The reference's CPU thread pool is replaced by HIP-stream frame slots; host threads only submit launches and hand samples back
// (with CFHD_AMD_ENTROPY=host they also run the entropy stage).
//
// One translation unit in parts (split in round 6; the parts share the handle types of an unnamed namespace and are #included below in this order):
//   cfhd_api_params.h    pixel formats, EncodeParams (what CFHD_PrepareToEncode derives), encoder-side metadata handle
//   cfhd_api_gather.h    calls that overlap share launches (Gatherer, EncodeService / DecodeService), the decoders-at-work signal
//   cfhd_api_handles.h   Encoder, EncoderPool + workers, SampleBuffer, Decoder, DecMetadata, the dealing of units to devices
//   cfhd_api_encoder.inc / _pool.inc / _decoder.inc / _metadata.inc      the CFHD_* entry points
#include "../../include/cfhd_amd.h"
#include "cfhd_core.h"
#include "cfhd_bitstream.h"
#include "cfhd_device.h"
#include "cfhd_params.h"
#include "cfhd_metadata.h"
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <vector>
#include <deque>
#include <memory>
#include <mutex>
#include <condition_variable>
#include <thread>
#include <atomic>
#include <random>

using namespace cfhd;

enum {
	ERR_OKAY = 0, ERR_INVALID_ARGUMENT = 1, ERR_OUTOFMEMORY = 2, ERR_BADFORMAT = 3, ERR_BADSCALING = 4, ERR_BADSAMPLE = 5, ERR_INTERNAL = 6,
	ERR_METADATA_END = 9, ERR_UNEXPECTED = 10, ERR_BAD_RESOLUTION = 11, ERR_NOT_FINISHED = 13, ERR_ENCODING_NOT_STARTED = 14, ERR_CODEC_ERROR = 2048,
};

#define FOURCC_BE(a, b, c, d) (((uint32_t)(a) << 24) | ((uint32_t)(b) << 16) | ((uint32_t)(c) << 8) | (uint32_t)(d))
static const uint32_t FMT_YUY2 = FOURCC_BE('Y', 'U', 'Y', '2'), FMT_2VUY = FOURCC_BE('2', 'v', 'u', 'y'), FMT_YUYV = FOURCC_BE('y', 'u', 'y', 'v'),
                      FMT_RG48 = FOURCC_BE('R', 'G', '4', '8'), FMT_B64A = FOURCC_BE('b', '6', '4', 'a'), FMT_BYR4 = FOURCC_BE('B', 'Y', 'R', '4'), FMT_YU64 = FOURCC_BE('Y', 'U', '6', '4'), FMT_V210 = FOURCC_BE('v', '2', '1', '0'), FMT_RG24 = FOURCC_BE('R', 'G', '2', '4'), FMT_BGRA = FOURCC_BE('B', 'G', 'R', 'A'), FMT_BGRa = FOURCC_BE('B', 'G', 'R', 'a'),
                      FMT_R210 = FOURCC_BE('r', '2', '1', '0'), FMT_DPX0 = FOURCC_BE('D', 'P', 'X', '0'), FMT_AB10 = FOURCC_BE('A', 'B', '1', '0'), FMT_AR10 = FOURCC_BE('A', 'R', '1', '0'),
                      FMT_RG30 = FOURCC_BE('R', 'G', '3', '0'), FMT_BYR5 = FOURCC_BE('B', 'Y', 'R', '5'), FMT_RG64 = FOURCC_BE('R', 'G', '6', '4');      // (AJA's name for the AB10 word layout: same pixels, its own colour format code in the sample header)

namespace {
#include "cfhd_api_params.h"
#include "cfhd_api_gather.h"
#include "cfhd_api_handles.h"

} // namespace

namespace cfhd {
int front_end_params(int width, int height, uint32_t pixel_format, int encoded_format, uint32_t encoding_flags, int quality, FrontEndParams *out)
{
	EncodeParams p;
	const int rc = make_params(p, width, height, pixel_format, encoded_format, encoding_flags, quality);
	if (rc) return rc;
	out->pixel_kind = p.pixel_kind; out->encoded_format = p.encoded_format; out->pixel_bytes = pixel_bytes_of(p.pixel_kind);
	out->color_format = p.pixel_format == FMT_RG30 ? 122 : color_format_of(p.pixel_kind); out->color_space = p.color_space; out->quality = p.quality; out->progressive = p.progressive;
	out->plan = p.plan; out->static_quantizer = quantizer_is_static(p);
	return 0;
}
}

extern "C" {

int cfhd_amd_device_count(void) { return device_count(); }
void cfhd_amd_set_clip_guid(const unsigned char guid[16]) { meta_fix_guid(guid); }
const char *cfhd_amd_last_error(void) { return device_last_error(); }
// Extension: a caller that reuses its frame / output buffers may page-lock them once; CFHD_EncodeSample, the encoder pool and
// CFHD_DecodeSample then DMA between them and HBM directly instead of staging through the library's own pinned memory.  The caller
// keeps the buffer alive and unregisters it before freeing it.  Buffers that were never registered work as before.
int cfhd_amd_register_host_buffer(void *buffer, size_t bytes) { return host_buffer_register(buffer, bytes); }
int cfhd_amd_unregister_host_buffer(void *buffer) { return host_buffer_unregister(buffer); }

#include "cfhd_api_encoder.inc"
#include "cfhd_api_pool.inc"
#include "cfhd_api_decoder.inc"
#include "cfhd_api_metadata.inc"

} // extern "C"
