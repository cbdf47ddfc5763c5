// cfhd_metadata.h -- metadata tuple blocks carried opaquely inside samples (tag, size|type, payload padded to 4 bytes).
// Layout and replace/append rules follow Codec/encoder.c:447 AddMetadata and Codec/metadata.c:70 MetadataFind.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <vector>

namespace cfhd {

#define CFHD_FOURCC(a, b, c, d) ((uint32_t)(a) | ((uint32_t)(b) << 8) | ((uint32_t)(c) << 16) | ((uint32_t)(d) << 24))
enum : uint32_t {
	MTAG_CLIP_GUID = CFHD_FOURCC('G', 'U', 'I', 'D'), MTAG_ENCODE_DATE = CFHD_FOURCC('D', 'A', 'T', 'E'), MTAG_ENCODE_TIME = CFHD_FOURCC('T', 'I', 'M', 'E'),
	MTAG_TIMECODE = CFHD_FOURCC('T', 'I', 'M', 'C'), MTAG_TIMECODE_BASE = CFHD_FOURCC('T', 'I', 'M', 'B'), MTAG_UNIQUE_FRAMENUM = CFHD_FOURCC('U', 'F', 'R', 'M'),
	MTAG_FREESPACE = CFHD_FOURCC('F', 'R', 'E', 'E'), MTAG_REGISTRY_NAME = CFHD_FOURCC('R', 'E', 'G', 'N'), MTAG_REGISTRY_VALUE = CFHD_FOURCC('R', 'E', 'G', 'V'),
	MTAG_NAME = CFHD_FOURCC('N', 'A', 'M', 'E'), MTAG_VALUE = CFHD_FOURCC('V', 'A', 'L', 'U'),
};

typedef std::vector<uint8_t> MetaBlock;

bool meta_add(MetaBlock &block, uint32_t tag, unsigned char type, uint32_t size, const void *data);
// Returns a pointer to the payload of the first tuple with this tag, or NULL.
const uint8_t *meta_find(const uint8_t *block, size_t block_size, uint32_t tag, uint32_t *size_out, unsigned char *type_out);
// Drops tuples of the hidden type 'h' (Codec/encoder.c:8906 RemoveHiddenMetadata).
void meta_remove_hidden(MetaBlock &block);

// Clip GUID of a new metadata block: random (RFC 4122 version 4) unless pinned with meta_fix_guid (cfhd_amd_set_clip_guid, for tests).
void meta_new_guid(unsigned char out[16]);
void meta_fix_guid(const unsigned char guid[16] /* NULL: random again */);

// The per-encoder metadata state machine (EncoderSDK/SampleEncoder.cpp:744-939 HandleMetadata): before every frame the encoder makes
// sure the global block carries a clip GUID, today's encode date / time, a timecode that advances by one frame per sample and a
// unique frame number that increases per sample.
struct MetaState {
	MetaBlock global, local;
	int last_timecode_base = 0, last_timecode_frame = -1, last_unique_frame = -1;
	void handle();
};

} // namespace cfhd
