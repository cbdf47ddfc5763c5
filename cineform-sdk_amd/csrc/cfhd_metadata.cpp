#include "cfhd_metadata.h"
#include <string.h>

namespace cfhd {

static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline void wr32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }
static inline uint32_t padded(uint32_t n) { return (n + 3) & ~3u; }

bool meta_add(MetaBlock &b, uint32_t tag, unsigned char type, uint32_t size, const void *data)
{
	if (!data || !size) return false;
	const uint32_t typesize = ((uint32_t)type << 24) | (size & 0xffffff);
	const uint32_t alloc = 8 + padded(size);
	if (b.size() + alloc >= 65500u * 4) return false;
	auto write_at = [&](size_t pos) {
		wr32(&b[pos], tag); wr32(&b[pos + 4], typesize);
		memcpy(&b[pos + 8], data, size);
		memset(&b[pos + 8 + size], 0, padded(size) - size);
	};
	// Tags whose last character is upper case are unique: replace in place (same padded size) or remove the old tuple.
	const bool unique = (tag >> 24) < 'a' && tag != MTAG_FREESPACE && tag != MTAG_REGISTRY_NAME && tag != MTAG_REGISTRY_VALUE &&
	                    tag != MTAG_NAME && tag != MTAG_VALUE;
	if (unique) {
		size_t pos = 0;
		while (pos + 8 <= b.size()) {
			uint32_t t = rd32(&b[pos]), ts = rd32(&b[pos + 4]);
			uint32_t len = padded(ts & 0xffffff);
			if (t == tag) {
				if (len == padded(size)) { write_at(pos); return true; }
				b.erase(b.begin() + pos, b.begin() + pos + 8 + len);
				break;
			}
			pos += 8 + len;
		}
	}
	// Reuse a FREE tuple that is large enough, else append.
	size_t pos = 0;
	while (pos + 8 <= b.size()) {
		uint32_t t = rd32(&b[pos]), ts = rd32(&b[pos + 4]);
		uint32_t len = ts & 0xffffff;
		if (t == MTAG_FREESPACE && len >= size) {
			int freebytes = (int)len - (int)padded(size) - 8;
			write_at(pos);
			if (freebytes > 16) { size_t p2 = pos + 8 + padded(size); wr32(&b[p2], MTAG_FREESPACE); wr32(&b[p2 + 4], ((uint32_t)'c' << 24) | (uint32_t)freebytes); }
			return true;
		}
		pos += 8 + padded(len);
	}
	size_t at = b.size();
	b.resize(at + alloc);
	write_at(at);
	return true;
}

const uint8_t *meta_find(const uint8_t *block, size_t n, uint32_t tag, uint32_t *size_out, unsigned char *type_out)
{
	size_t pos = 0;
	while (block && pos + 8 <= n) {
		uint32_t t = rd32(block + pos), ts = rd32(block + pos + 4);
		uint32_t len = ts & 0xffffff;
		if (t == 0) break;
		if (pos + 8 + len > n) break;
		if (t == tag) { if (size_out) *size_out = len; if (type_out) *type_out = (unsigned char)(ts >> 24); return block + pos + 8; }
		pos += 8 + padded(len);
	}
	return nullptr;
}

void meta_remove_hidden(MetaBlock &b)
{
	size_t pos = 0;
	while (pos + 8 <= b.size()) {
		uint32_t ts = rd32(&b[pos + 4]);
		uint32_t entry = 8 + padded(ts & 0xffffff);
		if ((ts >> 24) == 'h' && pos + entry <= b.size()) b.erase(b.begin() + pos, b.begin() + pos + entry);
		else pos += entry;
	}
}

} // namespace cfhd
