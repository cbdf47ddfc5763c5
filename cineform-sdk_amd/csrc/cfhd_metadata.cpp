#include "cfhd_metadata.h"
#include <string.h>
#include <stdio.h>
#include <time.h>
#include <mutex>
#include <random>

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

namespace {
std::mutex g_guid_mutex;
bool g_guid_fixed = false;
unsigned char g_guid[16];
}

void meta_new_guid(unsigned char out[16])
{
	std::lock_guard<std::mutex> lk(g_guid_mutex);
	if (g_guid_fixed) { memcpy(out, g_guid, 16); return; }
	std::random_device rd;
	for (int i = 0; i < 16; i += 4) { uint32_t r = rd(); memcpy(out + i, &r, 4); }
	out[6] = (out[6] & 0x0f) | 0x40; out[8] = (out[8] & 0x3f) | 0x80;    // RFC 4122 version 4
}

void meta_fix_guid(const unsigned char guid[16])
{
	std::lock_guard<std::mutex> lk(g_guid_mutex);
	if (guid) { memcpy(g_guid, guid, 16); g_guid_fixed = true; } else g_guid_fixed = false;
}

void MetaState::handle()
{
	if (global.empty()) { unsigned char g[16]; meta_new_guid(g); meta_add(global, MTAG_CLIP_GUID, 'G', 16, g); }
	time_t clock = time(NULL);
	struct tm tmv; localtime_r(&clock, &tmv);
	char datestr[32], timestr[32], tmp[32];
	snprintf(datestr, sizeof(datestr), "%04d-%02d-%02d", tmv.tm_year + 1900, tmv.tm_mon + 1, tmv.tm_mday);
	snprintf(timestr, sizeof(timestr), "%02d:%02d:%02d", tmv.tm_hour, tmv.tm_min, tmv.tm_sec);
	meta_add(global, MTAG_ENCODE_DATE, 'c', 10, datestr);
	meta_add(global, MTAG_ENCODE_TIME, 'c', 8, timestr);

	bool in_local = false;
	uint32_t sz; unsigned char ty;
	const uint8_t *data = meta_find(global.data(), global.size(), MTAG_TIMECODE, &sz, &ty);
	if (!data) {
		data = meta_find(local.data(), local.size(), MTAG_TIMECODE, &sz, &ty);
		if (!data) {
			last_timecode_base = 24;
			last_timecode_frame = tmv.tm_hour * 3600 * 24 + tmv.tm_min * 60 * 24 + tmv.tm_sec * 24;
			snprintf(tmp, sizeof(tmp), "%02d:%02d:%02d:00", tmv.tm_hour, tmv.tm_min, tmv.tm_sec);
			meta_add(global, MTAG_TIMECODE, 'c', 11, tmp);
		} else in_local = true;
	}
	if (data) {
		const char *tc = (const char *)data;
		int hours = (tc[0] - '0') * 10 + (tc[1] - '0'), mins = (tc[3] - '0') * 10 + (tc[4] - '0');
		int secs = (tc[6] - '0') * 10 + (tc[7] - '0'), frms = (tc[9] - '0') * 10 + (tc[10] - '0');
		if (last_timecode_base == 0) {
			const uint8_t *b = meta_find(local.data(), local.size(), MTAG_TIMECODE_BASE, &sz, &ty);
			if (!b) b = meta_find(global.data(), global.size(), MTAG_TIMECODE_BASE, &sz, &ty);
			last_timecode_base = b ? *b : 24;
			if (last_timecode_base == 0) last_timecode_base = 24;
		}
		int base = last_timecode_base;
		int framenum = hours * 3600 * base + mins * 60 * base + secs * base + frms;
		if (last_timecode_frame == -1) last_timecode_frame = framenum;
		else if (framenum == last_timecode_frame && base <= 30) {
			framenum = ++last_timecode_frame;
			frms = framenum % base; framenum /= base;
			secs = framenum % 60; framenum /= 60;
			mins = framenum % 60; framenum /= 60;
			hours = framenum % 60;
			snprintf(tmp, sizeof(tmp), "%02d:%02d:%02d:%02d", hours, mins, secs, frms);
			meta_add(in_local ? local : global, MTAG_TIMECODE, 'c', 11, tmp);
		}
	}
	in_local = false;
	data = meta_find(global.data(), global.size(), MTAG_UNIQUE_FRAMENUM, &sz, &ty);
	if (!data) {
		data = meta_find(local.data(), local.size(), MTAG_UNIQUE_FRAMENUM, &sz, &ty);
		if (!data) { last_unique_frame = 0; uint32_t v = 0; meta_add(global, MTAG_UNIQUE_FRAMENUM, 'L', 4, &v); }
		else in_local = true;
	}
	if (data) {
		int32_t n; memcpy(&n, data, 4);
		if (last_unique_frame == -1) last_unique_frame = n;
		else if (n <= last_unique_frame) { uint32_t v = (uint32_t)++last_unique_frame; meta_add(in_local ? local : global, MTAG_UNIQUE_FRAMENUM, 'L', 4, &v); }
	}
}

} // namespace cfhd
