// cfhd_entropy_jobs.h -- host-side construction of the tables the GPU entropy kernels consume (no HIP calls here, so the
// CPU emulation harness in tests/hipemu builds exactly the same tables as the device driver).
#pragma once
#include "cfhd_core.h"
#include "cfhd_bitstream.h"
#include "cfhd_gop.h"
#include "cfhd_entropy_kernels.h"
#include "cfhd_dec_kernels.h"
#include <vector>
#include <algorithm>
#include <string.h>

namespace cfhd {

enum { kEntTmplBytes = 6144, kEntWordHolesBytes = 1536, kEntHolesBytes = dev::ENT_MAX_HOLES * (int)sizeof(dev::EntHole), kEntMaxPatches = 96,
       kEntPatchBytes = kEntMaxPatches * (int)sizeof(dev::EntPatch), kEntTmplStride = kEntTmplBytes + kEntWordHolesBytes + kEntHolesBytes + kEntPatchBytes };

// Table 0: code set 17 (codebook 1), every band of a progressive intra frame (encoder.c:6120); table 1: code set 18 (codebook 2), the
// difference-coded band of an interlaced frame.
inline void ent_build_tables(dev::EntTables *h, int codebook = 1)
{
	const EntropyTables *t = entropy_tables(codebook);
	memcpy(h->value_code, t->value_code, sizeof(h->value_code));
	memcpy(h->run_bits, t->run_bits, sizeof(h->run_bits));
	for (int c = 0; c < 3072; c++) { h->run_count[c] = t->run_count[c]; h->run_size[c] = t->run_size[c]; }
	h->run_total[0] = 0;
	for (int c = 1; c < 3072; c++) h->run_total[c] = (uint16_t)(t->run_size[c] + h->run_total[c - t->run_count[c]]);
	h->band_end_bits = t->band_end_bits; h->band_end_size = t->band_end_size;
	for (int c = 0; c < 3072; c++) { h->run_pack[c].x = t->run_bits[c]; h->run_pack[c].y = (uint32_t)t->run_size[c] | ((uint32_t)t->run_count[c] << 8); }
}

// Two-level decode tables from the base Huffman codes (cfhd_tables.cpp keeps them in EntropyTables::dec_lut for 12 bits;
// here every code word up to 26 bits is resolved by table lookups only).  Returns the serialised dev::DecTables.
std::vector<uint32_t> build_dec_tables(int codebook);

// Tables of k_dec_index / k_dec_chain / k_dec_tiles (cfhd_dec_kernels.h) from the base code words of a code set.  false: the code set does
// not fit the table layout (cannot happen with code sets 17 and 18; checked, not assumed).
inline bool build_dec_index_tables(int codebook, dev::DecIdxTables *T)
{
	using namespace dev;
	memset(T, 0, sizeof(*T));
	RawCode codes[300];
	const int ncodes = raw_codes(codebook, codes);
	const EntropyTables *et = entropy_tables(codebook);
	for (int m = 0; m < 256; m++) { T->mag_expand[0][m] = entropy_tables(1)->mag_expand[m]; T->mag_expand[1][m] = entropy_tables(2)->mag_expand[m]; }
	(void)et;
	auto type_of = [](const RawCode &c) { return c.kind == 0 ? DX_T_RUN : (c.kind == 1 ? DX_T_VALUE : DX_T_END); };
	// first level: code words of up to 12 bits directly, longer ones through their 12-bit prefix
	uint32_t nlong = 0;
	int l2_of_prefix[1 << DX_K];
	for (int p = 0; p < (1 << DX_K); p++) l2_of_prefix[p] = -1;
	for (int i = 0; i < ncodes; i++) {
		const RawCode &c = codes[i];
		if (c.len <= DX_K) {
			if (c.kind == 2 || c.payload >= (1 << 11) || c.len > 12) return false;
			const uint32_t base = c.bits << (DX_K - c.len);
			const uint16_t e = (uint16_t)(c.len | (c.kind == 1 ? 16 : 0) | (c.payload << 5));
			for (uint32_t k = 0; k < (1u << (DX_K - c.len)); k++) T->sym12[base + k] = e;
		} else {
			if (c.len > DX_K + 2 * DX_L2_BITS) return false;
			const uint32_t p = c.bits >> (c.len - DX_K);
			if (l2_of_prefix[p] < 0) {
				if (nlong + (1u << DX_L2_BITS) > DX_LONG_MAX || nlong >= (1u << 11)) return false;
				l2_of_prefix[p] = (int)nlong; nlong += 1u << DX_L2_BITS;
				T->sym12[p] = (uint16_t)(16u | ((uint32_t)l2_of_prefix[p] << 5));       // length 0 + bit 4: escape
			}
		}
	}
	// second level: 7 more bits; code words of up to 19 bits end here, longer ones escape once more
	struct L3 { uint32_t key; int maxlen; uint32_t base; };
	std::vector<L3> l3;
	for (int i = 0; i < ncodes; i++) {
		const RawCode &c = codes[i];
		if (c.len <= DX_K + DX_L2_BITS) continue;
		const uint32_t key = c.bits >> (c.len - DX_K - DX_L2_BITS);
		auto it = std::find_if(l3.begin(), l3.end(), [&](const L3 &x) { return x.key == key; });
		if (it == l3.end()) l3.push_back(L3{ key, c.len, 0 }); else if (c.len > it->maxlen) it->maxlen = c.len;
	}
	for (L3 &x : l3) {
		const int nb = x.maxlen - DX_K - DX_L2_BITS;
		if (nlong + (1u << nb) > DX_LONG_MAX) return false;
		x.base = nlong; nlong += 1u << nb;
		const uint32_t p = x.key >> DX_L2_BITS;
		T->long_tab[(uint32_t)l2_of_prefix[p] + (x.key & ((1u << DX_L2_BITS) - 1u))] = (uint32_t)nb | ((uint32_t)DX_T_ESCAPE << 5) | (x.base << 8);
	}
	for (int i = 0; i < ncodes; i++) {
		const RawCode &c = codes[i];
		if (c.len <= DX_K) continue;
		// what the index walk needs of a long code word: how far it moves the bit position (sign bit included; 0 stops the walk: the band end marker) and the raster position
		const uint32_t entry = (c.kind == 2 ? 0u : (uint32_t)c.len + (c.kind == 1 ? 1u : 0u)) | ((uint32_t)type_of(c) << 5) | ((c.kind == 0 ? (uint32_t)c.payload : (c.kind == 1 ? 1u : 0u)) << 8);
		if (c.len <= DX_K + DX_L2_BITS) {
			const uint32_t p = c.bits >> (c.len - DX_K);
			const int spare = DX_K + DX_L2_BITS - c.len;
			const uint32_t base = (uint32_t)l2_of_prefix[p] + ((c.bits & ((1u << (c.len - DX_K)) - 1u)) << spare);
			for (uint32_t k = 0; k < (1u << spare); k++) T->long_tab[base + k] = entry;
		} else {
			const uint32_t key = c.bits >> (c.len - DX_K - DX_L2_BITS);
			const L3 &x = *std::find_if(l3.begin(), l3.end(), [&](const L3 &y) { return y.key == key; });
			const int nb = x.maxlen - DX_K - DX_L2_BITS, rest = c.len - DX_K - DX_L2_BITS, spare = nb - rest;
			const uint32_t base = x.base + ((c.bits & ((1u << rest) - 1u)) << spare);
			for (uint32_t k = 0; k < (1u << spare); k++) T->long_tab[base + k] = entry;
		}
	}
	T->nlong = nlong;
	// several code words per lookup: as many whole ones (sign bits included) as the 12-bit window holds
	for (uint32_t win = 0; win < (1u << DX_K); win++) {
		int used = 0; uint32_t count = 0;
		for (;;) {
			const uint32_t rest = (win << used) & ((1u << DX_K) - 1u);
			const uint16_t e = T->sym12[rest];
			const int len = e & 15, total = len + ((e & 16) ? 1 : 0);
			if (!len || total > DX_K - used) break;
			const uint32_t add = (e & 16) ? 1u : (uint32_t)(e >> 5);
			if (count + add > 4095u) break;
			used += total; count += add;
		}
		T->cnt12[win] = (uint16_t)(used | (count << 4));
	}
	// k_dec_tiles: up to two values and the zero runs around them from an 11-bit window
	for (uint32_t win = 0; win < (1u << DX_KM); win++) {
		int used = 0, pre = 0, mid = 0, post = 0, v1 = 0, v2 = 0;
		for (;;) {
			const uint32_t rest = ((win << (DX_K - DX_KM)) << used) & ((1u << DX_K) - 1u);     // remaining bits, left aligned in 12 (the 12th bit of the window is unknown: 0)
			const uint16_t e = T->sym12[rest];
			const int len = e & 15, total = len + ((e & 16) ? 1 : 0);
			if (!len || total > DX_KM - used) break;
			if (e & 16) {
				if (v2) break;
				const int m = e >> 5;
				if (T->mag_expand[0][m] != m || T->mag_expand[1][m] != m || m > 0x7fff) return false;                     // the table relies on magnitude = index for such short code words
				const int negative = (int)((rest >> (DX_K - len - 1)) & 1u);
				if (!v1) v1 = negative ? -m : m; else v2 = negative ? -m : m;
			} else {
				const int run = e >> 5;
				int *slot = v2 ? &post : (v1 ? &mid : &pre);
				if (*slot + run > (slot == &pre ? 0xfff : 0xff)) break;
				*slot += run;
			}
			used += total;
		}
		if (v1 && !v2) { post = mid; mid = 0; }             // zeros behind the only value
		// as the kernel wants it: how far the lookup moves the raster position, where its values land relative to the position in front of it
		// (DX_NO_VALUE: none -- far outside any tile, the store goes to the dump slot), the values themselves
		const int total = pre + (v1 ? 1 : 0) + mid + (v2 ? 1 : 0) + post;
		const uint32_t o1 = v1 ? (uint32_t)pre : (uint32_t)DX_NO_VALUE, o2 = v2 ? (uint32_t)(pre + 1 + mid) : (uint32_t)DX_NO_VALUE;
		if (total > 0xfff || v1 < -128 || v1 > 127 || v2 < -128 || v2 > 127 || (v1 && pre >= (int)DX_NO_VALUE) || (v2 && pre + 1 + mid >= (int)DX_NO_VALUE)) return false;
		T->multi[win].x = (uint32_t)(used & 15) | ((uint32_t)total << 4) | (o1 << 16);
		T->multi[win].y = o2 | ((uint32_t)(uint8_t)(int8_t)v1 << 16) | ((uint32_t)(uint8_t)(int8_t)v2 << 24);
	}
	// k_dec_tiles, windows whose first code word does not fit (multi[win].x & 15 == 0): the code word through a trie of its own -- 11 bits (the
	// window), DX_L11_BITS more, the rest -- whose entries carry everything the kernel needs (length, kind, run or both magnitudes): one LDS read
	// for code words of 12 to 18 bits, two beyond, none when only the sign bit lies outside the window.
	{
		auto entry_of = [&](const RawCode &c, bool *ok) {
			// bits of the code word with its sign bit; one payload field per code set: the expanded magnitude, the zero run (in both), 0 for the band end marker
			const uint32_t a = c.kind == 1 ? T->mag_expand[0][c.payload] : (c.kind == 0 ? (uint32_t)c.payload : 0u), b = c.kind == 1 ? T->mag_expand[1][c.payload] : a;
			const uint32_t bits = (uint32_t)c.len + (c.kind == 1 ? 1u : 0u);
			if (a >= (1u << 12) || b >= (1u << 12) || bits > 31u) *ok = false;
			return bits | ((uint32_t)type_of(c) << 5) | (a << 8) | (b << 20);
		};
		bool ok = true;
		uint32_t n11 = 0;
		int l2_of[1 << DX_KM];
		for (int p = 0; p < (1 << DX_KM); p++) l2_of[p] = -1;
		struct L3b { uint32_t key; int maxlen; uint32_t base; };
		std::vector<L3b> l3b;
		for (int i = 0; i < ncodes; i++) {
			const RawCode &c = codes[i];
			const int total = c.len + (c.kind == 1 ? 1 : 0);
			if (total <= DX_KM) continue;                       // served by the multi-symbol entries
			if (c.len <= DX_KM) {                               // the code word ends inside the window, its sign bit does not
				const uint32_t base = c.bits << (DX_KM - c.len);
				for (uint32_t k = 0; k < (1u << (DX_KM - c.len)); k++) { if (T->multi[base + k].x & 15u) ok = false; T->multi[base + k].y = entry_of(c, &ok); }
				continue;
			}
			const uint32_t p11 = c.bits >> (c.len - DX_KM);
			if (l2_of[p11] < 0) {
				if (n11 + (1u << DX_L11_BITS) > DX_LONG11_MAX) return false;
				l2_of[p11] = (int)n11; n11 += 1u << DX_L11_BITS;
				if (T->multi[p11].x & 15u) ok = false;
				T->multi[p11].y = (uint32_t)DX_L11_BITS | ((uint32_t)DX_T_ESCAPE << 5) | ((uint32_t)l2_of[p11] << 8);
			}
			if (c.len > DX_KM + DX_L11_BITS) {
				const uint32_t key = c.bits >> (c.len - DX_KM - DX_L11_BITS);
				auto it = std::find_if(l3b.begin(), l3b.end(), [&](const L3b &x) { return x.key == key; });
				if (it == l3b.end()) l3b.push_back(L3b{ key, c.len, 0 }); else if (c.len > it->maxlen) it->maxlen = c.len;
			}
		}
		for (L3b &x : l3b) {
			const int nb = x.maxlen - DX_KM - DX_L11_BITS;
			if (n11 + (1u << nb) > DX_LONG11_MAX) return false;
			x.base = n11; n11 += 1u << nb;
			T->long11[(uint32_t)l2_of[x.key >> DX_L11_BITS] + (x.key & ((1u << DX_L11_BITS) - 1u))] = (uint32_t)nb | ((uint32_t)DX_T_ESCAPE << 5) | (x.base << 8);
		}
		for (int i = 0; i < ncodes; i++) {
			const RawCode &c = codes[i];
			if (c.len <= DX_KM) continue;
			const uint32_t e = entry_of(c, &ok);
			if (c.len <= DX_KM + DX_L11_BITS) {
				const int spare = DX_KM + DX_L11_BITS - c.len;
				const uint32_t base = (uint32_t)l2_of[c.bits >> (c.len - DX_KM)] + ((c.bits & ((1u << (c.len - DX_KM)) - 1u)) << spare);
				for (uint32_t k = 0; k < (1u << spare); k++) T->long11[base + k] = e;
			} else {
				const uint32_t key = c.bits >> (c.len - DX_KM - DX_L11_BITS);
				const L3b &x = *std::find_if(l3b.begin(), l3b.end(), [&](const L3b &y) { return y.key == key; });
				const int nb = x.maxlen - DX_KM - DX_L11_BITS, rest = c.len - DX_KM - DX_L11_BITS, spare = nb - rest;
				const uint32_t base = x.base + ((c.bits & ((1u << rest) - 1u)) << spare);
				for (uint32_t k = 0; k < (1u << spare); k++) T->long11[base + k] = e;
			}
		}
		if (!ok) return false;
		// (windows that are no prefix of any code word keep y = 0: type DX_T_INVALID)
		for (uint32_t win = 0; win < (1u << DX_KM); win++) if ((T->multi[win].x & 15u) == 0 && (T->multi[win].x & 0xffffu)) return false;   // (a window nothing fits into covers no coefficients)
	}
	return true;
}

struct EntHostJobs {
	std::vector<dev::EntBandJob> bands;
	std::vector<dev::EntSegJob> segjobs;
	std::vector<int> band_of_hole;      // template hole -> band job of frame 0 (-1 for lowpass holes)
	int nbands = 0;                     // coded bands per frame
	// segment ranges [first, count) inside one frame's table: the level-1 bands (final once the level-1 transform has run) and everything else
	std::vector<std::pair<int, int>> ranges_l1, ranges_rest;
};

// Where the holes of a template find their coefficients: the band of the pyramid behind every hole, in template order (the same for every frame of a batch).
struct EntHoleGeom { size_t offset; int width, height, pitch; int table /* 0: code set 17, 1: code set 18 */; int mask_base /* block lists: first mask, or -1 */; bool level1; };
inline std::vector<EntHoleGeom> ent_hole_geometry(const FramePlan &plan, const SampleTemplate &t)
{
	int mask_base[kMaxChannels][kNumBands];
	block_list_layout(plan, mask_base);
	std::vector<EntHoleGeom> g;
	for (const SampleTemplate::Hole &hole : t.holes) {
		const BandDesc &bd = plan.ch[hole.channel].band[hole.level][hole.band];
		g.push_back(EntHoleGeom{ bd.offset, bd.width, bd.height, bd.pitch,
		                         plan.interlaced && hole.level == 0 && hole.band == 2 ? 1 : 0,      // subband 8 of the channel (cfhd_bitstream.cpp walk_sample)
		                         hole.kind == 1 && hole.level == 0 ? mask_base[hole.channel][hole.band] : -1, hole.level == 0 });
	}
	return g;
}
// a two-frame group (cfhd_gop.cpp walk_group_sample): `level` is the wavelet index; the bands of the two frame wavelets are final once level 1 of both frames has run
inline std::vector<EntHoleGeom> ent_hole_geometry(const GopPlan &plan, const SampleTemplate &t)
{
	std::vector<EntHoleGeom> g;
	for (const SampleTemplate::Hole &hole : t.holes) {
		const GopWavelet &wv = plan.ch[hole.channel].w[hole.level];
		g.push_back(EntHoleGeom{ wv.offset[hole.band], wv.width, wv.height, wv.pitch, gop_band_is_difference_coded(plan, hole.level, hole.band) ? 1 : 0, -1, hole.level < 2 });      // (code set 18: subbands 12 and 15 of an interlaced group)
	}
	return g;
}

inline bool ent_build_band_jobs(const std::vector<EntHoleGeom> &geom, const SampleTemplate &t0, int nframes, int16_t *coeffs, size_t stride, EntHostJobs *out)
{
	if ((int)t0.holes.size() > dev::ENT_MAX_HOLES || (int)t0.patches.size() > kEntMaxPatches || geom.size() != t0.holes.size()) return false;
	out->bands.clear(); out->segjobs.clear(); out->ranges_l1.clear(); out->ranges_rest.clear();
	out->band_of_hole.assign(t0.holes.size(), -1);
	for (int f = 0; f < nframes; f++) {
		int16_t *base = coeffs + (size_t)f * stride;
		for (size_t h = 0; h < t0.holes.size(); h++) {
			const SampleTemplate::Hole &hole = t0.holes[h];
			if (hole.kind != 1) continue;
			const EntHoleGeom &bd = geom[h];
			dev::EntBandJob j;
			j.coeffs = base + bd.offset; j.n = bd.height * bd.pitch;
			j.nseg = (j.n + dev::ENT_SEG - 1) / dev::ENT_SEG; j.seg_base = (int)out->segjobs.size();
			j.frame = f; j.hole = (int)h;
			j.table = bd.table;
			if (f == 0) {
				out->band_of_hole[h] = (int)out->bands.size();
				std::vector<std::pair<int, int>> &r = bd.level1 ? out->ranges_l1 : out->ranges_rest;
				if (!r.empty() && r.back().first + r.back().second == j.seg_base) r.back().second += j.nseg; else r.push_back({ j.seg_base, j.nseg });
			}
			for (int s = 0; s < j.nseg; s++) out->segjobs.push_back(dev::EntSegJob{ j.coeffs, j.n, s * dev::ENT_SEG, (int)out->bands.size(), j.table, bd.pitch, bd.mask_base });
			out->bands.push_back(j);
		}
	}
	out->nbands = (int)out->bands.size() / nframes;
	return true;
}
inline bool ent_build_band_jobs(const FramePlan &plan, const SampleTemplate &t0, int nframes, int16_t *coeffs, size_t stride, EntHostJobs *out)
{
	return ent_build_band_jobs(ent_hole_geometry(plan, t0), t0, nframes, coeffs, stride, out);
}

// Serialises frame f's template into one kEntTmplStride block: bytes | holes-in-front-of-word | EntHole[] | EntPatch[].
inline bool ent_fill_frame_block(const std::vector<EntHoleGeom> &geom, const SampleTemplate &t, int f, const EntHostJobs &jobs, const int16_t *coeffs_f, uint8_t *block)
{
	if (t.bytes.size() > (size_t)kEntTmplBytes || t.bytes.size() / 4 > (size_t)kEntWordHolesBytes || t.holes.size() != jobs.band_of_hole.size() ||
	    t.patches.size() > (size_t)kEntMaxPatches || geom.size() != t.holes.size()) return false;
	memcpy(block, t.bytes.data(), t.bytes.size());
	uint8_t *wh = block + kEntTmplBytes;
	size_t hole = 0;
	for (size_t w = 0; w < t.bytes.size() / 4; w++) {
		while (hole < t.holes.size() && (size_t)t.holes[hole].tmpl_offset <= w * 4) hole++;
		wh[w] = (uint8_t)hole;
	}
	dev::EntHole *eh = (dev::EntHole *)(block + kEntTmplBytes + kEntWordHolesBytes);
	for (size_t i = 0; i < t.holes.size(); i++) {
		const SampleTemplate::Hole &src = t.holes[i];
		const EntHoleGeom &bd = geom[i];
		dev::EntHole &d = eh[i];
		d.tmpl_offset = src.tmpl_offset; d.kind = src.kind; d.fixed_bytes = src.fixed_bytes;
		// (a peak table belongs to the coded band in front of it)
		d.band_job = src.kind == 1 ? jobs.band_of_hole[i] + f * jobs.nbands : (src.kind == 2 && i > 0 && t.holes[i - 1].kind == 1 ? jobs.band_of_hole[i - 1] + f * jobs.nbands : -1);
		if (src.kind == 2 && d.band_job < 0) return false;
		d.lowpass = src.kind == 0 ? coeffs_f + bd.offset : nullptr;
		d.lp_width = bd.width; d.lp_height = bd.height; d.lp_pitch = bd.pitch;
	}
	dev::EntPatch *ep = (dev::EntPatch *)(block + kEntTmplBytes + kEntWordHolesBytes + kEntHolesBytes);
	for (size_t i = 0; i < t.patches.size(); i++) {
		const SampleTemplate::Patch &p = t.patches[i];
		ep[i].kind = p.kind; ep[i].at_tmpl = p.at_tmpl; ep[i].at_holes = p.at_holes; ep[i].start_tmpl = p.start_tmpl; ep[i].start_holes = p.start_holes;
		ep[i].end_tmpl = p.end_tmpl; ep[i].end_holes = p.end_holes; ep[i].tag = p.tag;
	}
	return true;
}
inline bool ent_fill_frame_block(const FramePlan &plan, const SampleTemplate &t, int f, const EntHostJobs &jobs, const int16_t *coeffs_f, uint8_t *block)
{
	return ent_fill_frame_block(ent_hole_geometry(plan, t), t, f, jobs, coeffs_f, block);
}

// Frame job whose pointers refer to `block_addr` (the device -- or, under emulation, host -- address of the serialised block).
inline dev::EntFrameJob ent_frame_job(const SampleTemplate &t, uint8_t *block_addr, uint8_t *out, uint32_t out_cap, uint32_t *size_out, uint32_t *peak_flag)
{
	dev::EntFrameJob fj;
	fj.out = out; fj.out_cap = out_cap;
	fj.tmpl = block_addr; fj.tmpl_bytes = (int)t.bytes.size();
	fj.word_holes = block_addr + kEntTmplBytes;
	fj.holes = (const dev::EntHole *)(block_addr + kEntTmplBytes + kEntWordHolesBytes); fj.nholes = (int)t.holes.size();
	fj.patches = (const dev::EntPatch *)(block_addr + kEntTmplBytes + kEntWordHolesBytes + kEntHolesBytes); fj.npatches = (int)t.patches.size();
	fj.sample_bytes = size_out; fj.peak_flag = peak_flag;
	return fj;
}


// Decode jobs of one parsed sample. `sample_addr` / `coeff_base` are the addresses the kernels will see (device, or host under
// emulation).  Returns false when the sample does not match the plan.
inline bool dec_build_jobs(const ParsedSample &ps, const FramePlan &plan, const uint8_t *sample_addr, int16_t *coeff_base, int out_pixel_kind,
                           std::vector<dev::DecBandJob> *bands, std::vector<dev::DecLowpassJob> *lowpass, bool skip_level1 = false)
{
	for (int c = 0; c < plan.num_channels; c++) {
		const ParsedBand &lp = ps.lowpass[c];
		const BandDesc &ll = plan.ch[c].band[2][0];
		if (!lp.present || lp.width != ll.width || lp.height != ll.height) return false;
		dev::DecLowpassJob lj = { sample_addr + lp.offset, coeff_base + ll.offset, ll.width, ll.height, ll.pitch, lowpass_bias(plan.precision, ll.width, out_pixel_kind, c) };
		lowpass->push_back(lj);
		for (int lv = skip_level1 ? 1 : 0; lv < kNumLevels; lv++)      // half resolution: the level-1 highpass bands are not needed
			for (int b = 1; b < 4; b++) {
				const ParsedBand &pb = ps.high[c][lv][b];
				const BandDesc &bd = plan.ch[c].band[lv][b];
				if (!pb.present || pb.width != bd.width || pb.height != bd.height || (pb.offset & 3) || (pb.codebook != 1 && pb.codebook != 0) || (bd.offset & 7) || (bd.pitch & 7)) return false;   // k_dec_bands_par clears bands with 16-byte stores
				dev::DecBandJob bj = { sample_addr + pb.offset, pb.bytes, coeff_base + bd.offset, bd.height * bd.pitch, pb.quant };
				bands->push_back(bj);
			}
	}
	return true;
}

// Geometry the GPU sample parser (k_dec_parse) checks a sample against, and the launch order of the band jobs: by band area,
// largest first, so that the long bands start early and the short ones fill the tail of the k_dec_bands_par launch.
inline void dec_build_plan(const FramePlan &plan, int out_pixel_kind, dev::DecPlan *dp)
{
	memset(dp, 0, sizeof(*dp));
	dp->width = plan.width; dp->display_height = plan.display_height; dp->encoded_format = plan.encoded_format; dp->num_channels = plan.num_channels;
	struct Key { int area, c, lv, b; };
	std::vector<Key> keys;
	for (int c = 0; c < plan.num_channels; c++) {
		const BandDesc &ll = plan.ch[c].band[2][0];
		dp->low[c] = dev::DecPlanBand{ ll.width, ll.height, ll.pitch, (int)ll.offset };
		dp->low_bias[c] = lowpass_bias(plan.precision, ll.width, out_pixel_kind, c);
		for (int lv = 0; lv < kNumLevels; lv++)
			for (int b = 1; b < 4; b++) {
				const BandDesc &bd = plan.ch[c].band[lv][b];
				dp->high[c][lv][b] = dev::DecPlanBand{ bd.width, bd.height, bd.pitch, (int)bd.offset };
				keys.push_back(Key{ bd.width * bd.height, c, lv, b });
			}
	}
	std::stable_sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) { return a.area > b.area; });
	for (size_t k = 0; k < keys.size(); k++) dp->slot[keys[k].c][keys[k].lv][keys[k].b] = (int)k;
	dp->bands_per_frame = (int)keys.size();
}

// ---- cfhd_dec_kernels.h: job table [band slot][frame], tiles, chunk numbering ----
// Frame f's rows of the [slot][frame] band job table (slots as dec_build_plan numbers them) and its lowpass jobs from a parsed sample.
// Bands the caller does not want (skip_level1: half resolution) keep bytes = 0 and produce no work.
inline bool dx_build_jobs(const ParsedSample &ps, const FramePlan &plan, const dev::DecPlan &dp, const uint8_t *sample_addr, int16_t *coeff_base, int out_pixel_kind,
                          int f, int nframes, dev::DecBandJob *table, dev::DecLowpassJob *lowpass /* [num_channels] */, bool skip_level1 = false,
                          dev::DecDiffJob *diff = nullptr /* [num_channels]: interlaced samples (code set 18, difference coding, peak tables) are accepted */)
{
	for (int c = 0; c < plan.num_channels; c++) {
		if (diff) diff[c] = dev::DecDiffJob{ nullptr, 0, 0, 0, nullptr, 0u, 0 };
		const ParsedBand &lp = ps.lowpass[c];
		const BandDesc &ll = plan.ch[c].band[2][0];
		if (!lp.present || lp.width != ll.width || lp.height != ll.height) return false;
		lowpass[c] = dev::DecLowpassJob{ sample_addr + lp.offset, coeff_base + ll.offset, ll.width, ll.height, ll.pitch, lowpass_bias(plan.precision, ll.width, out_pixel_kind, c) };
		for (int lv = 0; lv < kNumLevels; lv++)
			for (int b = 1; b < 4; b++) {
				const ParsedBand &pb = ps.high[c][lv][b];
				const BandDesc &bd = plan.ch[c].band[lv][b];
				dev::DecBandJob &bj = table[(size_t)dp.slot[c][lv][b] * nframes + f];
				bj = dev::DecBandJob{ sample_addr, 0u, coeff_base + bd.offset, bd.height * bd.pitch, 1, 0u, 0 };
				if (skip_level1 && lv == 0) continue;
				if (!pb.present || pb.width != bd.width || pb.height != bd.height || (pb.offset & 3) || pb.codebook < 0 || pb.codebook > 2 || (bd.offset & 7) || (bd.pitch & 7)) return false;
				if ((pb.codebook == 2 || pb.difference) && !diff) return false;
				bj.bits = sample_addr + pb.offset; bj.bytes = pb.bytes; bj.quant = pb.quant; bj.table = pb.codebook == 2;
				if (pb.difference) {
					if (pb.peak_level && (size_t)pb.peak_offset + 2 > ps.size) return false;
					diff[c] = dev::DecDiffJob{ coeff_base + bd.offset, bd.width, bd.height, bd.pitch, pb.peak_level ? sample_addr + pb.peak_offset : nullptr,
					                           pb.peak_level ? (uint32_t)(ps.size - pb.peak_offset) : 0u, pb.peak_level };
				}
			}
	}
	return true;
}

// Tiles of the job table: a band of n coefficients is cut into ceil(n / tile_max) tiles of equal length (a multiple of 512: the chunks of the block lists), for every frame.
// tile_max: dev::DX_TILE (what the kernel's LDS image holds); the emulated kernel tests pass less, so that their small frames meet bands of many tiles.
inline dev::DxTilePlan dx_tile_plan(const FramePlan &plan, const dev::DecPlan &dp, int nframes, bool skip_level1 = false, bool level1_block_lists = false, bool interlaced = false, uint32_t tile_max = dev::DX_TILE)
{
	int mask_base[kMaxChannels][kNumBands];
	dec_block_list_layout(plan, mask_base);
	dev::DxTilePlan tp;
	memset(&tp, 0, sizeof(tp));
	tp.nslots = dp.bands_per_frame; tp.nframes = nframes;
	int pos = 0;
	uint32_t cum = 0;
	for (int group = 0; group < 2; group++) {            // positions: the bands of levels 2 and 3 first, the level-1 bands behind them
		for (int c = 0; c < plan.num_channels; c++)
			for (int lv = 0; lv < kNumLevels; lv++) {
				if ((lv == 0) != (group == 1)) continue;
				for (int b = 1; b < 4; b++) {
					const BandDesc &bd = plan.ch[c].band[lv][b];
					const uint32_t n = (uint32_t)(bd.height * bd.pitch);
					const uint32_t per = (skip_level1 && lv == 0) ? 0u : (n + tile_max - 1) / tile_max;
					tp.tile_len[pos] = per ? ((n + per - 1) / per + 511u) / 512u * 512u : tile_max;
					tp.slot_of[pos] = (uint8_t)dp.slot[c][lv][b];
					// (the level-1 bands as block lists: k_dec_tiles -> k_inv_yuv422_strip_blocks / k_inv_frame_yuv422_strip_blocks; the difference-coded band of an interlaced
					// frame stays dense: k_dec_undiff walks its rows)
					tp.mask_base[pos] = (level1_block_lists && lv == 0 && !(interlaced && b == 2)) ? mask_base[c][b] : -1;
					tp.cum[pos] = cum; cum += per * (uint32_t)nframes;
					tp.per_band[pos] = per ? per : 1;      // positions without tiles must not be looked at by the kernel's division: one (unreachable) tile per band
					pos++;
				}
			}
		if (group == 0) tp.split = cum;
	}
	tp.total = cum; tp.first = 0;
	return tp;
}

// Chunk numbering on the host (what k_dec_plan does on the device): returns the number of chunks.
inline uint32_t dx_number_chunks(dev::DecBandJob *jobs, int njobs, std::vector<dev::DxChunkDesc> *chunk_desc)
{
	uint32_t at = 0;
	chunk_desc->clear();
	for (int j = 0; j < njobs; j++) {
		const uint32_t n = (jobs[j].bytes + dev::DX_CHUNK_BYTES - 1) / dev::DX_CHUNK_BYTES;
		jobs[j].chunk0 = at;
		for (uint32_t c = 0; c < n; c++) chunk_desc->push_back(dev::DxChunkDesc{ jobs[j].bits, jobs[j].bytes, c, jobs[j].table, 0u });
		at += n;
	}
	return at;
}

} // namespace cfhd
