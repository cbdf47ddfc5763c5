// cfhd_dec_kernels.h -- entropy decoder of the coded bands, organised for the GPU (replaces the workgroup-per-band kernel of round 1).
//
// What it computes is what Codec/decoder.c:19534 DecodeBandFSM16sNoGap computes (run-length / variable-length codes of code set 17 or 18
// -> dequantized coefficients in raster order, zero runs skipped, companding curve of codebooks.c:1345-1378 and the band's divisor of
// decoder.c:20597-20608 folded into the values), but the work is cut along two different axes:
//
//   k_dec_index   over the *input*: every wave takes one chunk of 2016 payload bytes (63 lanes x 256 bits, plus one lane that runs in
//                 through the 256 bits in front of the chunk).  A variable-length code has no markers to resynchronise on, but a decoder's
//                 only state is its bit position, so a lane that starts at an arbitrary bit falls in step with the true sequence of code
//                 words after a few symbols.  Every lane decodes its 256 bits once from a guessed start (the lane boundary), counting
//                 coefficients only (a table that covers several code words per lookup); then every lane compares its start with the
//                 position at which its left neighbour actually crossed the boundary and repeats only the first few code words, up to
//                 the point where the new chain meets the old one at a 64-bit mark.  Lane 1 of the band's first chunk starts exactly, so
//                 the result is exact, not probabilistic.  Output, per 64 payload bits: where the first code word starts and how many
//                 coefficients lie in front of it (relative to the chunk); per chunk: where it assumed to start, where it ended, how many
//                 coefficients it holds.
//   k_dec_chain   one wave per band: checks that every chunk started where its predecessor ended (a chunk that did not -- its run-in
//                 lane never fell in step, rare -- is indexed again from the exact position), and turns the chunks' coefficient counts into
//                 raster positions.
//   k_dec_tiles   over the *output*: every wave owns 2048 consecutive coefficients of a band (4 KB), finds the 64-bit pieces of
//                 payload whose code words land in it, decodes each piece on its own lane into an LDS image of the tile and streams the
//                 tile out with 16-byte stores.  The zero runs are never decoded into stores and never written twice: the band is written
//                 exactly once, in full lines, whatever the picture looks like (the round-1 kernel cleared each band and then scattered 2-byte
//                 stores into it: 1.24x the bytes, and its run time followed the longest band).
//
// All three are grid-stride over uniform work items; the tables (8 + 8 + 4.5 KB) live in LDS once per workgroup.
#pragma once
#include <stdint.h>
#include "cfhd_entropy_kernels.h"

// k_dec_tiles (round 6): a WORKGROUP owns a tile -- up to DX_TILE consecutive coefficients of a band, assembled in one LDS image -- and its waves share the
// pieces that reach into it, 64 consecutive pieces per wave and round.  Rounds 2-5 gave every wave its own tile of 4096 coefficients; measured on the bench's
// Qbist samples (tools/dx_walk_stats.py) the level-1 bands spend 0.38 payload bits per coefficient, so such a tile met 21 pieces: two thirds of the lanes of
// three quarters of the tiles had nothing to decode, while the pieces themselves are even (5-9 table lookups each).  What bounds the pieces in flight on a CU is
// the LDS that holds their output, so the image is now shared: 14848 coefficients (29 KB) meet ~90 level-1 pieces = one full wave and a half, three workgroups of
// eight waves fit a CU beside their tables (52.5 KB each), and the waves without pieces only help to stream the tile out.  Measured (profiles/r06_a_*, 512 1080p frames,
// one step at a time): 1.59 ms (round 5) -> 0.97 ms; 15360 / 256 threads 2.2 ms before the prefetch wait moved in front of the stores (below), 28672 / 512 threads (two
// workgroups per CU) 1.07, 27648 / 1024 1.42, 8192 / 256 (four per CU) 1.39: the kernel now waits on latency, not on issue slots, and wants waves.
#ifndef CFHD_DX_TILE
#define CFHD_DX_TILE 14848
#endif
#ifndef CFHD_DX_TILE_THREADS
#define CFHD_DX_TILE_THREADS 512
#endif

namespace cfhd {
namespace dev {

#ifndef CFHD_DX_MEMO
#define CFHD_DX_MEMO 6
#endif
// k_dec_index waits on chains of dependent LDS lookups: a fifth wave per SIMD (what its 31 KB of LDS allow: five workgroups per CU) is worth more
// than the 56 bytes of scratch the compiler needs to get from 113 to 96 registers (3.30 -> 2.93 ms per 512 frames; tools/dx_index_sweep.sh).
#ifndef CFHD_DX_WAVES
#define CFHD_DX_WAVES 5
#endif
#if CFHD_DX_WAVES > 0
#define CFHD_DX_INDEX_ATTR __attribute__((amdgpu_waves_per_eu(CFHD_DX_WAVES, CFHD_DX_WAVES)))
#else
#define CFHD_DX_INDEX_ATTR
#endif
enum {
	DX_K = 12,                        // bits of the first-level tables
	DX_LANE_BITS = 256, DX_SUB_BITS = 64, DX_SUBS = DX_LANE_BITS / DX_SUB_BITS,
	DX_OWN_LANES = 63,                // lanes 1..63 own the chunk, lane 0 runs in through the 256 bits in front of it
	DX_CHUNK_BITS = DX_OWN_LANES * DX_LANE_BITS, DX_CHUNK_BYTES = DX_CHUNK_BITS / 8, DX_CHUNK_WORDS = DX_CHUNK_BITS / 32,
	DX_CHUNK_SUBS = DX_OWN_LANES * DX_SUBS,        // 252 pieces of 64 bits per chunk
	DX_ENTRY_STRIDE = 64 * DX_SUBS,                // entries per chunk in memory (lane-major; lane 0's four are unused)
	DX_STAGE_WORDS = 64 * (DX_LANE_BITS / 32) + 2, // payload words of a chunk + run-in lane + two words of look-ahead
	DX_LONG_MAX = 1408,               // entries of the second / third level tables (code words of 13..26 bits)
	DX_L2_BITS = 7,
	DX_TILE = CFHD_DX_TILE,           // most coefficients of an output tile (a multiple of 512: the chunks of the block lists); DxTilePlan::tile_len says how long the tiles of a band are
	DX_THREADS = 256, DX_WAVES = DX_THREADS / 64,
	DX_TILE_THREADS = CFHD_DX_TILE_THREADS, DX_TILE_WAVES = DX_TILE_THREADS / 64,      // k_dec_tiles: the waves of a workgroup share the tables and the tile
	DX_KM = 11,                       // bits of the window of the multi-symbol table of k_dec_tiles
	DX_NO_VALUE = 0xC000,             // multi[]: offset of a value the lookup does not hold (beyond any tile whatever the position in front of it: a piece starts at most a few thousand coefficients in front of its tile)
	DX_L11_BITS = 7, DX_LONG11_MAX = 1664,   // k_dec_tiles' own tables for the code words that do not fit the 11-bit window: second level 7 bits, third level the rest (up to 26 in all)
	DX_RUNIN_SHORT = 96, DX_LEAD = 96,    // bits of the quick run-in in front of a chunk / of the lead-in in front of a lane
	DX_MEMO = CFHD_DX_MEMO,           // outcomes a lane of k_dec_index remembers (start -> end, count)
	DX_OFF_INVALID = 31,              // entry: no code word of the true sequence starts in this piece (behind the band end marker / the payload)
};
enum : uint32_t { DX_END = 0xFFFFFFFFu, DX_BAD = 0xFFFFFFFEu, DX_SPECIAL = 0xFFFFFFFEu };
enum { DX_FLAG_END = 1, DX_FLAG_BAD = 2, DX_FLAG_UNRESOLVED = 4, DX_ERR_BAD = 1 << 1, DX_ERR_OVERFLOW = 1 << 2, DX_ERR_NOEND = 1 << 3, DX_ERR_SPACE = 1 << 4 };
// type field of a long-table entry
enum { DX_T_INVALID = 0, DX_T_RUN = 1, DX_T_VALUE = 2, DX_T_END = 3, DX_T_ESCAPE = 4 };

struct DecIdxTables {
	uint16_t cnt12[1 << DX_K];        // as many whole code words (sign bits included) as fit into the next 12 bits: bits 0-3 bits used (0: none fits), bits 4-15 coefficients covered
	uint16_t sym12[1 << DX_K];        // first code word: bits 0-3 length without the sign bit (0: longer than 12 bits or invalid), bit 4 value (else zero run) -- with length 0: escape --,
	                                  // bits 5-15 run length / index of the magnitude / base of the second-level table
	uint16_t mag_expand[2][256];      // magnitude after undoing the companding curve, by index: [0] code set 17 (cubic), [1] code set 18 (linear)
	uint32_t long_tab[DX_LONG_MAX];   // bits 0-4 bits of the code word with its sign bit, 0 for the band end marker (escape: index bits of the next level), bits 5-7 type, bits 8-19 coefficients covered (escape: bits 8-31 base of the next level)
	uint32_t nlong;
	// k_dec_tiles: everything that fits completely (sign bits included) into the next 11 bits, up to two values with the zero runs around
	// them: x = bits 0-3 bits used (0: nothing fits -> one code word at a time), bits 4-15 coefficients the lookup covers, bits 16-31 position of
	// the first value relative to the position in front of the lookup (DX_NO_VALUE: none); y = bits 0-15 position of the second value (likewise),
	// bits 16-23 / 24-31 the values (signed).  Values this short are below the knee of the companding curve (magnitude = index), so the table
	// serves both code sets.
	// x & 15 == 0 (the first code word does not fit the window): y is an entry in the format of long11[] -- the code word itself when only its
	// sign bit lies outside the window, else an escape to long11[] indexed by the next DX_L11_BITS bits (and once more for code words beyond 18 bits).
	uint2 multi[1 << DX_KM];
	// bits 0-4 bits of the code word, sign bit included (escape: index bits of the next level), bits 5-7 type (DX_T_*), bits 8-19 payload under code set 17, bits 20-31 under
	// code set 18: the zero run (in both), the magnitude with the set's companding curve undone (cubic | linear), 0 for the band end marker -- or, escape: base of the next level (bits 8-19)
	uint32_t long11[DX_LONG11_MAX];
};

// One coded band of one frame = DecBandJob (cfhd_entropy_kernels.h); the job table is [band slot][frame], a band that is not wanted -- half
// resolution skips level 1 -- has bytes 0.  chunk0 = first chunk of the band in the chunk arrays (k_dec_plan / host).
typedef DecBandJob DxBandJob;
struct DxChunkDesc { const uint8_t *bits; uint32_t bytes, k; int quant_table; uint32_t pad; };   // chunk c of the launch: payload of its band, chunk number inside the band, the band's code set (DecBandJob::table) (k_dec_plan / host)
struct DxChunkRec { uint32_t start, end, count, flags; };   // start / end: bit offset of the first code word relative to the chunk's / the next chunk's first bit; flags: DX_FLAG_* | candidates << 8
// A chunk in front of which the code has no unique alignment (its run-in from every possible offset leaves several candidates for its first
// code word) is indexed once per candidate: the record holds candidate 0 (the entries are written for it), this the others.
enum { DX_MAX_ALT = 3 };
struct DxChunkAlt { uint32_t start[DX_MAX_ALT], end[DX_MAX_ALT], count[DX_MAX_ALT]; uint32_t slot; };      // slot: the candidates' entries sit in alt_entries[slot + q] (DX_BAD: not kept)
struct DxReindex { uint32_t chunk, k, start; int job; };      // a chunk whose entries have to be written again for the start that turned out to be the true one
struct DxBandSum { uint32_t total; int last_chunk; };        // coefficients the band's code words cover; chunk that holds the band end marker


// k_dec_index and its helpers look code words up in ONE 32-bit table (a single LDS read per step, nothing that depends on a second one
// for code words of up to 12 bits): cnt12 and sym12 of the same window side by side.  Bits 0-3 o1 = bits of the first code word, sign bit
// included (0: longer than 12 bits or invalid -- then bits 16-31 hold the sym12 entry for the second-level lookup), bits 4-7 bits of all
// whole code words in the window (cnt12), bits 8-19 coefficients the first code word covers, bits 20-31 coefficients all of them cover.
__device__ __forceinline__ uint32_t dx_tab_entry(const uint32_t c /* cnt12 */, const uint32_t e /* sym12 */)
{
	const uint32_t len = e & 15u;
	if (!len) return e << 16;
	const uint32_t isval = (e >> 4) & 1u;
	return (len + isval) | ((c & 15u) << 4) | ((isval ? 1u : (e >> 5)) << 8) | ((c >> 4) << 20);
}
// the code word behind a first-level entry without a length (e = its sym12 entry): its long_tab entry -- bits 0-4 bits of the code word with its sign bit (0: band end
// marker / no code word), bits 5-7 type (DX_T_*), bits 8-19 coefficients it covers (the zero run, 1 for a value) -- or 0 (DX_T_INVALID)
__device__ __forceinline__ uint32_t dx_long_entry(const uint32_t e, const uint32_t *s_long, uint32_t win)
{
	if (!(e & 16u)) return 0u;
	uint32_t x = s_long[(e >> 5) + ((win >> (32 - DX_K - DX_L2_BITS)) & ((1u << DX_L2_BITS) - 1u))];
	if (((x >> 5) & 7u) == DX_T_ESCAPE) {
		const int nb = (int)(x & 31u);
		x = s_long[(x >> 8) + ((win << (DX_K + DX_L2_BITS)) >> (32 - nb))];
	}
	return x;
}

// The payload words of a chunk in LDS: staging word i sits at i + i / 8, so that the 64 lanes of a wave, each walking its own 8 words, hit
// 64 different banks when they move in step (a plain layout puts lanes 4 apart on the same bank).
__device__ __forceinline__ uint32_t dx_phys(uint32_t i) { return i + (i >> 3); }
enum { DX_STAGE_PHYS = DX_STAGE_WORDS + DX_STAGE_WORDS / 8 + 1, DX_FETCH = (DX_STAGE_WORDS + 63) / 64 };

// Bit reader over the staged words: the next 32 bits sit on top of `acc`; one LDS read per 32 bits consumed.
struct DxBits {
	uint64_t acc; int have; uint32_t next;
	__device__ __forceinline__ void seek(const uint32_t *s_words, uint32_t pos)
	{
		const uint32_t wi = pos >> 5, sh = pos & 31u;
		acc = (((uint64_t)s_words[dx_phys(wi)] << 32) | s_words[dx_phys(wi + 1)]) << sh;
		have = 64 - (int)sh; next = wi + 2;
	}
	__device__ __forceinline__ uint32_t window() const { return (uint32_t)(acc >> 32); }
	__device__ __forceinline__ void skip(const uint32_t *s_words, int n)
	{
		acc <<= n; have -= n;
		if (have < 32) { acc |= (uint64_t)s_words[dx_phys(next < (uint32_t)DX_STAGE_WORDS ? next : (uint32_t)DX_STAGE_WORDS - 1u)] << (32 - have); have += 32; next++; }
	}
};

// The same reader with the word for the next refill already in a register: the LDS read that replaces it is issued together with the table
// lookup of the step (prefetch()) and is not waited for before the step after it, so a step waits for one LDS round trip, not two.
struct DxBitsAhead {
	uint64_t acc; int have; uint32_t next, nw;            // nw = staging word `next`
	__device__ __forceinline__ static uint32_t word(const uint32_t *s_words, uint32_t i) { return s_words[dx_phys(i < (uint32_t)DX_STAGE_WORDS ? i : (uint32_t)DX_STAGE_WORDS - 1u)]; }
	__device__ __forceinline__ void seek(const uint32_t *s_words, uint32_t pos)
	{
		const uint32_t wi = pos >> 5, sh = pos & 31u;
		acc = (((uint64_t)s_words[dx_phys(wi)] << 32) | s_words[dx_phys(wi + 1)]) << sh;
		have = 64 - (int)sh; next = wi + 2; nw = word(s_words, next);
	}
	__device__ __forceinline__ uint32_t window() const { return (uint32_t)(acc >> 32); }
	__device__ __forceinline__ uint32_t prefetch(const uint32_t *s_words) const { return word(s_words, next + 1u); }
	__device__ __forceinline__ void skip(int n, uint32_t ahead /* prefetch() of this step */)
	{
		acc <<= n; have -= n;
		if (have < 32) { acc |= (uint64_t)nw << (32 - have); have += 32; next++; nw = ahead; }
	}
};

struct DxLane {                       // state of one lane of k_dec_index
	uint32_t start, end;              // bit positions in staging coordinates (lane t owns [256 t, 256 t + 256)); end may be DX_END / DX_BAD
	uint32_t cnt;                     // coefficients covered by the code words that start in the lane's range
	uint32_t rec_offs;                // per 64-bit piece, one byte each: first code word's offset into the piece (DX_OFF_INVALID: none)
	uint32_t rec_cnt[DX_SUBS];        // coefficients of the lane in front of that code word
};
enum : uint32_t { DX_OFFS_NONE = DX_OFF_INVALID * 0x01010101u };
__device__ __forceinline__ uint32_t dx_off_get(uint32_t offs, int k) { return (offs >> (8 * k)) & 0xffu; }
__device__ __forceinline__ uint32_t dx_off_set(uint32_t offs, int k, uint32_t v) { return (offs & ~(0xffu << (8 * k))) | (v << (8 * k)); }
// pieces k .. 3 hold no code word of this walk
__device__ __forceinline__ uint32_t dx_off_clear_from(uint32_t offs, int k) { const uint32_t m = k >= 4 ? 0u : 0xffffffffu << (8 * k); return (offs & ~m) | ((uint32_t)DX_OFFS_NONE & m); }

// Walks the code words of one lane from bit `pos` to the end of the lane's range.  merge: stop as soon as the walk reaches a 64-bit mark at
// the offset recorded by the previous walk -- from there on the two chains are the same, only the counts in front shift.
// One flat loop: the lanes of a wave cross their marks at different times, and a loop per piece would make every lane wait for the slowest
// one four times over.
// A walk may start in front of the lane (a lead-in through the neighbour's last bits, to fall in step before the lane begins): counting starts
// with the first code word inside the lane, whose position is returned in L.start.
// The steps of a walk up to a mark: one table lookup per step -- the first code word, or as many whole code words as fit the 12-bit window without
// passing `lim` -- until the position reaches lim.  This is the loop the kernel lives in; everything that happens once per 64-bit piece (records,
// the merge test) is outside it.  COUNT: add up the coefficients the code words cover.  Returns false when the walk met the band end marker or a
// broken code (endv says which).
template <bool COUNT>
__device__ __forceinline__ bool dx_steps(DxBitsAhead &B, uint32_t &pos, uint32_t &cnt, const uint32_t lim, uint32_t &endv, const uint32_t *s_words, const uint32_t *s_tab, const uint32_t *s_long)
{
	bool ok = true;
	bool go = pos < lim;
	while (go) {
		const uint32_t win = B.window();
		const uint32_t t = s_tab[win >> (32 - DX_K)];
		const uint32_t ahead = B.prefetch(s_words);
		const uint32_t used = (t >> 4) & 15u;
		uint32_t adv = t & 15u, add = (t >> 8) & 0xfffu;      // the first code word ...
		const bool all = adv != 0u && used != 0u && pos + used <= lim;
		adv = all ? used : adv; add = all ? t >> 20 : add;      // ... or several whole ones, none of them beyond the mark
		if (adv == 0u) {
			const uint32_t x = dx_long_entry(t >> 16, s_long, win), ty = (x >> 5) & 7u;
			adv = x & 31u; add = (x >> 8) & 0xfffu;             // (made for this place: cfhd_entropy_jobs.h build_dec_index_tables)
			if (ty - (uint32_t)DX_T_RUN >= 2u) { endv = ty == (uint32_t)DX_T_END ? (uint32_t)DX_END : (uint32_t)DX_BAD; ok = false; }
		}
		pos += adv;
		if (COUNT) cnt += add;
		B.skip((int)adv, ahead);
		go = ok && pos < lim;
	}
	return ok;
}

// Walks the code words of one lane from bit `pos` to the end of the lane's range.  merge: stop as soon as the walk enters a 64-bit piece at the
// offset recorded by the previous walk -- from there on the two chains are the same, only the counts in front shift.
// A walk may start in front of the lane (a lead-in through the neighbour's last bits, to fall in step before the lane begins): counting starts
// with the first code word inside the lane, whose position is returned in L.start.
// Round 2 ran this as ONE flat loop with the per-piece bookkeeping inside (so that no lane waits at a piece boundary for the slowest one): about
// 100 instructions per step, most of them bookkeeping of the execution mask.  Piece by piece the steps are a loop of about 30 (dx_steps); the
// lanes of a wave wait for each other five times per walk instead of once, which costs far less than it saves.
__device__ __forceinline__ void dx_walk(DxLane &L, uint32_t pos, const bool merge, const uint32_t lane_base, const uint32_t limit, const uint32_t *s_words, const uint32_t *s_tab,
                                        const uint32_t *s_long)
{
	uint32_t cnt = 0, start = pos, endv = pos;
	uint32_t offs = merge ? L.rec_offs : (uint32_t)DX_OFFS_NONE;
	uint32_t rc[DX_SUBS] = { L.rec_cnt[0], L.rec_cnt[1], L.rec_cnt[2], L.rec_cnt[3] };
	const uint32_t lane_end = lane_base + DX_LANE_BITS;
	const uint32_t stop = lane_end < limit ? lane_end : limit;
	int piece = -1, merged_at = 0;
	bool done = false, clear = false, merged = false;
	DxBitsAhead B;
	B.seek(s_words, pos);
	// the approach through the bits in front of the lane (a lead-in): nothing is counted
	{
		const uint32_t lim = lane_base < stop ? lane_base : stop;
		if (!dx_steps<false>(B, pos, cnt, lim, endv, s_words, s_tab, s_long)) { clear = true; done = true; }
	}
#pragma unroll
	for (int k = 0; k < DX_SUBS; k++) {
		const uint32_t mark_lo = lane_base + (uint32_t)k * DX_SUB_BITS, mark = mark_lo + DX_SUB_BITS;
		if (!done) {
			if (pos >= stop) {                                // the end of the lane, or of the payload in front of it
				endv = pos; clear = pos < lane_end; done = true;
			} else if (pos < mark) {
				// a code word starts in piece k
				const uint32_t off = pos - mark_lo;
				if (piece < 0) { cnt = 0u; start = pos; }     // the first code word of the lane
				if (merge && k > 0 && dx_off_get(offs, k) == off) {
					merged = true; merged_at = k; done = true;    // same chain from here on
				} else {
					offs = dx_off_set(offs, k, off); rc[k] = cnt; piece = k;
					const uint32_t lim = mark < stop ? mark : stop;
					if (!dx_steps<true>(B, pos, cnt, lim, endv, s_words, s_tab, s_long)) { clear = true; done = true; }
				}
			} else offs = dx_off_set(offs, k, (uint32_t)DX_OFF_INVALID);      // a late start: no code word of this walk begins in piece k
		}
	}
	if (!done) { endv = pos; clear = pos < lane_end; }     // through all four pieces
	if (merged) {
		// the counts recorded behind the mark where the chains met move by the difference in front of it; L.end stays
		const int k = merged_at;
		const uint32_t old = k == 1 ? L.rec_cnt[1] : (k == 2 ? L.rec_cnt[2] : L.rec_cnt[3]);
		const uint32_t delta = cnt - old;
		L.rec_cnt[0] = rc[0];
		L.rec_cnt[1] = k <= 1 ? (dx_off_get(offs, 1) != (uint32_t)DX_OFF_INVALID ? L.rec_cnt[1] + delta : L.rec_cnt[1]) : rc[1];
		L.rec_cnt[2] = k <= 2 ? (dx_off_get(offs, 2) != (uint32_t)DX_OFF_INVALID ? L.rec_cnt[2] + delta : L.rec_cnt[2]) : rc[2];
		L.rec_cnt[3] = dx_off_get(offs, 3) != (uint32_t)DX_OFF_INVALID ? L.rec_cnt[3] + delta : L.rec_cnt[3];
		L.cnt += delta;
		L.rec_offs = offs;
		L.start = start;
		return;
	}
	if (clear) offs = dx_off_clear_from(offs, piece + 1);
	L.rec_cnt[0] = rc[0]; L.rec_cnt[1] = rc[1]; L.rec_cnt[2] = rc[2]; L.rec_cnt[3] = rc[3];
	L.start = start; L.end = endv;
	L.rec_offs = offs;
	L.cnt = cnt;
}

// Staging of chunk k of a band: fetch the words into registers (the loads can be in flight while the previous chunk is walked), then put them
// into the wave's LDS area in big-endian bit order.  Staging word i = payload word k * DX_CHUNK_WORDS - 8 + i; words outside the payload read 0.
struct DxFetch { uint32_t w[DX_FETCH]; };
__device__ __forceinline__ void dx_fetch_chunk(const uint8_t *bits, const uint32_t bytes, const uint32_t k, DxFetch &F)
{
	const int lane = wave_lane();
	const uint32_t nwords = bytes >> 2;
	const uint32_t *words = (const uint32_t *)bits;
	const int64_t first = (int64_t)k * DX_CHUNK_WORDS - (DX_LANE_BITS / 32);
#pragma unroll
	for (int r = 0; r < DX_FETCH; r++) {
		const int64_t g = first + lane + 64 * r;
		F.w[r] = (g >= 0 && g < (int64_t)nwords && lane + 64 * r < DX_STAGE_WORDS) ? words[g] : 0u;
	}
}
__device__ __forceinline__ void dx_store_stage(const DxFetch &F, uint32_t *s_words)
{
	const int lane = wave_lane();
#pragma unroll
	for (int r = 0; r < DX_FETCH; r++) if (lane + 64 * r < DX_STAGE_WORDS) s_words[dx_phys((uint32_t)(lane + 64 * r))] = bswap32(F.w[r]);
	CFHD_WAVE_SYNC();
}

// Where the code words that pass through the 256 bits in front of a chunk can enter the chunk: lanes 0..26 walk that range from every
// offset a first code word can have; the distinct outcomes (bit offsets into the chunk) are the candidates for the chunk's first code word.
// Ordinary data leaves one; returns their number (at most DX_MAX_ALT + 2: more than DX_MAX_ALT + 1 means "too many"), 0 when every walk
// ended on the band end marker or a broken code (the chunk holds padding).
__device__ __forceinline__ int dx_runin_candidates(const uint32_t bytes, const uint32_t k, const uint32_t runin_bits, const uint32_t *s_words, const uint32_t *s_tab,
                                                   const uint32_t *s_long, uint32_t (&cand)[DX_MAX_ALT + 2])
{
	const int lane = wave_lane();
	const uint32_t nwords = bytes >> 2;
	const int64_t first = (int64_t)k * DX_CHUNK_WORDS - (DX_LANE_BITS / 32);
	const int64_t left = (int64_t)nwords - first;
	const uint32_t limit = left <= 0 ? 0u : (left * 32 > (int64_t)(64 * DX_LANE_BITS + 64) ? (uint32_t)(64 * DX_LANE_BITS + 64) : (uint32_t)(left * 32));
	uint32_t pos = (uint32_t)DX_LANE_BITS - runin_bits + (uint32_t)lane, end = DX_BAD;
	if (lane < 27) {
		DxBitsAhead B;
		B.seek(s_words, pos);
		bool alive = true;                                    // (one loop, one way out: see dx_walk)
		do {
			if (pos >= (uint32_t)DX_LANE_BITS) { end = pos; alive = false; }
			else if (pos >= limit) alive = false;
			else {
				const uint32_t win = B.window();
				const uint32_t t = s_tab[win >> (32 - DX_K)];
				const uint32_t ahead = B.prefetch(s_words);
				const uint32_t used = (t >> 4) & 15u;
				uint32_t adv = t & 15u;
				adv = (adv != 0u && used != 0u && pos + used <= (uint32_t)DX_LANE_BITS) ? used : adv;
				if (adv == 0u) {
					const uint32_t x = dx_long_entry(t >> 16, s_long, win);
					adv = x & 31u;
					if (((x >> 5) & 7u) - (uint32_t)DX_T_RUN >= 2u) alive = false;
				}
				if (alive) { pos += adv; B.skip((int)adv, ahead); }
			}
		} while (alive);
	}
	unsigned long long mask = __ballot(lane < 27 && end < DX_SPECIAL);
	int n = 0;
#pragma unroll 1
	while (mask && n < DX_MAX_ALT + 2) {
		const int l = __builtin_ctzll(mask);
		const uint32_t v = wave_get(end, l);
		cand[n++] = v - DX_LANE_BITS;
		mask &= ~__ballot(end == v);
	}
	return n;
}

// Index of one staged chunk by one wave from the bit offset (relative to the chunk's first bit) at which its first code word starts; the
// per-piece entries are written when `entries` is given.  Returns the chunk's record (start, end, count, flags) in every lane.
__device__ __forceinline__ DxChunkRec dx_index_staged(const uint32_t bytes, const uint32_t gchunk, const uint32_t k, const uint32_t exact_start, const uint32_t *s_words, const uint32_t *s_tab,
                                                      const uint32_t *s_long, uint32_t *entries, uint32_t *stats = nullptr)
{
	const int lane = wave_lane();
	const uint32_t nwords = bytes >> 2;
	const int64_t first = (int64_t)k * DX_CHUNK_WORDS - (DX_LANE_BITS / 32);
	const int64_t left = (int64_t)nwords - first;           // payload words from the start of the staging area on
	const uint32_t limit = left <= 0 ? 0u : (left * 32 > (int64_t)(64 * DX_LANE_BITS + 64) ? (uint32_t)(64 * DX_LANE_BITS + 64) : (uint32_t)(left * 32));   // staging bit position behind the payload
	const uint32_t lane_base = (uint32_t)lane * DX_LANE_BITS;
	// lanes whose range lies behind the payload take no part: the last lane with payload bits carries the end of the chain
	const bool live = lane_base < limit && lane >= 1;
	const int last_live = limit == 0u ? 0 : (int)((limit - 1u) / DX_LANE_BITS) < 63 ? (int)((limit - 1u) / DX_LANE_BITS) : 63;
	DxLane L;
	L.start = DX_BAD; L.end = lane_base; L.cnt = 0; L.rec_offs = DX_OFFS_NONE;
#pragma unroll
	for (int j = 0; j < DX_SUBS; j++) L.rec_cnt[j] = 0;
	if (lane == 0) { L.start = 0u; L.end = exact_start >= DX_SPECIAL ? exact_start : DX_LANE_BITS + exact_start; }   // the chunk's first code word
	// Every lane first walks from a guessed start (its own boundary), then takes over where its left neighbour ended, until nothing moves any
	// more.  Ordinary data falls in step within a few code words, so one round settles almost every lane (the repeated walk stops at the
	// first 64-bit mark where it meets the old one).  A stretch of identical code words (a smooth gradient: the same value in every
	// position) has no unique alignment: a lane cannot know its phase before its neighbour does, and the true starts move through the chunk
	// one lane per round.  A lane meets at most as many different starts as the pattern has phases, so every lane remembers the outcome of
	// the starts it has walked (DX_MEMO of them): once the phases are known a round costs a lookup, not a walk.  The per-piece records are
	// brought up to date in one last pass.
	uint32_t memo_s[DX_MEMO], memo_e[DX_MEMO], memo_c[DX_MEMO];
	int memo_at = 0;
#pragma unroll
	for (int i = 0; i < DX_MEMO; i++) { memo_s[i] = DX_BAD; memo_e[i] = 0; memo_c[i] = 0; }
	uint32_t rec_start = DX_BAD, rec_end = 0, rec_total = 0;     // the walk the per-piece records belong to: its start, end and coefficient count
	bool finishing = false;
#pragma unroll 1
	for (int round = 0; round < 70; round++) {
		uint32_t want; bool need;
		if (finishing) {
			// records for the final start (a lane that ended on a remembered outcome still holds the records of another walk)
			const bool special = L.start >= DX_SPECIAL || L.start < lane_base || L.start >= lane_base + DX_LANE_BITS;
			if (live && special) L.rec_offs = DX_OFFS_NONE;
			want = L.start; need = live && !special && rec_start != L.start;
		} else if (round == 0) {
			// lane 1 follows lane 0 (the chunk's first code word is known); every other lane leads in through its neighbour's last bits
			const uint32_t first = __shfl_up(L.end, 1u);
			want = lane == 1 ? first : lane_base - DX_LEAD; need = live;
		} else {
			want = __shfl_up(L.end, 1u); need = live && want != L.start;
		}
		const unsigned long long moved = __ballot(need);
		if (!finishing && round > 0) {
			if (stats && lane == 0 && moved) atomicAdd(&stats[4 + (round - 1 < 11 ? round - 1 : 11)], (uint32_t)__builtin_popcountll(moved));
			if (!moved) {
				if (stats && lane == 0) { atomicAdd(&stats[0], (uint32_t)round - 1u); atomicAdd(&stats[1], 1u); atomicMax(&stats[2], (uint32_t)round - 1u); }
				finishing = true; round--;                    // nothing moves any more: one more pass for the records
				continue;
			}
		}
		if (finishing && !moved) break;
		if (need) {
			L.start = want;
			const bool lead = !finishing && round == 0 && lane >= 2;
			if (!lead && (want >= DX_SPECIAL || want < lane_base || want >= lane_base + DX_LANE_BITS)) {
				// behind the band end marker (or a broken code), or the neighbour's chain does not reach this lane: nothing of the true sequence starts here
				L.end = want >= DX_SPECIAL ? want : (uint32_t)DX_BAD; L.cnt = 0;
			} else {
				int hit = -1;
#pragma unroll
				for (int i = 0; i < DX_MEMO; i++) if (memo_s[i] == want) hit = i;
				if (hit >= 0 && !finishing && !lead) {
#pragma unroll
					for (int i = 0; i < DX_MEMO; i++) if (i == hit) { L.end = memo_e[i]; L.cnt = memo_c[i]; }
				} else {
					L.end = rec_end; L.cnt = rec_total;           // a walk that meets the recorded one continues from ITS outcome (a remembered outcome may have replaced it meanwhile)
					dx_walk(L, want, !finishing && rec_start != DX_BAD, lane_base, limit, s_words, s_tab, s_long);
					const uint32_t walked = lead ? L.start : want;        // a lead-in reports the first code word inside the lane
					L.start = walked;
					rec_start = walked; rec_end = L.end; rec_total = L.cnt;
#pragma unroll
					for (int i = 0; i < DX_MEMO; i++) if (i == memo_at) { memo_s[i] = walked; memo_e[i] = L.end; memo_c[i] = L.cnt; }
					memo_at = memo_at + 1 < DX_MEMO ? memo_at + 1 : 0;
				}
			}
		}
		if (finishing) break;
	}
	// entries: offset of the first code word | coefficients of the chunk in front of it
	const uint32_t own = live ? L.cnt : 0u;
	const uint32_t incl = wave_incl_scan(own);
	const uint32_t before = incl - own;
	const size_t slot = (size_t)gchunk * DX_ENTRY_STRIDE + (size_t)lane * DX_SUBS;
	if (lane >= 1 && entries) {
		uint4 e;
		uint32_t v[DX_SUBS];
#pragma unroll
		for (int j = 0; j < DX_SUBS; j++) v[j] = (!live || dx_off_get(L.rec_offs, j) == (uint32_t)DX_OFF_INVALID) ? (uint32_t)DX_OFF_INVALID : (dx_off_get(L.rec_offs, j) | ((before + L.rec_cnt[j]) << 5));
		e.x = v[0]; e.y = v[1]; e.z = v[2]; e.w = v[3];
		*(uint4 *)(entries + slot) = e;
	}
	const uint32_t total = wave_get(incl, 63), el = wave_get(L.end, last_live);
	DxChunkRec r;
	r.start = exact_start;
	// a chain that stops in front of the chunk's end without the band end marker ran off the payload
	r.end = el >= DX_SPECIAL ? el : ((last_live < 63 || el < 64u * DX_LANE_BITS) ? (uint32_t)DX_BAD : el - 64u * DX_LANE_BITS);
	r.count = total;
	r.flags = (r.end == DX_END ? DX_FLAG_END : 0u) | (r.end == DX_BAD ? DX_FLAG_BAD : 0u);
	CFHD_WAVE_SYNC();
	return r;
}

// Stage + index from an exact start, not pipelined: the repair and re-index paths (rare; out of line so that it does not weigh on the callers' registers).
__device__ __attribute__((noinline)) DxChunkRec dx_index_chunk(const uint8_t *bits, const uint32_t bytes, const uint32_t gchunk, const uint32_t k, const uint32_t exact_start, uint32_t *s_words,
                                                               const uint32_t *s_tab, const uint32_t *s_long, uint32_t *entries, DxChunkRec *recs, uint32_t *stats)
{
	DxFetch F;
	dx_fetch_chunk(bits, bytes, k, F);
	dx_store_stage(F, s_words);
	const DxChunkRec r = dx_index_staged(bytes, gchunk, k, exact_start, s_words, s_tab, s_long, entries, stats);
	if (wave_lane() == 0 && recs) recs[gchunk] = r;
	return r;                                             // the same in every lane
}

__device__ __forceinline__ void dx_load_tables(const DecIdxTables *T, uint16_t *s_cnt, uint16_t *s_sym, uint32_t *s_long, bool want_cnt)
{
	const uint32_t *c = (const uint32_t *)T->cnt12, *s = (const uint32_t *)T->sym12;
	for (int i = threadIdx.x; i < (1 << DX_K) / 2; i += blockDim.x) { if (want_cnt) ((uint32_t *)s_cnt)[i] = c[i]; ((uint32_t *)s_sym)[i] = s[i]; }
	for (int i = threadIdx.x; i < DX_LONG_MAX; i += blockDim.x) s_long[i] = T->long_tab[i];
}

// the combined first-level table of the index walk (dx_tab_entry) and the long-code table, by the whole workgroup
__device__ __forceinline__ void dx_load_tab(const DecIdxTables *T, uint32_t *s_tab, uint32_t *s_long)
{
	for (int i = threadIdx.x; i < (1 << DX_K); i += blockDim.x) s_tab[i] = dx_tab_entry(T->cnt12[i], T->sym12[i]);
	for (int i = threadIdx.x; i < DX_LONG_MAX; i += blockDim.x) s_long[i] = T->long_tab[i];
}

__device__ __forceinline__ uint32_t dx_nchunks(uint32_t bytes) { return (bytes + DX_CHUNK_BYTES - 1) / DX_CHUNK_BYTES; }

// jobs[j].chunk0 = first chunk of job j; counters[0] = number of chunks.  One workgroup (a scan over a few thousand numbers).
__global__ void __launch_bounds__(1024) k_dec_plan(DxBandJob *jobs, int njobs, uint32_t max_chunks, uint32_t *counters, int *errors)
{
	__shared__ uint32_t s_part[1024];
	const int t = threadIdx.x, per = (njobs + 1023) / 1024;
	uint32_t sum = 0;
	for (int i = 0; i < per; i++) { const int j = t * per + i; if (j < njobs) sum += dx_nchunks(jobs[j].bytes); }
	s_part[t] = sum;
	__syncthreads();
	for (int d = 1; d < 1024; d <<= 1) {
		const uint32_t x = t >= d ? s_part[t - d] : 0u;
		__syncthreads();
		s_part[t] += x;
		__syncthreads();
	}
	uint32_t at = s_part[t] - sum;
	const uint32_t total = s_part[1023];
	if (t == 0) { counters[0] = total <= max_chunks ? total : 0u; if (total > max_chunks) atomic_or_u32((uint32_t *)errors, (uint32_t)DX_ERR_SPACE); }
	for (int i = 0; i < per; i++) {
		const int j = t * per + i;
		if (j >= njobs) break;
		jobs[j].chunk0 = at;
		at += dx_nchunks(jobs[j].bytes);
	}
}
// chunk_desc[c] = payload and number of chunk c: one wave per band writes its band's descriptors.
__global__ void __launch_bounds__(DX_THREADS) k_dec_plan_fill(const DxBandJob *jobs, int njobs, DxChunkDesc *chunk_desc, const uint32_t *counters)
{
	const int j = (int)blockIdx.x * DX_WAVES + wave_uniform((int)(threadIdx.x >> 6));
	if (j >= njobs || counters[0] == 0u) return;
	const DxBandJob job = jobs[j];
	const uint32_t n = dx_nchunks(job.bytes);
	for (uint32_t c = (uint32_t)wave_lane(); c < n; c += 64) chunk_desc[job.chunk0 + c] = DxChunkDesc{ job.bits, job.bytes, c, job.table, 0u };
}

__global__ void __launch_bounds__(DX_THREADS) CFHD_DX_INDEX_ATTR k_dec_index(const DxChunkDesc *chunk_desc, const uint32_t *counters, const DecIdxTables *T,
                                                          uint32_t *entries, DxChunkRec *recs, DxChunkAlt *alts, int speculate, uint32_t *stats,
                                                          uint32_t *alt_entries, uint32_t alt_slots, uint32_t *alt_counter, uint32_t *next_chunk)
{
	__shared__ uint32_t s_tab[1 << DX_K];
	__shared__ uint32_t s_long[DX_LONG_MAX];
	__shared__ uint32_t s_words_all[DX_WAVES][DX_STAGE_PHYS];
	dx_load_tab(T, s_tab, s_long);
	__syncthreads();
	const uint32_t total = counters[0];
	const int wave = wave_uniform((int)(threadIdx.x >> 6));
	uint32_t *s_words = s_words_all[wave];
	const uint32_t gwave = (uint32_t)blockIdx.x * DX_WAVES + (uint32_t)wave, nwaves = (uint32_t)gridDim.x * DX_WAVES;
	// software pipeline over this wave's chunks: while chunk c is walked, the payload words of its next chunk are already on their way
	// into registers.  The first chunk of a wave is its own number, the others are handed out by a counter (*next_chunk, zero at launch):
	// chunks differ in cost, and with a fixed stride a workgroup that gets its CU late -- another queue's kernel holds it: the blit kernel of
	// a sample download, say -- made the whole launch wait for a full second pass (seen under rocprofv3 as 2x the launch time).
	uint32_t c = gwave;
	if (c >= total) return;
	DxChunkDesc d = chunk_desc[c];
	DxFetch F;
	dx_fetch_chunk(d.bits, d.bytes, d.k, F);
#pragma unroll 1
	for (; c < total;) {
		dx_store_stage(F, s_words);
		uint32_t c1 = 0;
		if (wave_lane() == 0) c1 = nwaves + atomicAdd(next_chunk, 1u);
		c1 = wave_get(c1, 0);
		DxChunkDesc d1 = d;
#if !defined(CFHD_DX_NOPREFETCH)
		if (c1 < total) { d1 = chunk_desc[c1]; dx_fetch_chunk(d1.bits, d1.bytes, d1.k, F); }
#endif
		// A band's first chunk starts on its first bit; every other chunk finds the candidates for its first code word by running in through
		// the 256 bits in front of it from every possible offset.  speculate == 0 (tests): assume offset 0 instead, which is wrong for most
		// chunks -- k_dec_chain has to repair them.
		uint32_t cand[DX_MAX_ALT + 2] = { 0u, 0u, 0u, 0u, 0u };
		int n = 1;
		if (d.k != 0 && speculate) {
			// the last 96 bits settle ordinary data; only when they leave more than one candidate the whole 256 are walked
			n = dx_runin_candidates(d.bytes, d.k, DX_RUNIN_SHORT, s_words, s_tab, s_long, cand);
			if (n != 1) n = dx_runin_candidates(d.bytes, d.k, DX_LANE_BITS, s_words, s_tab, s_long, cand);
		}
		if (n == 0) {                                        // behind the band end marker: padding
			if (wave_lane() == 0) recs[c] = DxChunkRec{ (uint32_t)DX_END, (uint32_t)DX_END, 0u, (uint32_t)DX_FLAG_END | (1u << 8) };
		} else {
			const bool unresolved = n > DX_MAX_ALT + 1;
			if (unresolved) n = 1;
			DxChunkRec r = dx_index_staged(d.bytes, c, d.k, cand[0], s_words, s_tab, s_long, entries, stats);
			r.flags |= ((uint32_t)n << 8) | (unresolved ? (uint32_t)DX_FLAG_UNRESOLVED : 0u);
			if (wave_lane() == 0) recs[c] = r;
			if (n > 1) {
				DxChunkAlt a;
#pragma unroll
				for (int i = 0; i < DX_MAX_ALT; i++) { a.start[i] = DX_BAD; a.end[i] = DX_BAD; a.count[i] = 0; }
				// the entries of the other candidates are kept too (a slot each in alt_entries, while there is room): when k_dec_chain finds one
				// of them to be the true start, k_dec_reindex copies 1 KB instead of indexing the chunk again
				uint32_t slot0 = DX_BAD;
				if (alt_entries) {
					if (wave_lane() == 0) slot0 = atomicAdd(alt_counter, (uint32_t)(n - 1));
					slot0 = wave_get(slot0, 0);
					if (slot0 + (uint32_t)(n - 1) > alt_slots) slot0 = DX_BAD;
				}
				a.slot = slot0;
#pragma unroll 1
				for (int i = 1; i < n; i++) {
					const DxChunkRec ri = dx_index_staged(d.bytes, slot0 == (uint32_t)DX_BAD ? c : slot0 + (uint32_t)(i - 1), d.k, cand[i], s_words, s_tab, s_long,
					                                      slot0 == (uint32_t)DX_BAD ? nullptr : alt_entries, nullptr);
#pragma unroll
					for (int q = 0; q < DX_MAX_ALT; q++) if (q == i - 1) { a.start[q] = ri.start; a.end[q] = ri.end; a.count[q] = ri.count; }
				}
				if (wave_lane() == 0) alts[c] = a;
				if (stats && wave_lane() == 0) atomicAdd(&stats[3], 1u << 16);       // chunks with more than one candidate (upper half of the repair counter)
			}
		}
#if defined(CFHD_DX_NOPREFETCH)
		if (c1 < total) { d1 = chunk_desc[c1]; dx_fetch_chunk(d1.bits, d1.bytes, d1.k, F); }
#endif
		d = d1; c = c1;
	}
}

// The tables for a wave that has to repair a chunk: loaded by that wave alone, no workgroup barrier (another wave of the workgroup doing the
// same writes the same words).
__device__ __forceinline__ void dx_load_tables_wave(const DecIdxTables *T, uint32_t *s_tab, uint32_t *s_long)
{
	const int lane = wave_lane();
	for (int i = lane; i < (1 << DX_K); i += 64) s_tab[i] = dx_tab_entry(T->cnt12[i], T->sym12[i]);
	for (int i = lane; i < DX_LONG_MAX; i += 64) s_long[i] = T->long_tab[i];
}

// One band's chain of chunks, by one wave: every chunk starts where its predecessor ended; chunk_base[c] = raster position of chunk c's first
// code word.  A chunk indexed for several candidate starts contributes the outcome of the one that is true (and goes on the re-index list when
// that is not the one its entries were written for).  A chunk none of whose candidates is true -- more candidates than k_dec_index keeps -- is
// indexed again on the spot when REPAIR is set; otherwise the band is only reported (false) and left to k_dec_repair.
// How a chunk is indexed again from an exact start: RX.
struct DxNoReindex { __device__ __forceinline__ DxChunkRec operator()(const DxBandJob &, uint32_t, uint32_t, uint32_t) { return DxChunkRec{ 0u, 0u, 0u, 0u }; } };
struct DxReindexCount {
	const DecIdxTables *T; uint32_t *s_tab, *s_long, *s_words, *entries; DxChunkRec *recs; bool tables;
	__device__ __forceinline__ DxChunkRec operator()(const DxBandJob &job, uint32_t gchunk, uint32_t k, uint32_t start)
	{
		if (!tables) { dx_load_tables_wave(T, s_tab, s_long); tables = true; CFHD_WAVE_SYNC(); }
		return dx_index_chunk(job.bits, job.bytes, gchunk, k, start, s_words, s_tab, s_long, entries, recs, nullptr);
	}
};
template <bool REPAIR, class RX>
__device__ __forceinline__ bool dx_chain_band(const DxBandJob &job, const int j, RX &rx, DxChunkRec *recs, const DxChunkAlt *alts, uint32_t *chunk_base, DxBandSum *sums, int *errors,
                                              DxReindex *reindex_list, uint32_t *counters, uint32_t *stats)
{
	const int lane = wave_lane();
	const uint32_t nch = dx_nchunks(job.bytes);
	if (nch == 0) { if (lane == 0) { sums[j].total = 0; sums[j].last_chunk = -1; } return true; }
	uint32_t prev_end = 0, base = 0;
	int last = -1, err = 0;
#pragma unroll 1
	for (uint32_t c0 = 0; c0 < nch && last < 0 && !err; c0 += 64) {
		const uint32_t c = c0 + (uint32_t)lane;
		const bool have = c < nch;
		const int nblock = nch - c0 < 64u ? (int)(nch - c0) : 64;
		DxChunkRec r = { 0u, 0u, 0u, 0u };
		if (have) r = recs[(size_t)job.chunk0 + c];
		uint32_t my_end = r.end, my_cnt = r.count;
		int nvalid = nblock;
		bool stop = false;
		// the common case: every chunk's first candidate starts where the predecessor's ended, up to the chunk that holds the end of the band
		uint32_t pe = __shfl_up(r.end, 1u);
		if (lane == 0) pe = prev_end;
		unsigned long long bad = __ballot(have && r.start != pe);
		const unsigned long long stop0 = __ballot(have && (r.end >= DX_SPECIAL));
		if (stop0) bad &= (2ull << __builtin_ctzll(stop0)) - 1ull;
		if (!bad) {
			if (stop0) { nvalid = __builtin_ctzll(stop0) + 1; stop = true; }
		} else {
			// chunk by chunk: which candidate is the true one depends on where the predecessor ended
			uint32_t cur = prev_end;
#pragma unroll 1
			for (int i = 0; i < nblock; i++) {
				if (cur >= DX_SPECIAL) { nvalid = i; stop = true; break; }
				const uint32_t st = wave_get(r.start, i), fl = wave_get(r.flags, i);
				uint32_t e = wave_get(r.end, i), cn = wave_get(r.count, i);
				if (st != cur) {
					bool found = false;
					if (((fl >> 8) & 0xffu) > 1u) {
						const DxChunkAlt a = alts[(size_t)job.chunk0 + c0 + (uint32_t)i];
#pragma unroll
						for (int q = 0; q < DX_MAX_ALT; q++) if (!found && a.start[q] == cur) { e = a.end[q]; cn = a.count[q]; found = true; }
					}
					if (found) {                                // the entries were written for another candidate: once more, later, in parallel with the others
						if (lane == 0) reindex_list[atomicAdd(&counters[2], 1u)] = DxReindex{ job.chunk0 + c0 + (uint32_t)i, c0 + (uint32_t)i, cur, j };
					} else {
						if (!REPAIR) return false;
						if (stats && lane == 0) atomicAdd(&stats[3], 1u);
						const DxChunkRec rr = rx(job, job.chunk0 + c0 + (uint32_t)i, c0 + (uint32_t)i, cur);
						e = rr.end; cn = rr.count;
					}
				}
				if (lane == i) { my_end = e; my_cnt = cn; }
				cur = e;
				if (e >= DX_SPECIAL) { nvalid = i + 1; stop = true; break; }
			}
		}
		const uint32_t cntv = lane < nvalid ? my_cnt : 0u;
		const uint32_t incl = wave_incl_scan(cntv);
		if (lane < nvalid) chunk_base[(size_t)job.chunk0 + c] = base + incl - cntv;
		base += wave_get(incl, 63);
		if (stop) {
			last = (int)c0 + nvalid - 1;
			if (nvalid == 0 || wave_get(my_end, nvalid > 0 ? nvalid - 1 : 0) == (uint32_t)DX_BAD) err |= DX_ERR_BAD;
		}
		prev_end = wave_get(my_end, 63);
	}
	if (last < 0 && !err) err |= DX_ERR_NOEND;           // ran off the payload without meeting the end marker
	if (base > (uint32_t)job.n) err |= DX_ERR_OVERFLOW;   // more coefficients than the band holds (the tile kernel never writes outside its tile)
	if (lane == 0) {
		sums[j].total = base; sums[j].last_chunk = last < 0 ? (int)nch - 1 : last;
		if (err) atomic_or_u32((uint32_t *)errors, (uint32_t)err);
	}
	return true;
}

// One wave per band, without the means to index a chunk again: bands that need it go on the repair list (counters[1] = their number).
__global__ void __launch_bounds__(DX_THREADS) k_dec_chain(const DxBandJob *jobs, int njobs, const DxChunkRec *recs, const DxChunkAlt *alts, uint32_t *chunk_base, DxBandSum *sums, int *errors,
                                                          uint32_t *repair_list, DxReindex *reindex_list, uint32_t *counters)
{
	const int j = (int)blockIdx.x * DX_WAVES + wave_uniform((int)(threadIdx.x >> 6));
	if (j >= njobs) return;
	const DxBandJob job = jobs[j];
	DxNoReindex none;
	if (!dx_chain_band<false>(job, j, none, (DxChunkRec *)recs, alts, chunk_base, sums, errors, reindex_list, counters, nullptr) && wave_lane() == 0)
		repair_list[atomicAdd(&counters[1], 1u)] = (uint32_t)j;
}

// The bands of the repair list, a wave each (a small grid strides over the list; usually it is empty).
__global__ void __launch_bounds__(DX_THREADS) k_dec_repair(const DxBandJob *jobs, const DecIdxTables *T, uint32_t *entries, DxChunkRec *recs, const DxChunkAlt *alts, uint32_t *chunk_base,
                                                           DxBandSum *sums, int *errors, const uint32_t *repair_list, DxReindex *reindex_list, uint32_t *counters, uint32_t *stats)
{
	__shared__ uint32_t s_tab[1 << DX_K];
	__shared__ uint32_t s_long[DX_LONG_MAX];
	__shared__ uint32_t s_words_all[DX_WAVES][DX_STAGE_PHYS];
	const uint32_t n = counters[1];
	const int wave = wave_uniform((int)(threadIdx.x >> 6));
	for (uint32_t i = (uint32_t)blockIdx.x * DX_WAVES + (uint32_t)wave; i < n; i += (uint32_t)gridDim.x * DX_WAVES) {
		const int j = (int)repair_list[i];
		const DxBandJob job = jobs[j];
		DxReindexCount rx = { T, s_tab, s_long, s_words_all[wave], entries, recs, false };
		(void)dx_chain_band<true>(job, j, rx, recs, alts, chunk_base, sums, errors, reindex_list, counters, stats);
	}
}

// The chunks whose entries were written for a candidate start that turned out not to be the true one: once more from the true start, a wave
// each (counters[2] = their number; all of them at once, nothing depends on anything any more).
__global__ void __launch_bounds__(DX_THREADS) k_dec_reindex(const DxBandJob *jobs, const DecIdxTables *T, uint32_t *entries, const DxReindex *reindex_list, const uint32_t *counters, uint32_t *stats,
                                                            const DxChunkAlt *alts, const uint32_t *alt_entries)
{
	__shared__ uint32_t s_tab[1 << DX_K];
	__shared__ uint32_t s_long[DX_LONG_MAX];
	__shared__ uint32_t s_words_all[DX_WAVES][DX_STAGE_PHYS];
	const uint32_t n = counters[2];
	if ((uint32_t)blockIdx.x * DX_WAVES >= n) return;
	const int wave = wave_uniform((int)(threadIdx.x >> 6));
	bool tables = false;                                  // (workgroup-uniform use below: loaded by the first wave that needs them, for itself)
	for (uint32_t i = (uint32_t)blockIdx.x * DX_WAVES + (uint32_t)wave; i < n; i += (uint32_t)gridDim.x * DX_WAVES) {
		const DxReindex x = reindex_list[i];
		// the usual case: k_dec_index kept the entries of this candidate -- copy them into the chunk's place
		if (alts && alt_entries) {
			const DxChunkAlt a = alts[x.chunk];
			int q = -1;
#pragma unroll
			for (int k = 0; k < DX_MAX_ALT; k++) if (q < 0 && a.start[k] == x.start) q = k;
			if (q >= 0 && a.slot != (uint32_t)DX_BAD) {
				const uint4 *src = (const uint4 *)(alt_entries + ((size_t)a.slot + (size_t)q) * DX_ENTRY_STRIDE);
				uint4 *dst = (uint4 *)(entries + (size_t)x.chunk * DX_ENTRY_STRIDE);
				const int lane = wave_lane();
				if (lane >= 1) dst[lane] = src[lane];           // lane t's four pieces (lane 0 is the run-in: it has no entries)
				if (stats && lane == 0) atomicAdd(&stats[3], 1u << 8);
				continue;
			}
		}
		if (!tables) { dx_load_tables_wave(T, s_tab, s_long); tables = true; CFHD_WAVE_SYNC(); }
		const DxBandJob job = jobs[x.job];
		dx_index_chunk(job.bits, job.bytes, x.chunk, x.k, x.start, s_words_all[wave], s_tab, s_long, entries, nullptr, nullptr);
		if (stats && wave_lane() == 0) atomicAdd(&stats[3], 1u << 8);
	}
}

// Tiles of the [slot][frame] job table: tile_cum[s] = tiles in front of slot s (all frames), tiles_per_band[s] = tiles of one band of slot s.
// Tiles are numbered position by position: position p holds the tiles of band slot slot_of[p] for every frame (cum[p] tiles lie in front of it, per_band[p] per
// band).  The positions of the level-2 and level-3 bands come first (tiles 0 .. split - 1), those of the level-1 bands behind them, so that a launch over
// [first, total) can decode one group while the inverse transforms of the other run (GpuEntropyDecoder::launch_dx).
struct DxTilePlan { int nslots, nframes; uint32_t cum[40]; uint32_t per_band[40]; uint32_t tile_len[40]; /* coefficients per tile of the position (a multiple of 512, at most DX_TILE) */ uint32_t total, first, split; uint8_t slot_of[40];
                    int mask_base[40]; /* per position: first chunk mask of the band in a frame's mask array when its tiles leave as block lists (cfhd_core.h dec_block_list_layout), else -1 */ };

enum : uint32_t { DX_TILE_EMPTY = 0xFFFFFFFFu };


// What k_dec_tiles needs to know about a tile, ready made (k_dec_tile_index, one thread per tile): the decode kernel fetches one record per tile two turns ahead and has no
// chain of dependent loads (plan -> job -> band sums -> tile start) and no division left in its turn.
struct DxTileDesc {
	const uint8_t *bits; int16_t *dst; unsigned long long *masks;     // payload of the band; first coefficient of the tile; the tile's first chunk mask when it leaves as block lists, else null
	uint32_t bytes, chunk0, first_sub, end_sub;                       // payload bytes, first chunk of the band in the chunk arrays, pieces [first_sub, end_sub) may reach into the tile (first_sub DX_TILE_EMPTY: none)
	uint32_t T0, ncoef, quant, table;                                 // first coefficient, coefficients of the tile (0: nothing to write: a band that is not wanted), divisor, code set (DecBandJob::table)
	uint32_t pad[2];
};
static_assert(sizeof(DxTileDesc) == 64, "four 16-byte loads");

// The 64-bit piece of payload that holds the code word at (or the last one in front of) coefficient T0 of a band -- where k_dec_tiles starts to decode a tile that begins
// there.  Two binary searches over what k_dec_index / k_dec_chain left: chunks, then pieces.
__device__ __forceinline__ uint32_t dx_piece_at(const DxBandJob &job, const DxBandSum &sum, const uint32_t *entries, const uint32_t *chunk_base, const uint32_t T0)
{
	if (!(job.bytes != 0u && sum.last_chunk >= 0 && T0 < sum.total && T0 < (uint32_t)job.n)) return DX_TILE_EMPTY;
	const uint32_t *cb = chunk_base + job.chunk0;
	uint32_t lo = 0, hi = (uint32_t)sum.last_chunk;       // largest chunk k with cb[k] <= T0 (cb[0] = 0)
	while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (cb[mid] <= T0) lo = mid; else hi = mid - 1; }
	const uint32_t kc = lo, base = cb[kc];
	const uint32_t *e = entries + ((size_t)job.chunk0 + kc) * DX_ENTRY_STRIDE + DX_SUBS;      // the chunk's 252 pieces, in order
	uint32_t a = 0, b = DX_CHUNK_SUBS - 1;                // largest piece whose first code word lies at or in front of T0 (piece 0 does)
	while (a < b) {
		const uint32_t mid = (a + b + 1) >> 1, v = e[mid];
		if ((v & 31u) != (uint32_t)DX_OFF_INVALID && base + (v >> 5) <= T0) a = mid; else b = mid - 1;
	}
	return kc * DX_CHUNK_SUBS + a;
}

// A thread per output tile writes its record.  The pieces that may reach into the tile run from the one that holds its first coefficient to the one that holds the first
// coefficient of the band's next tile (the band's last tile: to the chunk with the band end marker): a workgroup of DX_THREADS threads serves DX_THREADS - 1 consecutive tiles,
// every thread searches the first piece of one tile (the last thread: of the tile behind the workgroup's) and hands it to its left neighbour through LDS -- one chain of
// dependent loads per thread, not two (a synchronous CFHD_DecodeSample waits for this kernel's latency, not for its throughput).
enum { DX_TILE_INDEX_TILES = DX_THREADS - 1 };
__global__ void __launch_bounds__(DX_THREADS) k_dec_tile_index(const DxBandJob *jobs, DxTilePlan plan, const uint32_t *entries, const uint32_t *chunk_base, const DxBandSum *sums,
                                                               DxTileDesc *tiles, unsigned long long *masks, uint32_t masks_per_frame)
{
	__shared__ uint32_t s_first[DX_THREADS];
	__shared__ int s_job[DX_THREADS];
	const uint32_t t = (uint32_t)blockIdx.x * DX_TILE_INDEX_TILES + (uint32_t)threadIdx.x;
	const bool have = t < plan.total;
	int slot = 0, j = -1;
	uint32_t per = 1, len = 0, f = 0, ti = 0, T0 = 0, first = DX_TILE_EMPTY;
	bool any = false;
	DxBandJob job; DxBandSum sum;
	memset(&job, 0, sizeof(job)); sum.total = 0; sum.last_chunk = -1;
	if (have) {
		while (slot + 1 < plan.nslots && t >= plan.cum[slot + 1]) slot++;
		const uint32_t r = t - plan.cum[slot];
		per = plan.per_band[slot]; len = plan.tile_len[slot];
		f = r / per; ti = r - f * per;
		j = (int)plan.slot_of[slot] * plan.nframes + (int)f;
		job = jobs[j]; sum = sums[j];
		T0 = ti * len;
		any = job.bytes != 0u && T0 < (uint32_t)job.n;
		if (any) first = dx_piece_at(job, sum, entries, chunk_base, T0);
	}
	s_first[threadIdx.x] = first; s_job[threadIdx.x] = have ? j : -1;
	__syncthreads();
	if (!have || threadIdx.x == DX_TILE_INDEX_TILES) return;      // (the last thread only searched for its left neighbour)
	const uint32_t T1 = any ? (T0 + len < (uint32_t)job.n ? T0 + len : (uint32_t)job.n) : T0;
	DxTileDesc d;
	d.bits = job.bits; d.bytes = job.bytes; d.chunk0 = job.chunk0; d.quant = (uint32_t)job.quant; d.table = (uint32_t)job.table;
	d.dst = job.dst + T0; d.T0 = T0; d.ncoef = T1 - T0;
	d.masks = (any && masks && plan.mask_base[slot] >= 0) ? masks + (size_t)f * masks_per_frame + (size_t)plan.mask_base[slot] + (size_t)(T0 / 512u) : nullptr;
	d.first_sub = first;
	d.end_sub = 0u;
	if (first != DX_TILE_EMPTY) {
		const uint32_t last = ((uint32_t)sum.last_chunk + 1u) * DX_CHUNK_SUBS;
		// the right neighbour's tile is the band's next tile when it belongs to the same band of the same frame
		const uint32_t next = (ti + 1u < per && s_job[threadIdx.x + 1] == j) ? s_first[threadIdx.x + 1] : (uint32_t)DX_TILE_EMPTY;
		d.end_sub = (next != DX_TILE_EMPTY && next + 1u < last) ? next + 1u : last;
	}
	d.pad[0] = 0u; d.pad[1] = 0u;
	tiles[t] = d;
}

// k_dec_tiles fetches a tile's record and the first 64 pieces of payload every wave decodes of it ahead of time, so that the loads are in flight while the tile
// in front is decoded.
// 0 in every lane, but not to the compiler: an address with it added is per-lane, so the load becomes a vector load (counted by vmcnt) and
// not a scalar one -- scalar loads share their counter with LDS, and the first LDS read of the decode loop would wait for the prefetch.
__device__ __forceinline__ uint32_t dx_lane_zero() { return __builtin_amdgcn_mbcnt_lo(0u, 0u); }
template <typename T> __device__ __forceinline__ void dx_vload(T &dst, const T *src)
{
	static_assert(sizeof(T) % 4 == 0, "dwords");
	const uint32_t *p = (const uint32_t *)src + dx_lane_zero();
	uint32_t *d = (uint32_t *)&dst;
#pragma unroll
	for (int i = 0; i < (int)(sizeof(T) / 4); i++) d[i] = p[i];
}
template <typename T> __device__ __forceinline__ T dx_uniform(const T &v)    // back to scalar registers (the value is the same in every lane)
{
	T r;
	const uint32_t *s = (const uint32_t *)&v; uint32_t *d = (uint32_t *)&r;
#pragma unroll
	for (int i = 0; i < (int)(sizeof(T) / 4); i++) d[i] = (uint32_t)wave_uniform((int)s[i]);
	return r;
}
struct DxPieces { uint32_t ent, cb, d[3]; };
__device__ __forceinline__ void dx_tile_pieces(const DxTileDesc &D, uint32_t q, const uint32_t *entries, const uint32_t *chunk_base, DxPieces &P)
{
	P.ent = DX_OFF_INVALID; P.cb = 0;
#pragma unroll
	for (int i = 0; i < 3; i++) P.d[i] = 0u;
	if (D.first_sub != DX_TILE_EMPTY && q < D.end_sub) {
		// entry, chunk position and the next 96 bits of the payload from the piece on (a walk starts inside the piece's 64 bits and looks at 32 bits at a time): independent loads
		const uint32_t kq = q / DX_CHUNK_SUBS, within = q - kq * DX_CHUNK_SUBS;
		P.ent = entries[((size_t)D.chunk0 + kq) * DX_ENTRY_STRIDE + DX_SUBS + within];
		P.cb = chunk_base[(size_t)D.chunk0 + kq];
		const uint32_t byte0 = q * (DX_SUB_BITS / 8);
		const uint32_t *src = (const uint32_t *)(D.bits + byte0);
#pragma unroll
		for (int i = 0; i < 3; i++) P.d[i] = byte0 + 4u * (uint32_t)i + 4u <= D.bytes ? CFHD_LDG32(src + i) : 0u;      // (global_load: a flat load would tick the LDS counter too, and the decode loop's first table lookup would wait for the prefetch)
	}
}

// The LDS image of a workgroup's tile: DX_TILE coefficients and, behind them, one dump slot per thread -- a store that has nothing to write (no value
// in this step, or a position outside the tile: a piece reaches in from the tile in front or out into the next one) goes there instead of being
// masked out, which keeps the decode loop free of execution-mask bookkeeping.  The image holds the values as the code words give them (the short
// ones are below the knee of the companding curve, the long ones carry their expanded magnitude in the table); the band's divisor is applied on the
// way out, two coefficients per multiply.
enum { DX_TILE_WORDS = DX_TILE / 2 + DX_TILE_THREADS / 2 };
enum { DX_WAIT_VMCNT0 = 0x0f70 };      // s_waitcnt vmcnt(0) on gfx9: vmcnt = bits 3:0 and 15:14 (0), expcnt = bits 6:4 (7: no wait), lgkmcnt = bits 11:8 (15: no wait)
static_assert(DX_TILE % 512 == 0, "a tile is a whole number of chunks of 64 blocks");
static_assert(DX_TILE + DX_TILE_THREADS + 4096 <= DX_NO_VALUE, "a position the table marks as absent must lie behind the dump slots wherever the piece starts (at most 6 x 320 coefficients in front of its tile)");

// One round of k_dec_tiles: this lane's piece q (entry, chunk position and payload words in P) decoded into the LDS image of the tile [T0, T1).  Returns false (to the
// whole wave) when a piece of the round starts behind the tile: pieces are in raster order, all later ones do too.
__device__ __forceinline__ bool dx_tile_round(const DxPieces &P, const uint32_t q, const uint32_t end_sub, const uint32_t T0, const uint32_t T1, const int len_tile, const uint32_t mag_shift,
                                              const uint32_t dump, const uint2 *s_multi, const uint32_t *s_long, int16_t *tile16)
{
	const uint32_t off = P.ent & 31u;
	const uint32_t idx0 = P.cb + (P.ent >> 5);
	const bool valid = q < end_sub && off != (uint32_t)DX_OFF_INVALID;
	const bool inside = valid && idx0 < T1;
	if (inside) {
		// The walk starts at bit `off` (< 31) of the piece and goes on while it is inside the piece's 64 bits; a code word has at most 27
		// bits, so every 32-bit window the walk looks at lies in the piece's first 96 bits: three words, no refill state -- the window at bit
		// position pos is cut out of the word pair it starts in.
		const uint32_t w0 = bswap32(P.d[0]), w1 = bswap32(P.d[1]), w2 = bswap32(P.d[2]);
		uint32_t pos = off;
		uint32_t rel = idx0 - T0;                          // position inside the tile; "negative" (the piece starts in front of the tile) wraps to a huge number
		// one loop with one way out; both kinds of step (a group out of the multi-symbol table | one long code word) feed the same two
		// unconditional stores
		bool alive = true;
		do {
			const bool second = pos >= 32u;
			const uint32_t win = (uint32_t)(((((uint64_t)(second ? w1 : w0)) << 32) | (second ? w2 : w1)) << (pos & 31u) >> 32);
			// up to two values and the zero runs around them per lookup.  A group may reach over the end of the piece: the lane of the
			// next piece then writes the same values to the same places again.
			const uint2 e = s_multi[win >> (32 - DX_KM)];
			uint32_t adv = e.x & 15u, total = (e.x >> 4) & 0xfffu, o1 = e.x >> 16, o2 = e.y & 0xffffu;
			int v1 = (int)(int8_t)(e.y >> 16), v2 = (int)(int8_t)(e.y >> 24);
			if (adv == 0u) {
				// a code word that does not fit the window (a large value, a long run, the band end marker): alone, through its own trie.  Its entry is made for
				// this place: bits incl. the sign bit, and one payload field per code set that is the run, the expanded magnitude, or 0 (end marker, no code)
				uint32_t le = e.y;
				if (((le >> 5) & 7u) == (uint32_t)DX_T_ESCAPE) {
					le = s_long[((le >> 8) & 0xfffu) + ((win << DX_KM) >> (32 - DX_L11_BITS))];
					if (((le >> 5) & 7u) == (uint32_t)DX_T_ESCAPE) le = s_long[((le >> 8) & 0xfffu) + ((win << (DX_KM + DX_L11_BITS)) >> (32 - (le & 31u)))];
				}
				const uint32_t ty = (le >> 5) & 7u, pay = (le >> mag_shift) & 0xfffu;
				const bool isval = ty == (uint32_t)DX_T_VALUE;
				adv = le & 31u;
				total = isval ? 1u : pay;
				o1 = isval ? 0u : (uint32_t)DX_NO_VALUE; o2 = (uint32_t)DX_NO_VALUE;
				v1 = (int)(win << ((adv - 1u) & 31u)) < 0 ? -(int)pay : (int)pay;        // (a run's "value" goes to the dump slot)
				alive = ty - (uint32_t)DX_T_RUN < 2u;                // else: the band end marker (or a broken code, reported by k_dec_chain)
			}
			// a value whose place lies outside the tile (or that is not there at all) goes to the thread's dump slot: min() does both
			const uint32_t p1 = rel + o1, p2 = rel + o2;
			tile16[p1 < dump ? p1 : dump] = (int16_t)v1;
			tile16[p2 < dump ? p2 : dump] = (int16_t)v2;
			rel += total;
			pos += adv;
			alive = alive && pos < (uint32_t)DX_SUB_BITS && (int)rel < len_tile;
		} while (alive);
	}
	return __ballot(valid && !inside) == 0ull;
}

template <int NT /* threads: DX_TILE_THREADS (the emulated kernel tests also run 64, so that small tiles take several rounds per wave) */>
__global__ void __launch_bounds__(NT) k_dec_tiles(const DxTileDesc *tiles, uint32_t first, uint32_t total, const DecIdxTables *T, const uint32_t *entries, const uint32_t *chunk_base)
{
	__shared__ uint2 s_multi[1 << DX_KM];
	__shared__ uint32_t s_long[DX_LONG11_MAX];
	__shared__ uint32_t s_tile[DX_TILE / 2 + NT / 2];
	for (int i = threadIdx.x; i < DX_LONG11_MAX; i += blockDim.x) s_long[i] = T->long11[i];
	for (int i = threadIdx.x; i < (1 << DX_KM); i += blockDim.x) s_multi[i] = T->multi[i];
	for (int i = threadIdx.x; i < DX_TILE / 2 + NT / 2; i += blockDim.x) s_tile[i] = 0u;
	const int lane = wave_lane(), wave = wave_uniform((int)(threadIdx.x >> 6));
	__syncthreads();
	const uint32_t stride = (uint32_t)gridDim.x;
	uint32_t t = first + (uint32_t)blockIdx.x;
	if (t >= total) return;                                // (the whole workgroup)
	// software pipeline: tile t is decoded while the record of tile t + 2 stride and this wave's first pieces of tile t + stride are on their way
	DxTileDesc D, D1;
	dx_vload(D, tiles + t);
	D1 = D;
	if (t + stride < total) dx_vload(D1, tiles + t + stride); else D1.first_sub = DX_TILE_EMPTY;
	const uint32_t mine = (uint32_t)wave * 64u + (uint32_t)lane;        // this thread's piece in the first round of a tile
	DxPieces P;
	dx_tile_pieces(D, D.first_sub + mine, entries, chunk_base, P);
	int16_t *tile16 = (int16_t *)s_tile;
	const uint32_t dump = (uint32_t)DX_TILE + (uint32_t)threadIdx.x;     // this thread's dump slot (16-bit index)
	__builtin_amdgcn_s_waitcnt(DX_WAIT_VMCNT0);               // (the first tile's record and pieces: from here on every turn waits for its loads in one place, see below)
#pragma unroll 1
	for (; t < total; t += stride) {
		DxTileDesc D2 = D1;
		DxPieces P1;
		if (t + 2 * stride < total) dx_vload(D2, tiles + t + 2 * stride); else D2.first_sub = DX_TILE_EMPTY;
		dx_tile_pieces(D1, D1.first_sub + mine, entries, chunk_base, P1);
		const DxTileDesc d = dx_uniform(D);
		const uint32_t T0 = d.T0, T1 = d.T0 + d.ncoef;
		if (d.first_sub != DX_TILE_EMPTY) {
			// 64 consecutive pieces per wave and round, one per lane, until the pieces start behind the tile
			const uint32_t mag_shift = (d.table & 1u) ? 20u : 8u;          // code set 18: the second magnitude of the long entries
			const int len_tile = (int)d.ncoef;
			// the first round on the pieces fetched a tile ahead -- straight-line code: behind a join with the path that loads further pieces the compiler would wait for
			// the prefetches just issued --, further rounds (a dense tile: more pieces than the workgroup has threads) load theirs
			bool more = dx_tile_round(P, d.first_sub + mine, d.end_sub, T0, T1, len_tile, mag_shift, dump, s_multi, s_long, tile16);
#pragma unroll 1
			for (uint32_t q0 = d.first_sub + (uint32_t)wave * 64u + (uint32_t)NT; more && q0 < d.end_sub; q0 += (uint32_t)NT) {
				DxPieces Q;
				dx_tile_pieces(d, q0 + (uint32_t)lane, entries, chunk_base, Q);
				more = dx_tile_round(Q, q0 + (uint32_t)lane, d.end_sub, T0, T1, len_tile, mag_shift, dump, s_multi, s_long, tile16);
			}
		}
		// The loads issued at the top of this turn -- the next tile's pieces, the record of the one after -- have had the whole decode to arrive: wait for them HERE, in
		// front of the stores below.  Left to the compiler the wait sits where the registers are first used, behind the stores, and vmcnt counts in order: every tile then
		// waited for its own stores to be acknowledged by memory (and, at the top of the decode, for the loads just issued): two exposed memory round trips per tile.
		__builtin_amdgcn_s_waitcnt(DX_WAIT_VMCNT0);
		D = D1; D1 = D2; P = P1;
		__syncthreads();
		if (d.ncoef) {
			// the tile goes out in 16-byte words, times the band's divisor (only the low 16 bits of value x divisor are kept, as in the reference's PIXEL arithmetic),
			// and the image is cleared for the next one: chunks of 64 blocks, dealt to the waves in turn
			uint4 *dst = (uint4 *)d.dst;
			const uint32_t n16 = d.ncoef / 8;
			const uint32_t quant2 = (d.quant & 0xffffu) * 0x10001u;
			const uint4 zero = { 0u, 0u, 0u, 0u };
			unsigned long long *const tmasks = d.masks;
			// (all of a wave's chunks are read -- and cleared -- before the first is looked at: one LDS round trip per tile, not one per chunk)
			enum { NCH = DX_TILE / 512, NW = NT / 64, NIT = (NCH + NW - 1) / NW };
			uint4 vv[NIT];
#pragma unroll
			for (int k = 0; k < NIT; k++) {
				const uint32_t it = (uint32_t)wave + (uint32_t)(k * NW);
				vv[k] = zero;
				if (it < (uint32_t)NCH) { vv[k] = ((const uint4 *)s_tile)[it * 64u + (uint32_t)lane]; ((uint4 *)s_tile)[it * 64u + (uint32_t)lane] = zero; }
			}
#pragma unroll
			for (int k = 0; k < NIT; k++) {
				const uint32_t it = (uint32_t)wave + (uint32_t)(k * NW);
				if (it >= (uint32_t)NCH || it * 64u >= n16) continue;      // (behind the tile's end: cleared, not stored)
				const uint32_t i = it * 64u + (uint32_t)lane;
				uint4 v = vv[k];
				const bool nz = i < n16 && (v.x | v.y | v.z | v.w) != 0u;
				v.x = pk_mulw(v.x, quant2); v.y = pk_mulw(v.y, quant2); v.z = pk_mulw(v.z, quant2); v.w = pk_mulw(v.w, quant2);
				if (tmasks) {
					// ... as block lists (cfhd_core.h dec_block_list_layout): of every chunk of 64 blocks only the blocks that hold a nonzero coefficient, compacted to the
					// chunk's own first places in the band (a lane's rank among the chunk's listed blocks: a ballot + v_mbcnt), and the chunk's occupancy mask.  The inverse
					// level-1 strip kernel gathers them (k_inv_yuv422_strip_blocks); what lies behind a chunk's listed blocks in the band is stale and never read.
					const unsigned long long m = __ballot(nz);
					if (nz) dst[it * 64u + wave_mbcnt(m)] = v;
					if (lane == 0) tmasks[it] = m;
				} else if (i < n16) dst[i] = v;
			}
		}
		__syncthreads();
	}
}

// The difference-coded band of interlaced frames (subband 8 of every channel) after k_dec_tiles: coefficients beyond the peak level take their
// values from the peak table in raster order (Codec/decoder.c:19809 DecodeBandFSM16sNoGapWithPeaks), then every row becomes its running sum
// (decoder.c:20822).  One workgroup per band walks down the rows: every thread holds a run of consecutive columns, block-wide prefix sums give
// the number of peaks in front of it and the sum of the columns in front of it.
enum { DXU_THREADS = 256, DXU_MAX = 16 /* columns per thread: rows of up to 4096 coefficients */, DXU_SPLIT = 32 /* workgroups per band (grid y) */ };
__global__ void __launch_bounds__(DXU_THREADS) k_dec_undiff(const DecDiffJob *jobs, int *errors, int only_peak_bands)
{
	__shared__ uint32_t s_w[2][DXU_THREADS / 64];
	const DecDiffJob job = jobs[blockIdx.x];
	if (!job.band || job.width <= 0) return;
	if (only_peak_bands && !job.level) return;            // (the bands without a peak table were served by k_dec_undiff_rows)
	const int t = threadIdx.x, lane = wave_lane(), wave = t >> 6;
	const int per = (job.width + DXU_THREADS - 1) / DXU_THREADS;
	if (per > DXU_MAX) { if (t == 0) atomic_or_u32((uint32_t *)errors, (uint32_t)DX_ERR_SPACE); return; }
	const int c0 = t * per, c1 = c0 + per < job.width ? c0 + per : job.width;
	uint32_t peaks_seen = 0;
	// the rows are independent unless the band has a peak table (whose values are consumed in raster order: rare, one workgroup walks the band);
	// otherwise the gridDim.y workgroups of the band share its rows
	if (job.level && blockIdx.y) return;
	const int rows_per = (job.height + (int)gridDim.y - 1) / (int)gridDim.y;
	const int y_first = job.level ? 0 : (int)blockIdx.y * rows_per, y_last = job.level ? job.height : (y_first + rows_per < job.height ? y_first + rows_per : job.height);
	for (int y = y_first; y < y_last; y++) {
		int16_t *line = job.band + (size_t)y * job.pitch;
		int v[DXU_MAX];
		uint32_t marks = 0;
#pragma unroll
		for (int i = 0; i < DXU_MAX; i++) {
			v[i] = (i < per && c0 + i < c1) ? (int)line[c0 + i] : 0;
			if (job.level && (v[i] > job.level || v[i] < -job.level)) marks++;
		}
		if (job.level) {
			// peak values in raster order: block-wide exclusive count of the marked coefficients in front of this thread's columns
			const uint32_t incl = wave_incl_scan(marks);
			if (lane == 63) s_w[0][wave] = incl;
			__syncthreads();
			uint32_t before = incl - marks, row_total = 0;
#pragma unroll
			for (int k = 0; k < DXU_THREADS / 64; k++) { const uint32_t x = s_w[0][k]; if (k < wave) before += x; row_total += x; }
			uint32_t at = peaks_seen + before;
#pragma unroll
			for (int i = 0; i < DXU_MAX; i++)
				if (i < per && c0 + i < c1 && (v[i] > job.level || v[i] < -job.level)) {
					if (2u * at + 2u <= job.peak_bytes) v[i] = (int)(int16_t)((uint32_t)job.peaks[2 * at] | ((uint32_t)job.peaks[2 * at + 1] << 8));
					at++;
				}
			peaks_seen += row_total;
		}
		// running sums: inside the thread, then over the threads in front of it (16-bit wrap, as the reference's PIXEL arithmetic)
		int sum = 0;
#pragma unroll
		for (int i = 0; i < DXU_MAX; i++) { sum += v[i]; v[i] = sum; }
		const uint32_t incl = wave_incl_scan((uint32_t)sum);
		if (lane == 63) s_w[1][wave] = incl;
		__syncthreads();
		uint32_t before = incl - (uint32_t)sum;
#pragma unroll
		for (int k = 0; k < DXU_THREADS / 64; k++) if (k < wave) before += s_w[1][k];
#pragma unroll
		for (int i = 0; i < DXU_MAX; i++) if (i < per && c0 + i < c1) line[c0 + i] = (int16_t)((uint32_t)v[i] + before);
		__syncthreads();                                  // s_w is written again by the next row
	}
}

// The same for the bands without a peak table (all but a few torture frames), in the shape the data wants: one wave per band row, a lane holds 16 consecutive
// columns (two 16-byte loads), running sums inside the lane, one wave scan for the lanes in front of it, rows of more than 1024 columns in pieces with a carry.
// No LDS, no barrier, no 2-byte access (k_dec_undiff: 256 threads per row, two barriers per row, 0.85 ms per 512 1080i frames; this one: the bands once in, once out).
// Columns behind the band's width (pitch padding) stay zero.  Bands with a peak table are left to k_dec_undiff(only_peak_bands = 1).
enum { DXR_WAVES = 4 };
__global__ void __launch_bounds__(64 * DXR_WAVES) k_dec_undiff_rows(const DecDiffJob *jobs)
{
	const DecDiffJob job = jobs[blockIdx.x];
	if (!job.band || job.width <= 0 || job.level) return;
	const int lane = wave_lane(), wave = wave_uniform((int)(threadIdx.x >> 6));
	const int step = (int)gridDim.y * DXR_WAVES;
	for (int y = (int)blockIdx.y * DXR_WAVES + wave; y < job.height; y += step) {
		int16_t *line = job.band + (size_t)y * job.pitch;
		uint32_t carry = 0;                                // wave-uniform: the sum of the columns in front of this piece
		for (int c0 = 0; c0 < job.width; c0 += 1024) {
			const int c = c0 + 16 * lane;
			uint4 a = { 0u, 0u, 0u, 0u }, b = { 0u, 0u, 0u, 0u };
			if (c < job.width) a = *(const uint4 *)(line + c);
			if (c + 8 < job.width) b = *(const uint4 *)(line + c + 8);
			const uint32_t w[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
			uint32_t v[16], sum = 0;
#pragma unroll
			for (int i = 0; i < 16; i++) {
				const uint32_t x = (i & 1) ? w[i >> 1] >> 16 : w[i >> 1] & 0xffffu;
				sum += (c + i < job.width) ? x : 0u;           // (16-bit wrap: only the low halves are ever kept)
				v[i] = sum;
			}
			const uint32_t incl = wave_incl_scan(sum), before = carry + incl - sum;
			uint32_t o[8];
#pragma unroll
			for (int i = 0; i < 8; i++) {
				const uint32_t lo = (c + 2 * i < job.width) ? (v[2 * i] + before) & 0xffffu : 0u, hi = (c + 2 * i + 1 < job.width) ? (v[2 * i + 1] + before) & 0xffffu : 0u;
				o[i] = lo | (hi << 16);
			}
			if (c < job.width) { uint4 q; q.x = o[0]; q.y = o[1]; q.z = o[2]; q.w = o[3]; *(uint4 *)(line + c) = q; }
			if (c + 8 < job.width) { uint4 q; q.x = o[4]; q.y = o[5]; q.z = o[6]; q.w = o[7]; *(uint4 *)(line + c + 8) = q; }
			carry += wave_get(incl, 63);
		}
	}
}

} // namespace dev
} // namespace cfhd
