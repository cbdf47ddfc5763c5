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

namespace cfhd {
namespace dev {

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
	DX_TILE = 2048,                   // coefficients per output tile
	DX_THREADS = 256, DX_WAVES = DX_THREADS / 64,
	DX_OFF_INVALID = 31,              // entry: no code word of the true sequence starts in this piece (behind the band end marker / the payload)
};
enum : uint32_t { DX_END = 0xFFFFFFFFu, DX_BAD = 0xFFFFFFFEu, DX_SPECIAL = 0xFFFFFFFEu };
enum { DX_FLAG_END = 1, DX_FLAG_BAD = 2, DX_ERR_BAD = 1 << 1, DX_ERR_OVERFLOW = 1 << 2, DX_ERR_NOEND = 1 << 3, DX_ERR_SPACE = 1 << 4 };
// type field of a long-table entry
enum { DX_T_INVALID = 0, DX_T_RUN = 1, DX_T_VALUE = 2, DX_T_END = 3, DX_T_ESCAPE = 4 };

struct DecIdxTables {
	uint16_t cnt12[1 << DX_K];        // as many whole code words (sign bits included) as fit into the next 12 bits: bits 0-3 bits used (0: none fits), bits 4-15 coefficients covered
	uint16_t sym12[1 << DX_K];        // first code word: bits 0-3 length without the sign bit (0: longer than 12 bits or invalid), bit 4 value (else zero run) -- with length 0: escape --,
	                                  // bits 5-15 run length / index of the magnitude / base of the second-level table
	uint16_t mag_expand[256];         // magnitude after undoing the companding curve, by index
	uint32_t long_tab[DX_LONG_MAX];   // bits 0-4 length (escape: index bits of the next level), bits 5-7 type, bits 8-31 run / magnitude index / base of the next level
	uint32_t nlong;
};

// One coded band of one frame = DecBandJob (cfhd_entropy_kernels.h); the job table is [band slot][frame], a band that is not wanted -- half
// resolution skips level 1 -- has bytes 0.  chunk0 = first chunk of the band in the chunk arrays (k_dec_plan / host).
typedef DecBandJob DxBandJob;
struct DxChunkRec { uint32_t start, end, count, flags; };   // start / end: bit offset of the first code word relative to the chunk's / the next chunk's first bit
struct DxBandSum { uint32_t total; int last_chunk; };        // coefficients the band's code words cover; chunk that holds the band end marker

struct DxSym { int len, type, payload; };                    // len without the sign bit

__device__ __forceinline__ DxSym dx_symbol(const uint16_t *s_sym, const uint32_t *s_long, uint32_t win /* next 32 bits, first bit on top */)
{
	const uint32_t e = s_sym[win >> (32 - DX_K)];
	DxSym s;
	s.len = (int)(e & 15u);
	if (s.len) { s.type = (e & 16u) ? DX_T_VALUE : DX_T_RUN; s.payload = (int)(e >> 5); return s; }
	if (!(e & 16u)) { s.type = DX_T_INVALID; s.payload = 0; return s; }
	uint32_t x = s_long[(e >> 5) + ((win >> (32 - DX_K - DX_L2_BITS)) & ((1u << DX_L2_BITS) - 1u))];
	if (((x >> 5) & 7u) == DX_T_ESCAPE) {
		const int nb = (int)(x & 31u);
		x = s_long[(x >> 8) + ((win << (DX_K + DX_L2_BITS)) >> (32 - nb))];
	}
	s.len = (int)(x & 31u); s.type = (int)((x >> 5) & 7u); s.payload = (int)(x >> 8);
	return s;
}

// The 32 bits that start at bit position p of a word stream held in LDS (words already in big-endian bit order).
__device__ __forceinline__ uint32_t dx_window(const uint32_t *s_words, uint32_t p)
{
	const uint32_t wi = p >> 5, sh = p & 31u;
	const uint32_t w0 = s_words[wi], w1 = s_words[wi + 1];
	return sh ? (w0 << sh) | (w1 >> (32u - sh)) : w0;
}

struct DxLane {                       // state of one lane of k_dec_index
	uint32_t start, end;              // bit positions in staging coordinates (lane t owns [256 t, 256 t + 256)); end may be DX_END / DX_BAD
	uint32_t cnt;                     // coefficients covered by the code words that start in the lane's range
	uint32_t rec_off[DX_SUBS], rec_cnt[DX_SUBS];    // per 64-bit piece: first code word's offset into the piece (DX_OFF_INVALID: none), coefficients of the lane in front of it
};

// Walks the code words of one lane from bit `pos` to the end of the lane's range.  MERGE: stop as soon as the walk reaches a 64-bit mark at
// the offset recorded by the previous walk -- from there on the two chains are the same, only the counts in front shift.
template <bool MERGE>
__device__ __forceinline__ void dx_walk(DxLane &L, uint32_t pos, const uint32_t lane_base, const uint32_t limit, const uint32_t *s_words, const uint16_t *s_cnt,
                                        const uint16_t *s_sym, const uint32_t *s_long)
{
	uint32_t cnt = 0, end = 0;
	bool done = false;
#pragma unroll
	for (int k = 0; k < DX_SUBS; k++) {
		const uint32_t mark = lane_base + (uint32_t)k * DX_SUB_BITS, next = mark + DX_SUB_BITS;
		if (done) { if (!MERGE) L.rec_off[k] = DX_OFF_INVALID; continue; }
		if (pos >= next) { L.rec_off[k] = DX_OFF_INVALID; continue; }       // a walk that starts beyond this piece (the first walk of a piece never does: a code word is shorter than 64 bits)
		if (pos >= limit) { L.rec_off[k] = DX_OFF_INVALID; end = pos; done = true; if (MERGE) for (int j = k + 1; j < DX_SUBS; j++) L.rec_off[j] = DX_OFF_INVALID; continue; }
		const uint32_t off = pos - mark;
		if (MERGE && k > 0 && L.rec_off[k] == off) {
			// same chain from here on: the counts recorded behind this mark move by the difference in front of it
			const uint32_t delta = cnt - L.rec_cnt[k];
#pragma unroll
			for (int j = 0; j < DX_SUBS; j++) if (j >= k && L.rec_off[j] != DX_OFF_INVALID) L.rec_cnt[j] += delta;
			L.cnt += delta;
			return;                                           // L.end stays
		}
		L.rec_off[k] = off; L.rec_cnt[k] = cnt;
		while (pos < next) {
			if (pos >= limit) { end = pos; done = true; break; }
			const uint32_t win = dx_window(s_words, pos);
			const uint32_t m = s_cnt[win >> (32 - DX_K)];
			const uint32_t used = m & 15u;
			if (used && pos + used <= next) { pos += used; cnt += m >> 4; continue; }       // several whole code words, none of them beyond the mark
			const DxSym s = dx_symbol(s_sym, s_long, win);
			if (s.type == DX_T_RUN) { pos += (uint32_t)s.len; cnt += (uint32_t)s.payload; }
			else if (s.type == DX_T_VALUE) { pos += (uint32_t)s.len + 1u; cnt += 1u; }
			else { end = s.type == DX_T_END ? DX_END : DX_BAD; done = true; break; }
		}
		if (MERGE && done) for (int j = k + 1; j < DX_SUBS; j++) L.rec_off[j] = DX_OFF_INVALID;
	}
	L.cnt = cnt;
	L.end = done ? end : pos;
}

// Index of one chunk by one wave.  exact_start: DX_BAD = the run-in lane finds it (speculation, checked by k_dec_chain); else the bit offset
// (relative to the chunk's first bit) at which the chunk's first code word starts.  s_words: DX_STAGE_WORDS words of this wave.
__device__ __forceinline__ void dx_index_chunk(const DxBandJob &job, const uint32_t k, const uint32_t exact_start, uint32_t *s_words, const uint16_t *s_cnt, const uint16_t *s_sym,
                                               const uint32_t *s_long, uint32_t *entries, DxChunkRec *recs)
{
	const int lane = wave_lane();
	const uint32_t nwords = job.bytes >> 2;
	const uint32_t *words = (const uint32_t *)job.bits;
	// staging word i = payload word k * DX_CHUNK_WORDS - 8 + i
	const int64_t first = (int64_t)k * DX_CHUNK_WORDS - (DX_LANE_BITS / 32);
	for (int i = lane; i < DX_STAGE_WORDS; i += 64) {
		const int64_t g = first + i;
		s_words[i] = (g >= 0 && g < (int64_t)nwords) ? bswap32(words[g]) : 0u;
	}
	CFHD_WAVE_SYNC();
	const int64_t left = (int64_t)nwords - first;           // payload words from the start of the staging area on
	const uint32_t limit = left <= 0 ? 0u : (left * 32 > (int64_t)(64 * DX_LANE_BITS + 64) ? (uint32_t)(64 * DX_LANE_BITS + 64) : (uint32_t)(left * 32));   // staging bit position behind the payload
	const uint32_t lane_base = (uint32_t)lane * DX_LANE_BITS;
	const bool runin = exact_start == DX_BAD && k > 0;
	DxLane L;
	L.start = lane_base; L.end = lane_base; L.cnt = 0;
#pragma unroll
	for (int j = 0; j < DX_SUBS; j++) { L.rec_off[j] = DX_OFF_INVALID; L.rec_cnt[j] = 0; }
	if (lane == 0) {
		if (runin) dx_walk<false>(L, lane_base, lane_base, limit, s_words, s_cnt, s_sym, s_long);
		else L.end = DX_LANE_BITS + (exact_start == DX_BAD ? 0u : exact_start);      // the chunk's first code word, exactly (a band's first chunk starts on its first bit)
	} else {
		dx_walk<false>(L, lane_base, lane_base, limit, s_words, s_cnt, s_sym, s_long);
	}
	// every lane takes over where its left neighbour ended, until nothing moves any more
	for (int round = 0; round < 64; round++) {
		const uint32_t ns = __shfl_up(L.end, 1u);
		const bool changed = lane >= 1 && ns != L.start;
		if (!__ballot(changed)) break;
		if (changed) {
			L.start = ns;
			if (ns >= DX_SPECIAL) {                            // behind the band end marker (or a broken code): nothing of the true sequence starts here
				L.end = ns; L.cnt = 0;
#pragma unroll
				for (int j = 0; j < DX_SUBS; j++) L.rec_off[j] = DX_OFF_INVALID;
			} else if (ns < lane_base || ns >= lane_base + DX_LANE_BITS) {   // the neighbour stopped in front of this lane (the payload ended) or reaches over it: nothing starts here
				L.end = ns; L.cnt = 0;
#pragma unroll
				for (int j = 0; j < DX_SUBS; j++) L.rec_off[j] = DX_OFF_INVALID;
			} else dx_walk<true>(L, ns, lane_base, limit, s_words, s_cnt, s_sym, s_long);
		}
	}
	// entries: offset of the first code word | coefficients of the chunk in front of it
	const uint32_t own = lane >= 1 ? L.cnt : 0u;
	const uint32_t incl = wave_incl_scan(own);
	const uint32_t before = incl - own;
	const size_t slot = ((size_t)job.chunk0 + k) * DX_ENTRY_STRIDE + (size_t)lane * DX_SUBS;
	if (lane >= 1) {
		uint4 e;
		uint32_t v[DX_SUBS];
#pragma unroll
		for (int j = 0; j < DX_SUBS; j++) v[j] = L.rec_off[j] == DX_OFF_INVALID ? (uint32_t)DX_OFF_INVALID : (L.rec_off[j] | ((before + L.rec_cnt[j]) << 5));
		e.x = v[0]; e.y = v[1]; e.z = v[2]; e.w = v[3];
		*(uint4 *)(entries + slot) = e;
	}
	const uint32_t total = wave_get(incl, 63), e0 = wave_get(L.end, 0), e63 = wave_get(L.end, 63);
	if (lane == 0) {
		DxChunkRec r;
		r.start = e0 >= DX_SPECIAL ? e0 : e0 - DX_LANE_BITS;
		// a chain that stops in front of the chunk's end without the band end marker ran off the payload
		r.end = e63 >= DX_SPECIAL ? e63 : (e63 < 64u * DX_LANE_BITS ? (uint32_t)DX_BAD : e63 - 64u * DX_LANE_BITS);
		r.count = total;
		r.flags = (r.end == DX_END ? DX_FLAG_END : 0u) | (r.end == DX_BAD ? DX_FLAG_BAD : 0u);
		recs[(size_t)job.chunk0 + k] = r;
	}
	CFHD_WAVE_SYNC();
}

__device__ __forceinline__ void dx_load_tables(const DecIdxTables *T, uint16_t *s_cnt, uint16_t *s_sym, uint32_t *s_long, bool want_cnt)
{
	const uint32_t *c = (const uint32_t *)T->cnt12, *s = (const uint32_t *)T->sym12;
	for (int i = threadIdx.x; i < (1 << DX_K) / 2; i += blockDim.x) { if (want_cnt) ((uint32_t *)s_cnt)[i] = c[i]; ((uint32_t *)s_sym)[i] = s[i]; }
	for (int i = threadIdx.x; i < DX_LONG_MAX; i += blockDim.x) s_long[i] = T->long_tab[i];
}

__device__ __forceinline__ uint32_t dx_nchunks(uint32_t bytes) { return (bytes + DX_CHUNK_BYTES - 1) / DX_CHUNK_BYTES; }

// chunk_job[c] = band job of chunk c; jobs[j].chunk0 = first chunk of job j; counters[0] = number of chunks.  One workgroup.
__global__ void __launch_bounds__(1024) k_dec_plan(DxBandJob *jobs, int njobs, uint32_t *chunk_job, uint32_t max_chunks, uint32_t *counters, int *errors)
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
	if (total > max_chunks) return;
	for (int i = 0; i < per; i++) {
		const int j = t * per + i;
		if (j >= njobs) break;
		const uint32_t n = dx_nchunks(jobs[j].bytes);
		jobs[j].chunk0 = at;
		for (uint32_t c = 0; c < n; c++) chunk_job[at + c] = (uint32_t)j;
		at += n;
	}
}

__global__ void __launch_bounds__(DX_THREADS) k_dec_index(const DxBandJob *jobs, const uint32_t *chunk_job, const uint32_t *counters, const DecIdxTables *T,
                                                          uint32_t *entries, DxChunkRec *recs, int speculate)
{
	__shared__ uint16_t s_cnt[1 << DX_K], s_sym[1 << DX_K];
	__shared__ uint32_t s_long[DX_LONG_MAX];
	__shared__ uint32_t s_words_all[DX_WAVES][DX_STAGE_WORDS];
	dx_load_tables(T, s_cnt, s_sym, s_long, true);
	__syncthreads();
	const uint32_t total = counters[0];
	const int wave = wave_uniform((int)(threadIdx.x >> 6));
	const uint32_t gwave = (uint32_t)blockIdx.x * DX_WAVES + (uint32_t)wave, nwaves = (uint32_t)gridDim.x * DX_WAVES;
	for (uint32_t c = gwave; c < total; c += nwaves) {
		const uint32_t j = chunk_job[c];
		const DxBandJob job = jobs[j];
		const uint32_t k = c - job.chunk0;
		// speculate == 0 (tests): every chunk assumes that a code word starts on its first bit, which is wrong for most of them -- k_dec_chain has to repair them
		dx_index_chunk(job, k, (k == 0 || !speculate) ? 0u : (uint32_t)DX_BAD, s_words_all[wave], s_cnt, s_sym, s_long, entries, recs);
	}
}

// The tables for a wave that has to repair a chunk: loaded by that wave alone, no workgroup barrier (another wave of the workgroup doing the
// same writes the same words).
__device__ __forceinline__ void dx_load_tables_wave(const DecIdxTables *T, uint16_t *s_cnt, uint16_t *s_sym, uint32_t *s_long)
{
	const int lane = wave_lane();
	const uint32_t *c = (const uint32_t *)T->cnt12, *s = (const uint32_t *)T->sym12;
	for (int i = lane; i < (1 << DX_K) / 2; i += 64) { ((uint32_t *)s_cnt)[i] = c[i]; ((uint32_t *)s_sym)[i] = s[i]; }
	for (int i = lane; i < DX_LONG_MAX; i += 64) s_long[i] = T->long_tab[i];
}

// One wave per band: every chunk must start where its predecessor ended; chunk_base[c] = raster position of chunk c's first code word.
__global__ void __launch_bounds__(DX_THREADS) k_dec_chain(const DxBandJob *jobs, int njobs, const DecIdxTables *T, uint32_t *entries, DxChunkRec *recs, uint32_t *chunk_base,
                                                          DxBandSum *sums, int *errors)
{
	__shared__ uint16_t s_cnt[1 << DX_K], s_sym[1 << DX_K];
	__shared__ uint32_t s_long[DX_LONG_MAX];
	__shared__ uint32_t s_words_all[DX_WAVES][DX_STAGE_WORDS];
	const int wave = wave_uniform((int)(threadIdx.x >> 6));
	uint32_t *s_words = s_words_all[wave];
	const int j = (int)blockIdx.x * DX_WAVES + wave, lane = wave_lane();
	if (j >= njobs) return;
	const DxBandJob job = jobs[j];
	const uint32_t nch = dx_nchunks(job.bytes);
	if (nch == 0) { if (lane == 0) { sums[j].total = 0; sums[j].last_chunk = -1; } return; }
	bool tables = false;
	uint32_t prev_end = 0, base = 0;
	int last = -1, err = 0;
	for (uint32_t c0 = 0; c0 < nch && last < 0 && !err; c0 += 64) {
		const uint32_t c = c0 + (uint32_t)lane;
		const bool have = c < nch;
		DxChunkRec r = { 0u, 0u, 0u, 0u };
		if (have) r = recs[(size_t)job.chunk0 + c];
		// a chunk is good when it started where its predecessor ended
		uint32_t pe = __shfl_up(r.end, 1u);
		if (lane == 0) pe = prev_end;
		unsigned long long bad = __ballot(have && r.start != pe);
		while (bad) {                                     // rare: index the first offending chunk again from the exact position, then look again
			const int b = __builtin_ctzll(bad);
			const uint32_t exact = b == 0 ? prev_end : __shfl(r.end, b - 1);
			if (exact >= DX_SPECIAL) break;                 // the predecessor holds the end of the band (or a broken code): what follows is padding
			if (!tables) { dx_load_tables_wave(T, s_cnt, s_sym, s_long); tables = true; CFHD_WAVE_SYNC(); }
			dx_index_chunk(job, c0 + (uint32_t)b, exact, s_words, s_cnt, s_sym, s_long, entries, recs);
			if (have) r = recs[(size_t)job.chunk0 + c];
			pe = __shfl_up(r.end, 1u);
			if (lane == 0) pe = prev_end;
			bad = __ballot(have && r.start != pe) & ~((2ull << b) - 1ull);        // chunks up to b are settled now
		}
		// the band ends in the first chunk that met the end marker (or a broken code); chunks behind it hold padding
		const unsigned long long stop = __ballot(have && (r.flags & (DX_FLAG_END | DX_FLAG_BAD)));
		const int nvalid = stop ? __builtin_ctzll(stop) + 1 : (nch - c0 < 64u ? (int)(nch - c0) : 64);
		const uint32_t cntv = lane < nvalid ? r.count : 0u;
		const uint32_t incl = wave_incl_scan(cntv);
		if (lane < nvalid) chunk_base[(size_t)job.chunk0 + c] = base + incl - cntv;
		base += wave_get(incl, 63);
		if (stop) {
			last = (int)c0 + nvalid - 1;
			if (wave_get(r.flags, nvalid - 1) & DX_FLAG_BAD) err |= DX_ERR_BAD;
		}
		prev_end = wave_get(r.end, 63);
	}
	if (last < 0 && !err) err |= DX_ERR_NOEND;           // ran off the payload without meeting the end marker
	if (base > (uint32_t)job.n) err |= DX_ERR_OVERFLOW;   // more coefficients than the band holds (the tile kernel never writes outside its tile)
	if (lane == 0) {
		sums[j].total = base; sums[j].last_chunk = last < 0 ? (int)nch - 1 : last;
		if (err) atomic_or_u32((uint32_t *)errors, (uint32_t)err);
	}
}

// Tiles of the [slot][frame] job table: tile_cum[s] = tiles in front of slot s (all frames), tiles_per_band[s] = tiles of one band of slot s.
struct DxTilePlan { int nslots, nframes; uint32_t cum[40]; uint32_t per_band[40]; uint32_t total; };

__global__ void __launch_bounds__(DX_THREADS) k_dec_tiles(const DxBandJob *jobs, DxTilePlan plan, const DecIdxTables *T, const uint32_t *entries, const uint32_t *chunk_base,
                                                          const DxBandSum *sums)
{
	__shared__ uint16_t s_sym[1 << DX_K];
	__shared__ uint32_t s_long[DX_LONG_MAX];
	__shared__ uint16_t s_mag[256];
	__shared__ uint32_t s_tile_all[DX_WAVES][DX_TILE / 2];
	dx_load_tables(T, nullptr, s_sym, s_long, false);
	for (int i = threadIdx.x; i < 256; i += blockDim.x) s_mag[i] = T->mag_expand[i];
	const int lane = wave_lane(), wave = wave_uniform((int)(threadIdx.x >> 6));
	uint32_t *s_tile = s_tile_all[wave];
	for (int i = lane; i < DX_TILE / 2; i += 64) s_tile[i] = 0u;
	__syncthreads();
	const uint32_t gwave = (uint32_t)blockIdx.x * DX_WAVES + (uint32_t)wave, nwaves = (uint32_t)gridDim.x * DX_WAVES;
	for (uint32_t t = gwave; t < plan.total; t += nwaves) {
		int slot = 0;
		while (slot + 1 < plan.nslots && t >= plan.cum[slot + 1]) slot++;
		const uint32_t r = t - plan.cum[slot], per = plan.per_band[slot];
		const uint32_t f = r / per, ti = r - f * per;
		const int j = slot * plan.nframes + (int)f;
		const DxBandJob job = jobs[j];
		const uint32_t T0 = ti * DX_TILE, T1 = T0 + DX_TILE < (uint32_t)job.n ? T0 + DX_TILE : (uint32_t)job.n;
		if (job.bytes == 0u || T0 >= (uint32_t)job.n) continue;               // wave-uniform
		const DxBandSum sum = sums[j];
		if (sum.last_chunk >= 0 && T0 < sum.total) {
			// 1. the chunk that holds raster position T0
			const uint32_t nch = (uint32_t)sum.last_chunk + 1u;
			uint32_t kc = 0;
			for (uint32_t c0 = 0; c0 < nch; c0 += 64) {
				const uint32_t c = c0 + (uint32_t)lane;
				const bool le = c < nch && chunk_base[(size_t)job.chunk0 + c] <= T0;
				const unsigned long long m = __ballot(le);
				if (!m) break;
				kc = c0 + (uint32_t)(63 - __builtin_clzll(m));
				if (~m & ((c0 + 64 <= nch) ? ~0ull : ((1ull << (nch - c0)) - 1ull))) break;      // some chunk of this block lies behind T0
			}
			// 2. the 64-bit piece inside it whose first code word is the last one at or in front of T0
			uint32_t first_sub;
			{
				const uint32_t cb = chunk_base[(size_t)job.chunk0 + kc];
				const uint4 e = *(const uint4 *)(entries + ((size_t)job.chunk0 + kc) * DX_ENTRY_STRIDE + (size_t)lane * DX_SUBS);
				const uint32_t v[DX_SUBS] = { e.x, e.y, e.z, e.w };
				uint32_t nle = 0;
#pragma unroll
				for (int s = 0; s < DX_SUBS; s++) {
					const bool le = lane >= 1 && (v[s] & 31u) != (uint32_t)DX_OFF_INVALID && cb + (v[s] >> 5) <= T0;
					nle += (uint32_t)__builtin_popcountll(__ballot(le));
				}
				first_sub = kc * DX_CHUNK_SUBS + (nle ? nle - 1u : 0u);
			}
			// 3. piece by piece, one per lane, until the pieces start behind the tile
			const uint32_t last_sub = nch * DX_CHUNK_SUBS;
			for (uint32_t q0 = first_sub; q0 < last_sub; q0 += 64) {
				const uint32_t q = q0 + (uint32_t)lane;
				bool active = q < last_sub;
				uint32_t ent = DX_OFF_INVALID, idx = 0;
				if (active) {
					const uint32_t kq = q / DX_CHUNK_SUBS, within = q - kq * DX_CHUNK_SUBS;
					ent = entries[((size_t)job.chunk0 + kq) * DX_ENTRY_STRIDE + DX_SUBS + within];
					idx = chunk_base[(size_t)job.chunk0 + kq] + (ent >> 5);
				}
				const uint32_t off = ent & 31u;
				const bool valid = active && off != (uint32_t)DX_OFF_INVALID;
				const bool inside = valid && idx < T1;
				if (inside) {
					// the next 128 bits of the payload from the piece on; the walk needs at most 64 + 26 + 27 of them
					const uint32_t byte0 = q * (DX_SUB_BITS / 8);
					const uint32_t *src = (const uint32_t *)(job.bits + byte0);
					uint32_t d[4];
#pragma unroll
					for (int i = 0; i < 4; i++) d[i] = byte0 + 4u * (uint32_t)i + 4u <= job.bytes ? bswap32(src[i]) : 0u;
					uint64_t acc = (((uint64_t)d[0] << 32) | d[1]) << off;
					int have = 64 - (int)off;
					uint32_t nextw = d[2], afterw = d[3];
					uint32_t pos = off;
					while (pos < (uint32_t)DX_SUB_BITS && idx < T1) {
						if (have < 32) { acc |= (uint64_t)nextw << (32 - have); have += 32; nextw = afterw; afterw = 0u; }
						const DxSym s = dx_symbol(s_sym, s_long, (uint32_t)(acc >> 32));
						if (s.type == DX_T_RUN) { idx += (uint32_t)s.payload; acc <<= s.len; have -= s.len; pos += (uint32_t)s.len; }
						else if (s.type == DX_T_VALUE) {
							const int negative = (int)((acc << s.len) >> 63);
							if (idx >= T0) {
								const int v = (int)s_mag[s.payload] * job.quant;
								((int16_t *)s_tile)[idx - T0] = (int16_t)(negative ? -v : v);
							}
							idx++;
							acc <<= s.len + 1; have -= s.len + 1; pos += (uint32_t)s.len + 1u;
						} else break;                                        // band end marker (or a broken code, reported by k_dec_chain)
					}
				}
				// pieces are in raster order: once a valid one starts behind the tile, all later ones do
				if (__ballot(valid && !inside) || !__ballot(active)) break;
			}
		}
		CFHD_WAVE_SYNC();
		// 4. the tile goes out in 16-byte words and is cleared for the next one
		{
			uint4 *dst = (uint4 *)(job.dst + T0);
			const uint32_t n16 = (T1 - T0) / 8;
			const uint4 zero = { 0u, 0u, 0u, 0u };
			for (uint32_t i = (uint32_t)lane; i < (uint32_t)DX_TILE / 8; i += 64) {
				const uint4 v = ((const uint4 *)s_tile)[i];
				((uint4 *)s_tile)[i] = zero;
				if (i < n16) dst[i] = v;
			}
		}
		CFHD_WAVE_SYNC();
	}
}

} // namespace dev
} // namespace cfhd
