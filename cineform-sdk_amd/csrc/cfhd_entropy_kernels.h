// cfhd_entropy_kernels.h -- GPU run-length / variable-length coding of the quantized bands, bit-exact with the reference's
// host loop (Codec/encoder.c:5386 EncodeQuantLongRuns + band end code :6538 + PadBitsTag), and assembly of the complete
// sample (headers, size chunks, raw lowpass words) in HBM.
//
//   k_ent_count   one workgroup per 2048-coefficient segment: zero-run structure + bits of the tokens that start in it
//   k_ent_scan    one workgroup per band: previous-nonzero max-scan and bit-offset sum-scan over its segments, band size
//   k_ent_layout  one workgroup per frame: payload offsets, header template copy, size-field patches, raw lowpass words,
//                 payload zeroing, trailing zero run + band end marker
//   k_ent_emit    one workgroup per segment: code words assembled in LDS (atomic OR), whole words stored big-endian
//
// A token = the zero run in front of a nonzero coefficient (greedy composite run codes, encoder.c:5488-5545) followed by
// the coefficient's code word (value LUT with cubic companding and sign, :5556-5582).  Runs continue across rows through
// the zeroed pad columns (the reference's `count += gap`, :5640) and across segments (resolved by k_ent_scan).
#pragma once
#include <stdint.h>
#include <cfhd_gfx950.h>

// Timing probes of round 3 (kernels that stop early or skip their stores: they write INVALID samples) exist only in builds made with -DCFHD_AMD_PROBES
// (tools/gpu_probe.sh); the shipped library compiles them out.
#ifdef CFHD_AMD_PROBES
#define CFHD_PROBE(p) (p)
#else
#define CFHD_PROBE(p) 0
#endif

namespace cfhd {
namespace dev {

// A segment = 1024 consecutive raster coefficients of a band = the work of one wave (16 coefficients per lane; 512 was measured
// slower: the per-wave descriptor loads dominate); four waves per
// workgroup, no workgroup barriers in k_ent_count / k_ent_emit: all exchanges are wave-level (ballot / bpermute / shuffles).
#ifndef CFHD_ENT_FILL
#define CFHD_ENT_FILL 16384      // (the emulated tests build with a few words, so that trailers span many pieces)
#endif
#ifndef CFHD_ENT_PER_THREAD
#define CFHD_ENT_PER_THREAD 16
#endif
#ifndef CFHD_ENT_TOK_CAP
#define CFHD_ENT_TOK_CAP 256
#endif
enum { FWD_CHUNK_COLS_ENT = 496 };      // = FWD_CHUNK_COLS of cfhd_kernels.h (static_assert in cfhd_device.hip, which sees both headers)
enum { ENT_THREADS = 256, ENT_LANES = 64, ENT_WAVES = ENT_THREADS / ENT_LANES, ENT_PER_THREAD = CFHD_ENT_PER_THREAD, ENT_SEG = ENT_LANES * ENT_PER_THREAD,
       ENT_LDS_WORDS = 256, ENT_TOK_CAP = CFHD_ENT_TOK_CAP, ENT_MAX_HOLES = 60 /* an intra sample has up to 40, a two-frame group 51; k_ent_layout sums them in one wave */,
       ENT_FILL = CFHD_ENT_FILL /* bytes of a sample one workgroup of k_ent_layout fills at a time */,
       // what k_ent_count leaves per segment for k_ent_emit: one 32-bit word per token -- its finished bit string (run code + value code, left aligned in
       // the upper 26 bits) and its length in the low 6 bits; a token whose run takes several run codes or whose string is longer than 26 bits carries
       // run << 22 | value << 6 | ENT_CODE_COMPLEX instead and k_ent_emit walks the tables for it (rare: long codes belong to large values)
       ENT_TOK_STRIDE = ENT_SEG, ENT_STR_BITS = 26, ENT_CODE_COMPLEX = 63 /* length field: the token's run takes several run codes -- k_ent_emit walks the tables for it */,
       ENT_RUN_COMPLEX = 0xff /* EntSegState::run_size: the run in front of the segment's first token takes several run codes */ };
// ENT_LDS_WORDS: 32-bit words of the per-wave bit window in LDS (a segment of ordinary pictures codes into 10-40 words; beyond the window
// the code words go to the payload with global atomics).  ENT_TOK_CAP: tokens (nonzero coefficients) of a segment held in LDS at a
// time (ordinary: ~80 of 1024; a denser segment is worked off in passes).  Both are sized for occupancy, not for the worst case:
// with worst-case windows (4 KB + 4 KB per wave) LDS capped k_ent_emit at 4 waves per SIMD and the kernel, a chain of dependent
// lookups, ran latency-bound.

struct EntTables {
	uint32_t value_code[2048];     // size << 27 | code word, index = value & 0x7ff
	uint32_t run_bits[3072];       // composite run code for min(run, 3071)
	uint16_t run_total[3072];      // bits the greedy loop spends on a run of this length (all iterations)
	uint16_t run_count[3072];      // zeros covered by run_bits[]
	uint8_t run_size[3072];
	uint32_t band_end_bits; int band_end_size;
	uint2 run_pack[3072];          // x = run_bits, y = run_size | run_count << 8: one load per run code in k_ent_emit
};

struct EntBandJob {
	const int16_t *coeffs;         // band base, rows padded with zeros to `pitch`
	int n;                         // raster length = height * pitch
	int seg_base, nseg;            // its segments in the per-segment arrays
	int frame, hole;               // frame of the batch, hole index in the frame's template
	int table;                     // 0: code set 17 (codebook 1), 1: code set 18 (codebook 2, the difference-coded band of interlaced frames)
};

struct EntSegJob {                 // static per segment: everything k_ent_count / k_ent_emit need to find their coefficients with one scalar load
	const int16_t *coeffs;         // band base
	int n;                         // raster length of the band
	int first;                     // raster index of the segment's first coefficient
	int band;                      // band job index
	int table;                     // entropy table of the band (EntBandJob::table); such a band is also the one that may need a peak table
	int pitch;                     // coefficients per band row (k_ent_count_blocks: a level-1 band's chunks are cut row by row)
	int mask_base;                 // the band's first chunk in a frame's mask array (FwdBlockLists::mask_base); -1: the band has no block lists
};

struct EntSegState {               // per segment, written by k_ent_count / k_ent_scan
	int first_nz, last_nz;         // raster index within the band, -1 when the segment is all zero
	uint32_t bits;                 // k_ent_count: bits of its tokens without the run in front of first_nz; k_ent_scan: with it
	int prev_nz;                   // last nonzero before this segment (-1: none)
	uint32_t bitoff;               // bit offset of its first token relative to the band payload
	uint32_t ntok;                 // k_ent_count: nonzero coefficients of the segment = entries of its token list
	uint32_t run_code, run_size;   // k_ent_scan: the run code in front of the segment's first token (reaches back into earlier segments) when one code covers it; run_size ENT_RUN_COMPLEX otherwise
	uint32_t run_bits;             // k_ent_scan: bits of all the run codes in front of the first token
	// The first bits of the segment's code words (left aligned) and how many of them are known (up to 32): k_ent_count leaves those of its
	// token strings, k_ent_scan puts the first run's code in front.  The segment in FRONT writes the payload word the two share, with these
	// bits merged in, as a plain store: no atomic on the payload for ordinary segments (ent_neighbours_merge()).
	uint32_t lead32, lead_valid;
	// What k_ent_emit would otherwise have to fetch through two more tables, one load behind the other (segment job -> band state): where the band's
	// payload lies (k_ent_layout; null while the sample is not placed or overflowed its buffer) and bit 0: a segment of the same band precedes,
	// bit 1: one follows, bits 8..: the band's entropy table (k_ent_count).  One 64-byte record per segment, three scalar loads per emitting wave.
	uint8_t *out;
	uint32_t info;
	uint32_t peaks;                // a band coded with table 1: k_ent_count: values of the segment beyond the peak threshold (up to ENT_SEG); k_ent_scan: that count << ENT_PEAK_OFF_BITS | those of the band's earlier segments
};
static_assert(sizeof(EntSegState) == 64, "one segment state per 64 bytes");

struct EntBandState { uint32_t seg_bits, tail_run, payload_bytes, base_byte; uint8_t *out; /* payload address in the sample; null until k_ent_layout placed it */
                      uint32_t npeaks, pad; uint8_t *peak_out; /* a band coded with table 1: values beyond the peak threshold (k_ent_scan), where its table's values go (k_ent_layout; null: no table) */ };

struct EntHole { int tmpl_offset, kind, fixed_bytes, band_job; const int16_t *lowpass; int lp_width, lp_height, lp_pitch; };
struct EntPatch { int kind, at_tmpl, at_holes, start_tmpl, start_holes, end_tmpl, end_holes, tag; };

struct EntFrameJob {
	uint8_t *out; uint32_t out_cap;            // sample buffer of this frame
	const uint8_t *tmpl; int tmpl_bytes;       // this frame's header template (frame number / metadata differ per frame)
	const uint8_t *word_holes;                 // [tmpl_bytes / 4] number of holes in front of each template word
	const EntHole *holes; int nholes;          // holes[] is per frame (band_job / lowpass pointers)
	const EntPatch *patches; int npatches;
	uint32_t *sample_bytes;                    // out: size of the finished sample (0 on overflow)
	uint32_t *peak_flag;                       // the frame's word of peak_flags[] (see ENT_PEAK_THRESHOLD)
};

__device__ __forceinline__ uint32_t bswap32(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24); }

// Exclusive block scans over ENT_THREADS values: a scan inside every wave (lane exchange), the waves' totals through LDS -- two barriers per
// scan (the Hillis-Steele loop in LDS they replace took eighteen: k_ent_scan walks the 8100 segments of a level-1 band of an 8K frame in 32
// rounds of two scans, 0.44 ms per launch before, with one workgroup per band).  buf: at least ENT_WAVES words.
__device__ __forceinline__ int block_excl_sum(int v, int *buf, int *total)
{
	const int lane = wave_lane(), wave = (int)(threadIdx.x >> 6);
	const int incl = (int)wave_incl_scan((uint32_t)v);
	if (lane == ENT_LANES - 1) buf[wave] = incl;
	__syncthreads();
	int base = 0, tot = 0;
#pragma unroll
	for (int w = 0; w < ENT_WAVES; w++) { const int x = buf[w]; if (w < wave) base += x; tot += x; }
	if (total) *total = tot;
	__syncthreads();
	return base + incl - v;
}
__device__ __forceinline__ int block_excl_max(int v, int *buf, int *total)
{
	const int lane = wave_lane(), wave = (int)(threadIdx.x >> 6);
	int incl = v;
#pragma unroll
	for (int d = 1; d < ENT_LANES; d <<= 1) { const int x = __shfl_up(incl, (unsigned)d); if (lane >= d && x > incl) incl = x; }
	if (lane == ENT_LANES - 1) buf[wave] = incl;
	int excl = __shfl_up(incl, 1u);
	if (lane == 0) excl = -1;
	__syncthreads();
	int base = -1, tot = -1;
#pragma unroll
	for (int w = 0; w < ENT_WAVES; w++) { const int x = buf[w]; if (w < wave && x > base) base = x; if (x > tot) tot = x; }
	if (total) *total = tot;
	__syncthreads();
	return excl > base ? excl : base;
}

__device__ __forceinline__ uint32_t run_bits_any(const EntTables *T, uint32_t run)
{
	// the greedy loop of the reference (encoder.c:5488-5545) takes the longest run code while at least 3072 zeros are left: in closed form
	// (the all-zero alpha band of an 8K frame is 2700 such steps, which one thread of k_ent_scan used to walk one dependent load at a time)
	uint32_t bits = 0;
	if (run >= 3072u) {
		const uint32_t c = T->run_count[3071], n = (run - 3072u) / c + 1u;
		bits = n * T->run_size[3071]; run -= n * c;
	}
	return bits + T->run_total[run];
}

__device__ __forceinline__ uint32_t value_entry(const EntTables *T, int v)
{
	v = v < -1023 ? -1023 : (v > 1023 ? 1023 : v);       // (one v_med3; a negative value's entry sits at value + 2048)
	return T->value_code[v & 2047];
}

// =============================================================================================
// The segment table describes frame 0 only (every frame of a batch has the same geometry): frame f's segment s uses entry s with the
// coefficient base moved by f pyramids and the band index by f * bands_per_frame, so the table (96 KB at 1080p) stays in L2 instead
// of being streamed once per frame.
struct EntBatchGeom { int segs_per_frame, bands_per_frame; size_t coeff_stride; };
__device__ __forceinline__ EntSegJob ent_seg_job(const EntSegJob *seg_jobs, const EntBatchGeom &g, int seg, int *frame = nullptr)
{
	const int f = seg / g.segs_per_frame, s = seg - f * g.segs_per_frame;
	if (frame) *frame = f;
	EntSegJob job = seg_jobs[s];
	job.coeffs += (size_t)f * g.coeff_stride; job.band += f * g.bands_per_frame;
	return job;
}

// A band coded with table 1 (difference coded) is coded with peaks (EncodeQuantLongRunsPlusPeaks, encoder.c:4802): a coefficient beyond +-ENT_PEAK_THRESHOLD goes
// into the stream as +-(threshold + 1) and its value x divisor into the peak table behind the band (encoder.c:6543-6585), in raster order.  k_ent_count codes the
// clamped value and counts the peaks of its segment, k_ent_scan turns the counts into positions, k_ent_layout sizes the table's hole, writes its chunk header and the
// three tags in front of the band, k_ent_peaks fills the values in.  peak_flags[frame]: bit 0 raised when the frame has a peak at all (statistics).
enum { ENT_PEAK_THRESHOLD = 250, ENT_PEAK_OFF_BITS = 21, ENT_PEAK_MAX = (1 << ENT_PEAK_OFF_BITS) - 1,
       // The table's chunk header counts longwords in 16 bits: the reference writes a table only while (peaks rounded up to even) / 2 <= MAX_CHUNK_SIZE = 0xffff
       // (encoder.c:6557, codec.h:195) and otherwise nothing at all -- no table, the three tags in front of the band left zero, the clamped values stay in the stream.
       ENT_PEAK_TABLE_MAX = 2 * 0xffff };
static_assert(ENT_PEAK_TABLE_MAX <= ENT_PEAK_MAX, "every table that is written has positions for all its peaks");
static_assert(ENT_SEG < (1 << (32 - ENT_PEAK_OFF_BITS)), "a segment's peak count and its position in the band's table share a word");
// The finished bit string of every token (nonzero coefficient) of every segment is kept in `tokens` (ENT_TOK_STRIDE words per segment, the first
// 2 * ntok used): k_ent_emit places those -- a third of the bytes -- instead of reading, compacting and coding the coefficients a second time.
// The coefficients of one segment as lane L of the wave holds them: dwords j * 64 + L (coalesced 4-byte loads), zero beyond the band.
__device__ __forceinline__ void ent_load_segment(const EntSegJob &job, int lane, uint32_t *w)
{
	const int rem = job.n - job.first;                   // coefficients from the segment's start to the end of the band (>= 1)
	const uint32_t *src = (const uint32_t *)(job.coeffs + job.first);
	if (rem >= ENT_SEG) {                                // wave-uniform: every segment of a band but its last
#pragma unroll
		for (int j = 0; j < ENT_SEG / 128; j++) w[j] = CFHD_LDG32(src + j * ENT_LANES + lane);
	} else {
#pragma unroll
		for (int j = 0; j < ENT_SEG / 128; j++) {
			const int d = j * ENT_LANES + lane;
			uint32_t x = 2 * d < rem ? CFHD_LDG32(src + d) : 0u;
			if (2 * d + 1 >= rem) x &= 0xffffu;
			w[j] = x;
		}
	}
}

__device__ __forceinline__ void ent_count_tokens(int seg, const EntSegJob &job, int frame, int ntok, int lane, uint32_t *s_tok, EntSegState *segs, const EntTables *tables,
                                                 uint32_t *peak_flags, uint32_t *tokens, int probe);

// One segment of k_ent_count: the wave's loaded coefficients -> token strings in `tokens`, the segment's state in segs[seg].
// The picture is sparse (about one coefficient in twelve is nonzero), so the code lookups run over a compacted token list, one token per lane
// and round.  Compaction is what this kernel's instructions went into while every lane held 16 consecutive coefficients (a count, a wave scan and
// 16 predicated LDS stores per lane: VALU-bound at 1.7 ms).  With lane L holding the dwords j * 64 + L raster order is j-major, lane, low / high
// half, which is the order of a ballot -- a token's position is the number of nonzero halves in the rounds before (scalar popcounts) plus those in
// the lanes below (v_mbcnt): no scan, no per-lane count.
__device__ __forceinline__ void ent_count_segment(int seg, const EntSegJob &job, int frame, const uint32_t *w, int lane, uint32_t *s_tok, EntSegState *segs, const EntTables *tables,
                                                  uint32_t *peak_flags, uint32_t *tokens, int probe)
{
	if (CFHD_PROBE(probe) == 1) { uint32_t o = 0; for (int j = 0; j < ENT_SEG / 128; j++) o |= w[j]; const unsigned long long q = __ballot(o == 0x12345u); if (lane == 0) { EntSegState &z = segs[seg]; z.first_nz = -1; z.last_nz = -1; z.bits = q == 0x123456789ull; z.ntok = 0; z.lead32 = 0; z.lead_valid = 0; } return; }
	int ntok = 0;                                        // wave-uniform
#pragma unroll
	for (int j = 0; j < ENT_SEG / 128; j++) {
		const uint32_t vl = w[j] & 0xffffu, vh = w[j] >> 16;
		const unsigned long long ml = __ballot(vl != 0u), mh = __ballot(vh != 0u);
		const int pl = ntok + (int)wave_mbcnt(ml) + (int)wave_mbcnt(mh);
		const uint32_t idx = (uint32_t)(j * 128 + 2 * lane) << 16;
		if (vl) s_tok[pl] = idx | vl;
		if (vh) s_tok[pl + (vl != 0u)] = (idx + 0x10000u) | vh;
		ntok += __popcll(ml) + __popcll(mh);
	}
	CFHD_WAVE_SYNC();
	ent_count_tokens(seg, job, frame, ntok, lane, s_tok, segs, tables, peak_flags, tokens, probe);
}

// The second half of a segment's count, from its compacted token list s_tok[0 .. ntok) (local raster index << 16 | value): the tokens' finished bit strings into
// `tokens`, the segment's state into segs[seg].  Shared by k_ent_count (tokens compacted from the dense band) and k_ent_count_blocks (from the block lists).
__device__ __forceinline__ void ent_count_tokens(int seg, const EntSegJob &job, int frame, int ntok, int lane, uint32_t *s_tok, EntSegState *segs, const EntTables *tables,
                                                 uint32_t *peak_flags, uint32_t *tokens, int probe)
{
	const EntTables *T = tables + job.table;
	uint32_t bits = 0, lead32 = 0, lead_valid = 0;
	if (CFHD_PROBE(probe) == 2) { const unsigned long long q = __ballot(s_tok[lane] == 0x12345u); if (lane == 0) { EntSegState &z = segs[seg]; z.first_nz = -1; z.last_nz = -1; z.bits = q == 0x123456789ull && ntok == 0x12345; z.ntok = 0; z.lead32 = 0; z.lead_valid = 0; } return; }
	uint32_t npeaks = 0;                                 // wave-uniform
	for (int t0 = 0; t0 < ntok; t0 += ENT_LANES) {       // wave-uniform
		const int t = t0 + lane;
		const bool have = t < ntok;
		const uint32_t tok = have ? s_tok[t] : 0u;
		const uint32_t before = (have && t > 0) ? s_tok[t - 1] : 0u;
		// zero run in front of the token, inside the segment (< 1024); the run in front of the segment's first token reaches into
		// the earlier segments and is added by k_ent_scan
		const uint32_t run = (have && t > 0) ? (tok >> 16) - (before >> 16) - 1u : 0u;
		int value = (int)(int16_t)(tok & 0xffffu);
		const bool is_peak = job.table && (value > ENT_PEAK_THRESHOLD || value < -ENT_PEAK_THRESHOLD);
		if (is_peak) value = value > 0 ? ENT_PEAK_THRESHOLD + 1 : -ENT_PEAK_THRESHOLD - 1;
		npeaks += (uint32_t)__popcll(__ballot(is_peak));
		const uint32_t ve = value_entry(T, value);
		const uint32_t rt = T->run_total[run];
		const uint2 rp = T->run_pack[run];
		uint32_t my_top = 0, my_len = 0;
		if (have) {
			bits += rt + (ve >> 27);
			uint32_t *seg_out = tokens + (size_t)seg * ENT_TOK_STRIDE;
			// the token's finished bit string for k_ent_emit: run code (when one code covers the run: nearly always) + value code, left aligned;
			// the first token of the segment carries its value code only (its run reaches into the earlier segments: k_ent_scan works that one out)
			const uint32_t vs = ve >> 27, vc = ve & 0x7FFFFFFu, rs = run ? rp.y & 0xffu : 0u;
			const bool simple = (run == 0u || (rp.y >> 8) == run) && rs + vs <= (uint32_t)ENT_STR_BITS;
			const uint32_t str = simple ? (((run ? rp.x : 0u) << vs) | vc) << (32u - rs - vs) : 0u;      // (rs + vs >= 2: a value code has at least its sign)
			const uint32_t len = simple ? rs + vs : (uint32_t)ENT_CODE_COMPLEX;
			const uint32_t rec = simple ? str | len : (run << 22) | (((uint32_t)value & 0xffffu) << 6) | len;
			if (CFHD_PROBE(probe) != 4) seg_out[t] = rec;
			my_top = str; my_len = len;
		}
		if (t0 == 0) {
			// the first 32 bits of the strings of this round, as far as they are plain strings (up to the first token that needs the table walk)
			const int hi = ntok < ENT_LANES ? ntok : ENT_LANES;
			const unsigned long long cx = __ballot(have && my_len == (uint32_t)ENT_CODE_COMPLEX);
			const uint32_t l = (have && my_len != (uint32_t)ENT_CODE_COMPLEX) ? my_len : 0u;
			const uint32_t sc = wave_incl_scan(l), ex = sc - l;
			const int stop_lane = cx ? __builtin_ctzll(cx) : hi;      // strings of lanes 0 .. stop_lane-1 count
			uint32_t part = (lane < stop_lane && ex < 32u) ? my_top >> ex : 0u;
#pragma unroll
			for (int d = 1; d < ENT_LANES; d <<= 1) part |= __shfl_xor(part, d);
			lead32 = part;
			const uint32_t known = stop_lane > 0 ? wave_get(sc, stop_lane - 1) : 0u;
			lead_valid = known < 32u ? known : 32u;
		}
	}
	if (npeaks && lane == 0) atomic_or_u32(&peak_flags[frame], 1u);
	bits = wave_get(wave_incl_scan(bits), ENT_LANES - 1);
	if (lane == 0) {
		EntSegState &s = segs[seg];
		s.first_nz = ntok ? job.first + (int)(s_tok[0] >> 16) : -1;
		s.last_nz = ntok ? job.first + (int)(s_tok[ntok - 1] >> 16) : -1;
		s.bits = bits;
		s.ntok = (uint32_t)ntok;
		s.lead32 = lead32; s.lead_valid = lead_valid;
		s.out = nullptr;
		s.info = (job.first != 0 ? 1u : 0u) | (job.first + ENT_SEG < job.n ? 2u : 0u) | ((uint32_t)job.table << 8);
		s.peaks = npeaks;
	}
}

// ENT_COUNT_SEGS consecutive segments per wave, the loads of all of them issued before the first is worked on.  Measured with 2 (round 3): no change
// (1.33 ms either way) -- the kernel moves 4.2 GB in and 0.7 GB out, 0.81 ms of it are the loads at 5.2 TB/s: it is at the memory system's pace, not
// waiting on latency.  Kept at 1.
enum { ENT_COUNT_SEGS = 1 };
// range_lo / range_n: the launch covers segments range_lo .. range_lo + range_n - 1 of every frame's table (0 / segs_per_frame: all of them); total_segs
// = frames x range_n.  The level-1 bands -- three quarters of the coefficients, final when the level-1 transform is done -- are counted on a second stream
// beside the level-2 / level-3 transforms (GpuEntropyEncoder::launch).
__global__ void __launch_bounds__(ENT_THREADS) k_ent_count(const EntSegJob *seg_jobs, EntBatchGeom geom, int total_segs, EntSegState *segs, const EntTables *tables,
                                                            uint32_t *peak_flags, uint32_t *tokens, int range_lo, int range_n,
                                                            int probe = 0 /* timing experiments: 1 behind the loads, 2 behind the compaction, 4 no token stores */)
{
	__shared__ uint32_t s_tok_all[ENT_WAVES][ENT_SEG];       // the segment's tokens: local raster index << 16 | value (16 bits); every coefficient may be one
	const int lane = wave_lane();
	const int wave = wave_uniform((int)(threadIdx.x >> 6));
	const int idx0 = wave_uniform(((int)blockIdx.x * ENT_WAVES + wave) * ENT_COUNT_SEGS);
	if (idx0 >= total_segs) return;                      // whole wave
	const int seg0 = (idx0 / range_n) * geom.segs_per_frame + range_lo + idx0 % range_n;      // (ENT_COUNT_SEGS == 1: one index, one segment)
	static_assert(ENT_COUNT_SEGS == 1, "the range mapping takes one segment per wave");
	EntSegJob job[ENT_COUNT_SEGS]; int frame[ENT_COUNT_SEGS];
	uint32_t w[ENT_COUNT_SEGS][ENT_SEG / 128];
#pragma unroll
	for (int k = 0; k < ENT_COUNT_SEGS; k++)
		if (idx0 + k < total_segs) { job[k] = ent_seg_job(seg_jobs, geom, seg0 + k, &frame[k]); ent_load_segment(job[k], lane, w[k]); }
#pragma unroll
	for (int k = 0; k < ENT_COUNT_SEGS; k++)
		if (idx0 + k < total_segs) {
			ent_count_segment(seg0 + k, job[k], frame[k], w[k], lane, s_tok_all[wave], segs, tables, peak_flags, tokens, probe);
			CFHD_WAVE_SYNC();                                 // the next segment reuses the token window
		}
}

// k_ent_count over the block lists k_fwd_yuv422_strip_blocks leaves of the level-1 bands (cfhd_kernels.h FwdBlockLists) instead of the dense bands.  A segment's
// 1024 raster coefficients are 128 blocks of 8: lane L takes blocks L and L + 64.  For each it works out the chunk the block lies in (band row, position in the
// row), loads that chunk's occupancy mask and works out the slot its rank in the mask gives a listed block; the listed blocks are then dealt to the lanes in raster
// order (round 6: a pass of 64 lanes takes 64 *listed* blocks) and fetched.  All lanes do this at
// once: two dependent memory round trips per segment (masks, blocks), whatever the number of chunks a segment touches (the first version walked the chunks one
// after the other, five dependent rounds: 1.9 ms per 512 frames where the dense k_ent_count takes 1.6).  The nonzero coefficients then go to the token list in
// raster order (lanes in block order: a scan of the lanes' counts, first half of the segment, then the second).  What the kernel reads are the listed blocks -- a
// third of the band on ordinary pictures -- and 8 bytes per chunk.  Same segment states and token strings as k_ent_count, by construction (the same second half).
struct EntBlockLists { const uint4 *blocks; const unsigned long long *masks; const int16_t *coeff0; size_t masks_per_frame; };
__global__ void __launch_bounds__(ENT_THREADS) k_ent_count_blocks(const EntSegJob *seg_jobs, EntBatchGeom geom, int total_segs, EntSegState *segs, const EntTables *tables,
                                                                   uint32_t *peak_flags, uint32_t *tokens, int range_lo, int range_n, EntBlockLists lists)
{
	__shared__ uint32_t s_tok_all[ENT_WAVES][ENT_SEG];
	const int lane = wave_lane();
	const int wave = wave_uniform((int)(threadIdx.x >> 6));
	const int idx0 = wave_uniform((int)blockIdx.x * ENT_WAVES + wave);
	if (idx0 >= total_segs) return;                      // whole wave
	const int seg = (idx0 / range_n) * geom.segs_per_frame + range_lo + idx0 % range_n;
	int frame;
	const EntSegJob job = ent_seg_job(seg_jobs, geom, seg, &frame);
	uint32_t *s_tok = s_tok_all[wave];
	const int pitch = job.pitch, cpr = (pitch + FWD_CHUNK_COLS_ENT - 1) / FWD_CHUNK_COLS_ENT;
	const int end = job.first + ENT_SEG < job.n ? job.first + ENT_SEG : job.n;
	const uint4 *blocks = wave_uniform_ptr(lists.blocks + (size_t)(job.coeffs - lists.coeff0) / 8);      // first block slot of the band (bands start on 128-byte boundaries)
	const unsigned long long *masks = wave_uniform_ptr(lists.masks + (size_t)frame * lists.masks_per_frame + job.mask_base);
	const int row0 = job.first / pitch, col0 = job.first - row0 * pitch;      // wave-uniform: where the segment starts
	// 1. which of the segment's 128 blocks are listed, and where each lies: lane L looks at blocks L and L + 64 (two chunk masks per lane, the loads side by side)
	uint32_t entry[2]; bool listed[2];
#pragma unroll
	for (int h = 0; h < 2; h++) {
		const int q = lane + 64 * h, pos = job.first + 8 * q;
		int row = row0, col = col0 + 8 * q;
		while (col >= pitch) { col -= pitch; row++; }       // (a segment covers one to three rows of the bands this kernel sees; any number works)
		// chunk k of the row and block i of the chunk: block c8 of the row / 62 as a multiply and a shift (exact below 1092 blocks = rows of 8736 coefficients; the widest level-1 row a FramePlan accepts has 8192: kMaxFrameDim), and
		// every product on the full-rate 24-bit multiplier (the plain forms compile to v_mul_hi / v_mad_u64_u32 at a quarter of the rate: measured 6 % of this kernel)
		static_assert(FWD_CHUNK_COLS_ENT == 62 * 8, "the division below is by 62 blocks");
		const uint32_t c8 = (uint32_t)col >> 3, k = mul_u24(c8, 1058u) >> 16, i = c8 - mul_u24(k, 62u);
		unsigned long long m = 0ull;
		if (pos < end) m = masks[mul_u24((uint32_t)row, (uint32_t)cpr) + k];
		listed[h] = ((m >> i) & 1ull) != 0ull;
		// the slot of the block among the band's blocks (below 2^25: a band of 268 M coefficients) and its number in the segment, one word
		entry[h] = ((mul_u24((uint32_t)row, (uint32_t)pitch >> 3) + mul_u24(k, 62u) + (uint32_t)__popcll(m & ((1ull << i) - 1ull))) << 7) | (uint32_t)q;
	}
	// 2. the listed blocks (a third of them on ordinary pictures), in raster order, dealt to the lanes: a listed block's rank among the listed ones (ballot + v_mbcnt) is the
	//    lane that takes it.  Round 6: before, every lane fetched and compacted its two raster blocks whether listed or not -- two passes of count, scan and eight predicated
	//    stores per segment where one does (the kernel runs at 89 % of the VALU issue rate).  The ranks travel through the head of the wave's token area.
	const unsigned long long b0 = __ballot(listed[0]), b1 = __ballot(listed[1]);
	const int n0 = __popcll(b0), nlisted = n0 + __popcll(b1);             // wave-uniform
	if (listed[0]) s_tok[wave_mbcnt(b0)] = entry[0];
	if (listed[1]) s_tok[n0 + (int)wave_mbcnt(b1)] = entry[1];
	CFHD_WAVE_SYNC();
	const uint32_t mine[2] = { lane < nlisted ? s_tok[lane] : 0u, lane + 64 < nlisted ? s_tok[lane + 64] : 0u };      // (both read before the first token is written over them)
	CFHD_WAVE_SYNC();
	int ntok = 0;                                        // wave-uniform
#pragma unroll
	for (int h = 0; h < 2; h++) {
		if (64 * h >= nlisted) break;                    // wave-uniform: the second pass only for segments with more than 64 listed blocks
		uint32_t w[4] = { 0u, 0u, 0u, 0u };
		if (lane + 64 * h < nlisted) { const cfhd_u4 v = CFHD_LDG128(&blocks[mine[h] >> 7]); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
		uint32_t cnt = 0;
#pragma unroll
		for (int d = 0; d < 4; d++) cnt += ((w[d] & 0xffffu) != 0u) + ((w[d] >> 16) != 0u);
		const uint32_t incl = wave_incl_scan(cnt);
		uint32_t p = (uint32_t)ntok + incl - cnt;
		const uint32_t local = 8u * (mine[h] & 127u);       // raster index of the block's first coefficient within the segment
#pragma unroll
		for (int d = 0; d < 4; d++) {
			const uint32_t vl = w[d] & 0xffffu, vh = w[d] >> 16;
			if (vl) s_tok[p++] = ((local + 2u * d) << 16) | vl;
			if (vh) s_tok[p++] = ((local + 2u * d + 1u) << 16) | vh;
		}
		ntok += (int)wave_get(incl, ENT_LANES - 1);
	}
	CFHD_WAVE_SYNC();
	ent_count_tokens(seg, job, frame, ntok, lane, s_tok, segs, tables, peak_flags, tokens, 0);
}

// =============================================================================================
__global__ void __launch_bounds__(ENT_THREADS) k_ent_scan(const EntBandJob *bands, EntSegState *segs, EntBandState *band_state, const EntTables *tables)
{
	// One workgroup per band, ENT_SCAN_PER consecutive segments per thread and round: a round is a chain of three dependent memory round trips
	// (segment states, run-length table, stores) whatever it covers, and a level-1 band of an 8K frame has 8100 segments.
	enum { ENT_SCAN_PER = 4 };
	__shared__ int s_scan[ENT_THREADS];
	const EntBandJob &job = bands[blockIdx.x];
	const EntTables *T = tables + job.table;
	int carry_prev = -1; uint32_t carry_bits = 0, carry_peaks = 0;
	for (int c0 = 0; c0 < job.nseg; c0 += ENT_THREADS * ENT_SCAN_PER) {
		const int i0 = c0 + (int)threadIdx.x * ENT_SCAN_PER;
		EntSegState s[ENT_SCAN_PER];
#pragma unroll
		for (int k = 0; k < ENT_SCAN_PER; k++) {
			s[k].first_nz = -1; s[k].last_nz = -1; s[k].bits = 0; s[k].lead32 = 0; s[k].lead_valid = 0; s[k].peaks = 0;
			if (i0 + k < job.nseg) s[k] = segs[job.seg_base + i0 + k];
		}
		int mine = -1;
#pragma unroll
		for (int k = 0; k < ENT_SCAN_PER; k++) mine = s[k].last_nz > mine ? s[k].last_nz : mine;
		int chunk_last;
		int prev = block_excl_max(mine, s_scan, &chunk_last);
		if (prev < carry_prev) prev = carry_prev;
		int prevs[ENT_SCAN_PER]; uint32_t bits[ENT_SCAN_PER], rcode[ENT_SCAN_PER], rsize[ENT_SCAN_PER], rbits[ENT_SCAN_PER];
#pragma unroll
		for (int k = 0; k < ENT_SCAN_PER; k++) { prevs[k] = prev; prev = s[k].last_nz > prev ? s[k].last_nz : prev; }
		uint32_t total = 0;
#pragma unroll
		for (int k = 0; k < ENT_SCAN_PER; k++) {
			bits[k] = s[k].bits;
			rcode[k] = 0u; rsize[k] = 0u; rbits[k] = 0u;
			if (s[k].first_nz >= 0) {
				const uint32_t run = (uint32_t)(s[k].first_nz - prevs[k] - 1);
				rbits[k] = run_bits_any(T, run);
				bits[k] += rbits[k];
				if (run > 0u && run < 3072u && T->run_count[run] == run) { rcode[k] = T->run_bits[run]; rsize[k] = T->run_size[run]; }
				else if (run > 0u) rsize[k] = (uint32_t)ENT_RUN_COMPLEX;
				// the segment's first bits: the run code in front of what k_ent_count collected
				const uint32_t rs = rsize[k];
				if (rs == (uint32_t)ENT_RUN_COMPLEX) { s[k].lead32 = 0u; s[k].lead_valid = 0u; }
				else if (rs) {
					s[k].lead32 = (rcode[k] << (32u - rs)) | (s[k].lead32 >> rs);
					s[k].lead_valid = s[k].lead_valid + rs < 32u ? s[k].lead_valid + rs : 32u;
				}
			}
			total += bits[k];
		}
		int chunk_bits;
		uint32_t off = carry_bits + (uint32_t)block_excl_sum((int)total, s_scan, &chunk_bits);
		// a band coded with peaks (uniform): where each segment's peak values go in the band's table
		uint32_t pk_off = 0; int chunk_peaks = 0;
		if (job.table) {
			uint32_t mine_pk = 0;
#pragma unroll
			for (int k = 0; k < ENT_SCAN_PER; k++) mine_pk += s[k].peaks;
			pk_off = carry_peaks + (uint32_t)block_excl_sum((int)mine_pk, s_scan, &chunk_peaks);
		}
#pragma unroll
		for (int k = 0; k < ENT_SCAN_PER; k++) {
			if (i0 + k < job.nseg) {
				EntSegState &o = segs[job.seg_base + i0 + k];
				o.prev_nz = prevs[k]; o.bits = bits[k]; o.bitoff = off; o.run_code = rcode[k]; o.run_size = rsize[k]; o.run_bits = rbits[k]; o.lead32 = s[k].lead32; o.lead_valid = s[k].lead_valid;
				if (job.table) o.peaks = (s[k].peaks << ENT_PEAK_OFF_BITS) | (pk_off < (uint32_t)ENT_PEAK_MAX ? pk_off : (uint32_t)ENT_PEAK_MAX);
			}
			off += bits[k]; pk_off += s[k].peaks;
		}
		carry_peaks += (uint32_t)chunk_peaks;
		if (chunk_last > carry_prev) carry_prev = chunk_last;
		carry_bits += (uint32_t)chunk_bits;
	}
	if (threadIdx.x == 0) {
		EntBandState &b = band_state[blockIdx.x];
		b.seg_bits = carry_bits;
		b.tail_run = (uint32_t)(job.n - 1 - carry_prev);
		uint32_t bits = carry_bits + run_bits_any(T, b.tail_run) + (uint32_t)T->band_end_size;
		b.payload_bytes = ((bits + 31u) >> 5) << 2;
		b.base_byte = 0; b.out = nullptr;
		b.npeaks = carry_peaks; b.pad = 0; b.peak_out = nullptr;
	}
}

// Appends one code word at bit position pos of a big-endian word stream held in (little-endian) memory words; single writer.
__device__ __forceinline__ void put_code_plain(uint32_t *words, uint64_t pos, uint32_t code, int size)
{
	if (!size) return;
	const uint64_t v = (uint64_t)code << (64 - size - (int)(pos & 31));
	uint32_t *w = words + (pos >> 5);
	w[0] |= bswap32((uint32_t)(v >> 32));
	const uint32_t lo = (uint32_t)v;
	if (lo) w[1] |= bswap32(lo);
}

// =============================================================================================
__global__ void __launch_bounds__(ENT_THREADS) k_ent_layout(const EntFrameJob *frames, const EntBandJob *bands, EntSegState *segs,
                                                             EntBandState *band_state, const EntTables *tables)
{
	// gridDim.y workgroups share a frame: every one of them works out the layout (40 holes), workgroup 0 writes the template words and
	// the size fields, and the payload holes -- raw lowpass bands, zero fill and trailer of the coded bands: the bytes of this kernel --
	// are dealt out in pieces of ENT_FILL bytes (an 8K frame has holes of several MB; a batch of eight such frames still fills the chip).
	const EntFrameJob &f = frames[blockIdx.x];
	const int part = blockIdx.y, nparts = gridDim.y;
	__shared__ uint32_t s_cum[ENT_MAX_HOLES + 1];       // bytes of the holes in front of hole h
	__shared__ uint32_t s_piece[ENT_MAX_HOLES + 1];     // pieces of ENT_FILL bytes in front of hole h (step 3)
	__shared__ int s_ok;
	const int tid = threadIdx.x;
	if (tid < 64) {
		// one wave: every lane fetches the size of one hole (the loads overlap), two prefix sums over the lanes
		uint32_t bytes = 0;
		if (tid < f.nholes) {
			const EntHole &hole = f.holes[tid];
			if (hole.kind == 0) bytes = (uint32_t)hole.fixed_bytes;
			else if (hole.kind == 1) bytes = band_state[hole.band_job].payload_bytes;
			else {
				// the peak table of the band in front: chunk header + 16-bit values padded to a whole longword (encoder.c:6543-6585); nothing without peaks
				const uint32_t np = band_state[hole.band_job].npeaks;
				bytes = (np && np <= (uint32_t)ENT_PEAK_TABLE_MAX) ? 4u + 4u * ((np + 1u) >> 1) : 0u;      // (more than the chunk header can count: no table, as the reference)
			}
		}
		const uint32_t pieces = tid < f.nholes ? (bytes / ENT_FILL ? bytes / ENT_FILL : 1u) : 0u;
		const uint32_t cum = wave_incl_scan(bytes), pc = wave_incl_scan(pieces);
		if (tid <= f.nholes && tid <= ENT_MAX_HOLES) { s_cum[tid] = cum - bytes; s_piece[tid] = pc - pieces; }
		const uint32_t total = (uint32_t)f.tmpl_bytes + wave_get(cum, 63);
		if (tid == 0) { s_ok = total <= f.out_cap; if (part == 0) *f.sample_bytes = total <= f.out_cap ? total : 0u; }
	}
	__syncthreads();
	if (!s_ok) return;                                   // uniform: the whole workgroup leaves
	uint32_t *out = (uint32_t *)f.out;
	// 1. fixed words of the template
	const uint32_t *tw = (const uint32_t *)f.tmpl;
	if (part == 0) for (int i = tid; i < f.tmpl_bytes / 4; i += ENT_THREADS) out[i + (s_cum[f.word_holes[i]] >> 2)] = tw[i];
	__syncthreads();
	// 2. size fields
	for (int i = tid; part == 0 && i < f.npatches; i += ENT_THREADS) {
		const EntPatch &p = f.patches[i];
		const uint32_t at = (uint32_t)p.at_tmpl + s_cum[p.at_holes], end = (uint32_t)p.end_tmpl + s_cum[p.end_holes];
		if (p.kind == 0) {
			uint32_t size = (end - at) >> 2; size = size ? size - 1 : 0;
			int tag = p.tag;
			if (tag & 0x2000) { tag |= (int)((size >> 16) & 0xff); size &= 0xffff; } else size &= 0xffff;
			tag = -tag;
			out[at >> 2] = bswap32(((uint32_t)(uint16_t)tag << 16) | size);
		} else if (p.kind == 2) {
			// the three optional tags in front of a band coded with peaks: distance to its table, threshold x divisor (encoder.c:6543-6585); they stay zero without a table
			if (s_cum[p.start_holes + 1] != s_cum[p.start_holes]) {
				const uint32_t offset = end - at;
				out[at >> 2] = bswap32(((uint32_t)(uint16_t)(-75) << 16) | (offset & 0xffffu));                  // TAG_PEAK_TABLE_OFFSET_L
				out[(at >> 2) + 1] = bswap32(((uint32_t)(uint16_t)(-76) << 16) | (offset >> 16));                // TAG_PEAK_TABLE_OFFSET_H
				out[(at >> 2) + 2] = bswap32(((uint32_t)(uint16_t)(-74) << 16) | ((uint32_t)(ENT_PEAK_THRESHOLD * p.tag) & 0xffffu));      // TAG_PEAK_LEVEL
			}
		} else {
			const uint32_t start = (uint32_t)p.start_tmpl + s_cum[p.start_holes];
			out[at >> 2] = bswap32(end - start);
		}
	}
	// 3. payload holes in pieces of ENT_FILL bytes, piece p by workgroup p % nparts (a hole's last piece takes the remainder, so it is never
	//    shorter than ENT_FILL unless it is the whole hole: the trailer of a coded band lies inside it)
	const uint32_t npieces = s_piece[f.nholes];
	int h = 0;
	for (uint32_t p = (uint32_t)part; p < npieces; p += (uint32_t)nparts) {
		while (s_piece[h + 1] <= p) h++;                  // pieces come in rising order
		const EntHole &hole = f.holes[h];
		const uint32_t base = (uint32_t)hole.tmpl_offset + s_cum[h], bytes = s_cum[h + 1] - s_cum[h];
		const uint32_t k = p - s_piece[h], last = s_piece[h + 1] - s_piece[h] - 1;
		const uint32_t w0 = k * (ENT_FILL / 4), w1 = k == last ? bytes / 4 : w0 + ENT_FILL / 4;
		if (hole.kind == 0) {
			// raw lowpass: 16-bit big-endian, row after row without the pitch padding, zero padded to 32 bits (encoder.c:4423-4441)
			const int count = hole.lp_width * hole.lp_height;
			for (uint32_t i = w0 + tid; i < w1; i += ENT_THREADS) {
				uint32_t w = 0;
#pragma unroll
				for (int q = 0; q < 2; q++) {
					const int e = 2 * (int)i + q;
					uint32_t v = 0;
					if (e < count) { const int r = e / hole.lp_width, c = e - r * hole.lp_width; v = (uint16_t)hole.lowpass[(size_t)r * hole.lp_pitch + c]; }
					w = (w << 16) | v;
				}
				out[(base >> 2) + i] = bswap32(w);
			}
			continue;
		}
		if (hole.kind == 2) {
			// peak table: the chunk header (optional tag 0x4001 with the number of longwords that follow), zeros where k_ent_peaks puts the values
			const uint32_t np = band_state[hole.band_job].npeaks;      // (<= ENT_PEAK_TABLE_MAX here: the hole of a larger table has no bytes)
			for (uint32_t i = w0 + tid; i < w1; i += ENT_THREADS) out[(base >> 2) + i] = i == 0 ? bswap32(((uint32_t)(uint16_t)(-0x4001) << 16) | ((np + 1u) >> 1)) : 0u;
			if (k == 0 && tid == 0) band_state[hole.band_job].peak_out = bytes ? f.out + base + 4 : nullptr;
			continue;
		}
		// coded band: k_ent_emit ORs its code words into zeroed words.  Everything from the word that holds the first bit of the trailer
		// (trailing zero run + band end marker) belongs to the workgroup of the last piece, which stores those words whole.
		const EntBandState bs = band_state[hole.band_job];
		const uint32_t tail_word = (uint32_t)(bs.seg_bits >> 5) < bytes / 4 ? (uint32_t)(bs.seg_bits >> 5) : bytes / 4;
		for (uint32_t i = w0 + tid; i < (w1 < tail_word ? w1 : tail_word); i += ENT_THREADS) out[(base >> 2) + i] = 0;
		if (k != last) continue;
		// 4. the trailer.  A band that ends in a long zero run (an empty alpha plane of an 8K frame: 8 million zeros) takes thousands of
		//    copies of the longest run code: the lanes compute the words of that periodic stretch in closed form; the few codes behind it
		//    and the end marker are appended by one thread.
		const EntTables *T = tables + bands[hole.band_job].table;
		const uint32_t maxc = T->run_count[3071], maxs = T->run_size[3071], maxb = T->run_bits[3071];
		const uint32_t copies = bs.tail_run >= 3071u ? (bs.tail_run - 3071u) / maxc + 1u : 0u;
		const uint64_t p0 = bs.seg_bits, p1 = p0 + (uint64_t)copies * maxs;          // the copies cover bits [p0, p1) of the payload
		uint32_t *words = out + (base >> 2);
		for (uint32_t i = tail_word + tid; i < bytes / 4; i += ENT_THREADS) {
			uint32_t w = 0;
			const uint64_t lo = (uint64_t)i * 32u, hi = lo + 32u;
			if (copies && hi > p0 && lo < p1) {
				uint64_t j = lo > p0 ? (lo - p0) / maxs : 0u;                         // first copy that reaches into this word
				for (; j < copies && p0 + j * maxs < hi; j++) {
					const int64_t sh = (int64_t)(hi - (p0 + j * maxs)) - (int64_t)maxs;   // distance of the code's last bit from the word's last bit
					w |= sh >= 0 ? (sh < 32 ? maxb << sh : 0u) : maxb >> (-sh);
				}
			}
			words[i] = bswap32(w);
		}
		__syncthreads();                                  // (uniform: k, last are) the words of the trailer are in place
		for (int i = tid; i < bands[hole.band_job].nseg; i += ENT_THREADS) segs[bands[hole.band_job].seg_base + i].out = f.out + base;      // (k_ent_emit reads it there)
		if (tid == 0) {
			EntBandState &b = band_state[hole.band_job];
			b.base_byte = base; b.out = f.out + base;
			uint64_t pos = p1;
			uint32_t run = bs.tail_run - copies * maxc;
			while (run > 0) {
				const uint32_t idx = run < 3072 ? run : 3071;
				put_code_plain(words, pos, T->run_bits[idx], T->run_size[idx]);
				pos += T->run_size[idx];
				run -= T->run_count[idx];
			}
			put_code_plain(words, pos, T->band_end_bits, T->band_end_size);
		}
	}
}

// =============================================================================================
// ORs a left-aligned bit string of up to 64 bits (zero beyond its length) into a big-endian bit stream at bit position pos.  LDS: words of
// the wave's window (host byte order, swapped on the way out); else the payload itself with global atomics.
__device__ __forceinline__ void ent_put_string(uint32_t *s_words, uint32_t *out, bool use_lds, uint32_t first_word, uint64_t pos, uint64_t str)
{
	const uint32_t sh = (uint32_t)pos & 31u, w = (uint32_t)(pos >> 5);
	const uint64_t top = str >> sh;                                     // words w, w + 1
	const uint32_t a = (uint32_t)(top >> 32), b = (uint32_t)top, c = (uint32_t)(((uint64_t)(uint32_t)str << 32) >> sh);      // c: what the shift pushed out of the second word
	if (use_lds) {
		atomic_or_u32(&s_words[w - first_word], a);
		if (b) atomic_or_u32(&s_words[w - first_word + 1], b);
		if (c) atomic_or_u32(&s_words[w - first_word + 2], c);
	} else {
		atomic_or_u32(&out[w], bswap32(a));
		if (b) atomic_or_u32(&out[w + 1], bswap32(b));
		if (c) atomic_or_u32(&out[w + 2], bswap32(c));
	}
}

// The same for a string of up to 32 bits: two words at most, 32-bit shifts only (what an ordinary token takes).
__device__ __forceinline__ void ent_put_string32(uint32_t *s_words, uint32_t *out, bool use_lds, uint32_t first_word, uint64_t pos, uint32_t str)
{
	const uint32_t sh = (uint32_t)pos & 31u, w = (uint32_t)(pos >> 5);
	const uint32_t a = str >> sh, b = sh ? str << (32u - sh) : 0u;
	if (use_lds) {
		atomic_or_u32(&s_words[w - first_word], a);
		if (b) atomic_or_u32(&s_words[w - first_word + 1], b);
	} else {
		atomic_or_u32(&out[w], bswap32(a));
		if (b) atomic_or_u32(&out[w + 1], bswap32(b));
	}
}

// The run codes of `run` zeros from bit position pos on, by the whole wave: the copies of the longest composite code a run of 3072 zeros or more
// starts with (greedy loop, encoder.c:5488-5545) are written 64 at a time, the rest by lane `owner` (one lane writing hundreds of copies one
// after the other held its wave for > 100 us).  Wave-uniform arguments.
__device__ __forceinline__ void ent_put_long_run(const EntTables *T, uint32_t *s_words, uint32_t *out, bool use_lds, uint32_t first_word, uint64_t pos, uint32_t run, int owner)
{
	const int lane = wave_lane();
	const uint32_t cmax = T->run_count[3071], smax = T->run_size[3071], codemax = T->run_bits[3071];
	const uint32_t nrep = run >= 3072u ? (run - 3072u) / cmax + 1u : 0u;
	for (uint32_t i = (uint32_t)lane; i < nrep; i += ENT_LANES) ent_put_string(s_words, out, use_lds, first_word, pos + (uint64_t)i * smax, (uint64_t)codemax << (64u - smax));
	if (lane == owner) {
		uint32_t left = run - nrep * cmax;
		pos += (uint64_t)nrep * smax;
		while (left > 0u) {
			const uint2 rc = T->run_pack[left < 3072u ? left : 3071u];
			const uint32_t size = rc.y & 0xffu;
			ent_put_string(s_words, out, use_lds, first_word, pos, (uint64_t)rc.x << (64u - size));
			pos += size; left -= rc.y >> 8;
		}
	}
}

// Two consecutive segments a, b of one band share the payload word in which a ends and b begins (unless b begins on a word boundary).  When
// both are ordinary -- at least 32 bits each, so that nobody else touches that word, and short enough for the LDS window -- and b's first bits
// are known, a stores the word whole with b's bits merged in and b leaves it alone; otherwise both OR their part into the zeroed payload with
// atomics.  Both waves evaluate this on the same numbers.  (Atomics on payload lines that other waves fill with plain partial stores cost
// k_ent_emit 0.8 of its 2.0 ms in round 2: profiles/r03_*.)
__device__ __forceinline__ bool ent_ordinary(const EntSegState &st)
{
	return st.bits >= 32u && (uint32_t)((st.bitoff + st.bits - 1u) >> 5) - (uint32_t)(st.bitoff >> 5) + 1u <= (uint32_t)ENT_LDS_WORDS;
}
__device__ __forceinline__ bool ent_neighbours_merge(const EntSegState &a, const EntSegState &b)
{
	const uint32_t o = b.bitoff & 31u;
	return o != 0u && ent_ordinary(a) && ent_ordinary(b) && b.lead_valid >= 32u - o;
}

// One segment of k_ent_emit.  The segment's finished bit strings as k_ent_count left them (one per nonzero coefficient: run code + value code): this
// kernel only adds up their lengths and puts them in place -- no table is consulted for an ordinary token.  (Round 2 looked the codes up here: three
// dependent gathers behind the token load; SQ counters showed the waves parked on memory 79 % of their time.)
// st / pv / nx: the segment's state and its neighbours' in the band (the segments of a band are consecutive): who writes the words shared with them.
__device__ __forceinline__ void ent_emit_segment(const EntSegState &st, const EntSegState &pv, const EntSegState &nx, const uint32_t first_rec, const uint32_t *seg_str, int lane,
                                                 uint32_t *s_words, const EntTables *tables, int probe)
{
	if (st.bits == 0) return;                            // wave-uniform: nothing starts in this segment
	const EntTables *T = tables + (st.info >> 8);
	const bool has_prev = (st.info & 1u) != 0u, has_next = (st.info & 2u) != 0u;
	const bool prev_writes_first = has_prev && ent_neighbours_merge(pv, st);
	const bool merge_next = has_next && ent_neighbours_merge(st, nx);
	uint32_t *out = (uint32_t *)st.out;
	if (!out) return;                                    // the sample overflowed its buffer (k_ent_layout reported size 0)
	if (CFHD_PROBE(probe) == 2) { if (first_rec == 0x12345u && lane == 0) out[0] = 1; return; }
	const int ntok = (int)st.ntok;
	const uint64_t seg_pos = st.bitoff;                  // bit position of the segment inside the band payload
	const uint32_t first_word = (uint32_t)(seg_pos >> 5), last_word = (uint32_t)((seg_pos + st.bits - 1) >> 5);
	const uint32_t nwords = last_word - first_word + 1;
	const bool use_lds = nwords <= ENT_LDS_WORDS;        // wave-uniform
	if (use_lds) for (int i = lane; i < (int)nwords + 2; i += ENT_LANES) s_words[i] = 0;
	CFHD_WAVE_SYNC();
	// 1. the run in front of the first token (it reaches back to the last nonzero of the earlier segments; k_ent_scan left its code)
	if (st.run_size == (uint32_t)ENT_RUN_COMPLEX) ent_put_long_run(T, s_words, out, use_lds, first_word, seg_pos, (uint32_t)(st.first_nz - st.prev_nz - 1), 0);
	else if (st.run_size && lane == 0) ent_put_string(s_words, out, use_lds, first_word, seg_pos, (uint64_t)st.run_code << (64u - st.run_size));
	// 2. one token per lane and round: bit position by a wave scan over the lengths, the string OR-ed into the wave's LDS window
	uint64_t round_pos = seg_pos + st.run_bits;
	for (int t = lane, t0 = 0; t0 < ntok; t0 += ENT_LANES, t += ENT_LANES) {
		const bool have = t < ntok;
		uint32_t rec = t0 == 0 ? first_rec : seg_str[have ? t : 0];
		if (!have) rec = 0u;
		uint32_t len = rec & 63u;
		const bool complex = len == (uint32_t)ENT_CODE_COMPLEX;
		uint32_t run = 0, ve = 0;
		if (__ballot(complex)) {
			// rare: a run inside the segment that one composite code does not cover -- this lane walks the tables for its token
			if (complex) {
				run = rec >> 22;                                               // (zero for the first token: its run is k_ent_scan's)
				ve = value_entry(T, (int)(int16_t)((rec >> 6) & 0xffffu));
				len = (uint32_t)T->run_total[run] + (ve >> 27);
			}
		}
		const uint32_t sc = wave_incl_scan(len);
		uint64_t pos = round_pos + (sc - len);
		round_pos += wave_get(sc, ENT_LANES - 1);
		if (have && !complex) ent_put_string32(s_words, out, use_lds, first_word, pos, rec & ~63u);
		if (complex) {
			uint32_t left = run;
			while (left > 0u) {
				const uint2 rc = T->run_pack[left];
				const uint32_t size = rc.y & 0xffu;
				ent_put_string(s_words, out, use_lds, first_word, pos, (uint64_t)rc.x << (64u - size));
				pos += size; left -= rc.y >> 8;
			}
			ent_put_string(s_words, out, use_lds, first_word, pos, (uint64_t)(ve & 0x7FFFFFFu) << (64u - (ve >> 27)));
		}
	}
	CFHD_WAVE_SYNC();
	if (CFHD_PROBE(probe) == 3) { if (s_words[lane] == 0x12345u) out[0] = 1; return; }
	if (use_lds) {
		// interior words belong to this segment alone: plain coalesced stores; the first and last word may be shared with
		// the neighbouring segments (or the band's trailer): OR them into the zeroed payload
		const bool first_shared = (seg_pos & 31u) != 0u, last_shared = ((seg_pos + st.bits) & 31u) != 0u;
		for (int i = lane; i < (int)nwords; i += ENT_LANES) {
			uint32_t w = s_words[i];
			const bool is_first = i == 0, is_last = i == (int)nwords - 1;
			if (is_first && prev_writes_first) continue;                       // the segment in front stores this word, our bits included
			if (is_last && merge_next) w |= nx.lead32 >> ((seg_pos + st.bits) & 31u);      // the next segment's first bits: the word is complete
			w = bswap32(w);
			if (((is_first && first_shared) || (is_last && last_shared && !merge_next)) && CFHD_PROBE(probe) != 4) { if (w) atomic_or_u32(&out[first_word + i], w); }
			else if (CFHD_PROBE(probe) != 5) out[first_word + i] = w;
		}
	}
}

// ENT_EMIT_SEGS consecutive segments per wave: their states (one more on either side) and first strings are fetched in one round of independent loads before
// the first is worked on.  (Round 3's probes: with one segment per wave and the chain segment job -> band state / neighbour states in front of the first
// useful instruction, that front was 0.58 of this kernel's 1.65 ms; 0.41 with the single round of loads; two segments per wave: front 0.36 but the kernel
// 1.62 instead of 1.56 -- the second segment waits for the first.  Kept at 1.)
// (Round 6: resident waves that walk through the segments with the next segment's loads in flight take the same 1.4 ms -- the kernel is bound by the issue of its
// ~345 mostly scalar instructions per segment, SQ_ACTIVE_INST_ANY = 1.04 per SIMD, not by the latency in front of them: profiles/r06_r_k_ent_emit_as_resident_waves_not_adopted.txt.)
enum { ENT_EMIT_SEGS = 1 };
__global__ void __launch_bounds__(ENT_THREADS) k_ent_emit(int total_segs, const EntSegState *segs, const EntTables *tables, const uint32_t *tokens,
                                                           int probe = 0 /* timing experiments: 1 leave at once, 2 behind the descriptor loads, 3 without the final stores, 4 plain stores for the shared words, 5 no plain stores */)
{
	__shared__ uint32_t s_words_all[ENT_WAVES][ENT_LDS_WORDS + 3];
	if (CFHD_PROBE(probe) == 1) return;
	const int lane = wave_lane();
	const int wave = wave_uniform((int)(threadIdx.x >> 6));
	const int seg0 = wave_uniform(((int)blockIdx.x * ENT_WAVES + wave) * ENT_EMIT_SEGS);
	if (seg0 >= total_segs) return;
	uint32_t first_rec[ENT_EMIT_SEGS];
	EntSegState st[ENT_EMIT_SEGS + 2];                   // st[k + 1] = segment seg0 + k
#pragma unroll
	for (int k = 0; k < ENT_EMIT_SEGS; k++) {            // (issued before anything is known about the segments: at worst 512 bytes each read for nothing)
		const int sg = seg0 + k < total_segs ? seg0 + k : seg0;
		first_rec[k] = tokens[(size_t)sg * ENT_TOK_STRIDE + lane];
	}
#pragma unroll
	for (int k = 0; k < ENT_EMIT_SEGS + 2; k++) {
		int sg = seg0 - 1 + k;
		sg = sg < 0 ? 0 : (sg < total_segs ? sg : total_segs - 1);      // (the records say themselves whether a neighbour belongs to their band)
		st[k] = segs[sg];
	}
#pragma unroll
	for (int k = 0; k < ENT_EMIT_SEGS; k++)
		if (seg0 + k < total_segs) {
			ent_emit_segment(st[k + 1], st[k], st[k + 2], first_rec[k], tokens + (size_t)(seg0 + k) * ENT_TOK_STRIDE, lane, s_words_all[wave], tables, probe);
			CFHD_WAVE_SYNC();                                 // the next segment reuses the window
		}
}


// =============================================================================================
// The values of the peak tables (bands coded with table 1 only; launched for interlaced plans).  One workgroup row per band job and frame; a wave per segment that has
// peaks: the segment's coefficients as k_ent_count holds them (dwords j * 64 + lane: raster order is j, lane, low / high half -- the order of a ballot), each peak's
// value x divisor as a 16-bit word in host order at the position k_ent_scan gave the segment (encoder.c:4860-4890: the PIXEL product, low 16 bits).
// grid: (parts, peak tables per sample, frames); which.hole[y]: the y-th table's hole in every frame's template (band job and divisor are the frame's own).
struct EntPeakHoles { int n; int hole[7]; };
__global__ void __launch_bounds__(ENT_THREADS) k_ent_peaks(const EntFrameJob *frames, EntPeakHoles which, const EntBandJob *bands, const EntSegJob *seg_jobs, EntBatchGeom geom,
                                                            const EntSegState *segs, const EntBandState *band_state)
{
	const EntHole &hole = frames[blockIdx.z].holes[which.hole[blockIdx.y]];
	const int bj = hole.band_job, quant = hole.fixed_bytes;
	const EntBandState bs = band_state[bj];
	if (!bs.npeaks || !bs.peak_out || bs.npeaks > (uint32_t)ENT_PEAK_TABLE_MAX) return;       // uniform: the common case leaves here
	const EntBandJob band = bands[bj];
	int16_t *table = (int16_t *)bs.peak_out;
	const int lane = wave_lane();
	const int wave = wave_uniform((int)(threadIdx.x >> 6));
	for (int sg = (int)blockIdx.x * ENT_WAVES + wave; sg < band.nseg; sg += (int)gridDim.x * ENT_WAVES) {      // wave-uniform
		const uint32_t pk = segs[band.seg_base + sg].peaks;
		if (!(pk >> ENT_PEAK_OFF_BITS)) continue;
		int frame;
		const EntSegJob job = ent_seg_job(seg_jobs, geom, band.seg_base + sg, &frame);
		uint32_t w[ENT_SEG / 128];
		ent_load_segment(job, lane, w);
		uint32_t at = pk & (uint32_t)ENT_PEAK_MAX;
#pragma unroll
		for (int j = 0; j < ENT_SEG / 128; j++) {
			const int vl = (int)(int16_t)(w[j] & 0xffffu), vh = (int)(int16_t)(w[j] >> 16);
			const bool pl = vl > ENT_PEAK_THRESHOLD || vl < -ENT_PEAK_THRESHOLD, ph = vh > ENT_PEAK_THRESHOLD || vh < -ENT_PEAK_THRESHOLD;
			const unsigned long long ml = __ballot(pl), mh = __ballot(ph);
			const uint32_t mine = at + wave_mbcnt(ml) + wave_mbcnt(mh);
			if (pl) table[mine] = (int16_t)(vl * quant);
			if (ph) table[mine + (pl ? 1u : 0u)] = (int16_t)(vh * quant);
			at += (uint32_t)(__popcll(ml) + __popcll(mh));
		}
	}
}

// =============================================================================================
// The finished samples sit in fixed-stride slots (sized for the worst case); the host wants them as bytes.  k_ent_pack_offsets
// turns the sample sizes into 64-byte aligned offsets of a dense buffer, k_ent_pack copies every sample there, and one D2H
// copy of offsets[n] bytes replaces one copy per frame.
// =============================================================================================
__global__ void __launch_bounds__(ENT_THREADS) k_ent_pack_offsets(const uint32_t *sizes, int n, uint32_t *offsets)
{
	__shared__ int s_scan[ENT_THREADS];
	uint32_t carry = 0;
	for (int c0 = 0; c0 < n; c0 += ENT_THREADS) {
		const int i = c0 + (int)threadIdx.x;
		const uint32_t sz = i < n ? (sizes[i] + 63u) & ~63u : 0u;
		int total;
		const uint32_t off = carry + (uint32_t)block_excl_sum((int)sz, s_scan, &total);
		if (i < n) offsets[i] = off;
		carry += (uint32_t)total;
	}
	if (threadIdx.x == 0) offsets[n] = carry;
}

__global__ void __launch_bounds__(ENT_THREADS) k_ent_pack(const uint8_t *samples, size_t stride, const uint32_t *sizes, const uint32_t *offsets, uint8_t *packed)
{
	const int f = blockIdx.y;
	const uint4 *src = (const uint4 *)(samples + stride * (size_t)f);
	uint4 *dst = (uint4 *)(packed + offsets[f]);
	const uint32_t n16 = (sizes[f] + 15u) >> 4;
	for (uint32_t i = blockIdx.x * ENT_THREADS + threadIdx.x; i < n16; i += gridDim.x * ENT_THREADS) dst[i] = src[i];
}

// =============================================================================================
// Decoder side: one lane per coded band (the code is sequential inside a band; bands are independent because each has its
// own SUBBAND_SIZE chunk, codec.c:1778).  A wave decodes 64 bands of similar size in lock step; the first-level lookup table
// (12 bits) lives in LDS, the second level (code words of 13..26 bits: magnitudes >= 24 and the band end marker) in HBM.
// Equivalent to Codec/decoder.c:19534 DecodeBandFSM16sNoGap with the companding expansion of codebooks.c:1345-1378 and the
// dequantization of decoder.c:20597-20608 folded into the table values (value * quant, 16-bit product).
// =============================================================================================
enum { DEC_K1 = 12, DEC_THREADS = 64 };

// Multi-symbol first level: everything that fits completely (sign bits included) into the next 12 bits, up to two nonzero
// values with the zero runs around them -- the same idea as the reference's nibble FSM (decoder.c:19597-19642: pre-skip, up to
// two values, post-skip per step), sized for one LDS lookup per step instead of two table walks per byte.
struct DecMulti {             // 8 bytes
	uint16_t pre_bits;        // bits 0-3: bits consumed (0: no complete symbol in the window -> single-symbol path), bits 4-15: zeros before v1
	int16_t v1;               // expanded signed magnitude (0: none)
	uint16_t mid_post;        // bits 0-7: zeros between v1 and v2, bits 8-15: zeros after the last value
	int16_t v2;
};

struct DecTables {
	DecMulti multi[1 << DEC_K1];
	uint32_t lut1[1 << DEC_K1];   // bits 0-4 code length (31: continue in lut2, base = e >> 10, extra index bits = (e >> 5) & 31)
	uint32_t lut2_size;           // bits 5-15 zero run, bits 16-31 expanded magnitude (0xffff: band end)
	uint32_t lut2[1];             // variable length
};

struct DecBandJob {
	const uint8_t *bits; uint32_t bytes;      // coded payload (after BAND_HEADER, before BAND_TRAILER); 4-byte aligned
	int16_t *dst; int n;                      // band raster (height * pitch), zeroed beforehand
	int quant;
	uint32_t chunk0;                          // cfhd_dec_kernels.h: first chunk of the band in the chunk arrays
	int table;                                // 0: code set 17 (cubic companding), 1: code set 18 (linear; the difference-coded band of interlaced frames)
};

// The difference-coded band of one channel of an interlaced frame, after its code words were decoded: peak values (if any) and running sums.
struct DecDiffJob { int16_t *band; int width, height, pitch; const uint8_t *peaks; uint32_t peak_bytes; int level; };

struct DecLowpassJob { const uint8_t *src; int16_t *dst; int width, height, pitch, bias; };

__global__ void __launch_bounds__(DEC_THREADS) k_dec_bands(const DecBandJob *jobs, int njobs, const DecTables *T, int *errors)
{
	__shared__ uint2 s_multi[1 << DEC_K1];               // 32 KB
	for (int i = threadIdx.x; i < (1 << DEC_K1); i += DEC_THREADS) s_multi[i] = ((const uint2 *)T->multi)[i];
	__syncthreads();
	const int j = blockIdx.x * DEC_THREADS + threadIdx.x;
	if (j >= njobs) return;
	const DecBandJob job = jobs[j];
	const uint32_t *words = (const uint32_t *)job.bits;
	const uint32_t nwords = job.bytes >> 2;
	// four words in flight ahead of the bit buffer: the refill below never waits for HBM
	uint32_t w0 = nwords > 0 ? words[0] : 0u, w1 = nwords > 1 ? words[1] : 0u, w2 = nwords > 2 ? words[2] : 0u, w3 = nwords > 3 ? words[3] : 0u;
	uint32_t wpos = 0;
	uint64_t acc = 0; int have = 0;
	int idx = 0, err = 0;
	const int quant = job.quant, n = job.n;
	for (;;) {
		if (have <= 32) {
			const uint32_t w = bswap32(w0);
			w0 = w1; w1 = w2; w2 = w3;
			w3 = (wpos + 4 < nwords) ? words[wpos + 4] : 0u;
			wpos++;
			acc |= (uint64_t)w << (32 - have);
			have += 32;
		}
		const uint2 m = s_multi[(uint32_t)(acc >> (64 - DEC_K1))];
		const int used = (int)(m.x & 15u);
		if (used) {
			const int pre = (int)((m.x >> 4) & 0xfffu), v1 = (int)(int16_t)(m.x >> 16);
			const int mid = (int)(m.y & 0xffu), post = (int)((m.y >> 8) & 0xffu), v2 = (int)(int16_t)(m.y >> 16);
			acc <<= used; have -= used;
			idx += pre;
			if (v1) { if (idx >= n) { err = 2; break; } job.dst[idx++] = (int16_t)(v1 * quant); }
			idx += mid;
			if (v2) { if (idx >= n) { err = 2; break; } job.dst[idx++] = (int16_t)(v2 * quant); }
			idx += post;
			if (idx > n) { err = 2; break; }              // runs may only reach the end of the band: idx never grows far enough to wrap
		} else {
			// a code word longer than the window (or the band end marker): resolve it alone through lut1 / lut2
			uint32_t e = T->lut1[(uint32_t)(acc >> (64 - DEC_K1))];
			if ((e & 31u) == 31u) {
				const int nb = (int)((e >> 5) & 31u);
				e = T->lut2[(e >> 10) + (uint32_t)((acc << DEC_K1) >> (64 - nb))];
			}
			const int len = (int)(e & 31u);
			if (len == 0) { err = 1; break; }
			acc <<= len; have -= len;
			const uint32_t mag = e >> 16;
			if (mag == 0xffffu) break;                   // band end marker
			if (mag) {
				const int negative = (int)(acc >> 63);
				acc <<= 1; have -= 1;
				if (idx >= n) { err = 2; break; }
				const int v = (int)mag * quant;
				job.dst[idx++] = (int16_t)(negative ? -v : v);
			} else { idx += (int)((e >> 5) & 0x7ffu); if (idx > n) { err = 2; break; } }
		}
		if (wpos > nwords + 6) { err = 3; break; }       // ran off the payload without meeting the end marker
	}
	if (err) atomic_or_u32((uint32_t *)errors, 1u);
}

// ---------------------------------------------------------------------------------------------
// Parallel decode of one band by one workgroup.  The code has no resynchronisation points, but a decoder's only state is its
// bit position, so two decoders that ever meet on the same bit stay together: the payload is cut into 256-bit subsequences,
// every lane decodes its subsequence from a guessed start (the subsequence boundary), hands the position where it crossed into
// the next subsequence to its neighbour, and lanes whose start turned out different decode again until nothing changes (a
// few rounds; lane 0 always starts from the exact position carried over from the previous 8 KB sequence, so round k makes at
// least the first k subsequences exact).  A prefix sum over the per-lane coefficient counts gives every lane its raster
// position and a last pass writes the dequantized values.  Same result as the one-lane kernel above and as
// Codec/decoder.c:19534; the workgroup also zeroes its band first (the reference memset()s the band, decoder.c:19563).
// ---------------------------------------------------------------------------------------------
// Two instantiations: 256 threads x 256-bit subsequences for throughput (a batch of frames keeps every CU busy with six workgroups), 512
// threads x 128-bit subsequences for latency (a single frame's decode waits for the serial steps of its longest band: 0.37 -> 0.22 ms;
// at 256 frames this shape is 6 % slower).  Both walk 8 KB of payload per step.
enum { DECP_THREADS = 256, DECP_SUB_BITS = 256, DECP_LL_THREADS = 512, DECP_LL_SUB_BITS = 128 };
enum : uint32_t { DECP_END = 0xFFFFFFFFu, DECP_BAD = 0xFFFFFFFEu };

// Exclusive prefix sum over the DECP_THREADS threads of the workgroup: DPP scan inside the waves, the wave totals through LDS (two
// barriers instead of the 2 log2(n) of a Hillis-Steele scan in LDS).
template <int NT>
__device__ __forceinline__ uint32_t decp_excl_sum(uint32_t v, uint32_t *s_wsum /*[NT / 64]*/, uint32_t *total)
{
	const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
	const uint32_t incl = wave_incl_scan(v);
	__syncthreads();                                     // s_wsum may still be read from the previous use
	if (lane == 63) s_wsum[wave] = incl;
	__syncthreads();
	uint32_t before = 0, all = 0;
#pragma unroll
	for (int w = 0; w < NT / 64; w++) { const uint32_t x = s_wsum[w]; if (w < wave) before += x; all += x; }
	*total = all;
	return before + incl - v;
}

struct DecSub { uint32_t end, cnt; };

// Decodes the code words that start in [p, limit) of the sequence held in s_words (big-endian bit order, already byte
// swapped).  Returns where the next code word starts (DECP_END after the band end marker, DECP_BAD for an invalid code) and
// how many coefficients (zero runs included) were produced; WRITE stores the nonzero ones at dst[idx...].
template <bool WRITE>
__device__ __forceinline__ DecSub dec_sub(const uint32_t *s_words, const uint32_t *s_lut1, const DecTables *T, uint32_t p, const uint32_t limit,
                                          int16_t *dst, uint32_t idx, const uint32_t n, const int quant)
{
	uint32_t wi = p >> 5;
	const int sh = (int)(p & 31u);
	uint64_t acc = (((uint64_t)s_words[wi] << 32) | s_words[wi + 1]) << sh;
	int have = 64 - sh;
	wi += 2;
	uint32_t cnt = 0;
	while (p < limit) {
		if (have < 32) { acc |= (uint64_t)s_words[wi++] << (32 - have); have += 32; }
		uint32_t e = s_lut1[(uint32_t)(acc >> (64 - DEC_K1))];
		if ((e & 31u) == 31u) {
			const int nb = (int)((e >> 5) & 31u);
			e = T->lut2[(e >> 10) + (uint32_t)((acc << DEC_K1) >> (64 - nb))];
		}
		int len = (int)(e & 31u);
		if (len == 0) return DecSub{ DECP_BAD, cnt };
		const uint32_t mag = e >> 16;
		if (mag == 0xffffu) return DecSub{ DECP_END, cnt };
		if (mag) {
			if (WRITE) {
				const int negative = (int)((acc << len) >> 63);
				const int v = (int)mag * quant;
				if (idx + cnt < n) dst[idx + cnt] = (int16_t)(negative ? -v : v);
			}
			len++; cnt++;
		} else cnt += (e >> 5) & 0x7ffu;
		acc <<= len; have -= len; p += (uint32_t)len;
	}
	return DecSub{ p, cnt };
}

template <int NT, int SUB_BITS>
__device__ __forceinline__ void dec_bands_par(const DecBandJob *jobs, const DecTables *T, int *errors)
{
	enum { DECP_THREADS = NT, DECP_SUB_BITS = SUB_BITS, DECP_SEQ_BITS = NT * SUB_BITS, DECP_SEQ_WORDS = DECP_SEQ_BITS / 32 };
	__shared__ uint32_t s_lut1[1 << DEC_K1];             // 16 KB
	__shared__ uint32_t s_words[DECP_SEQ_WORDS + 4];     // 8 KB: the current sequence of the payload
	__shared__ uint32_t s_end[DECP_THREADS];
	__shared__ uint32_t s_wsum[DECP_THREADS / 64];
	__shared__ int s_flag[2];
	const int t = threadIdx.x;
	const DecBandJob job = jobs[blockIdx.x];
	const uint32_t *words = (const uint32_t *)job.bits;
	const uint32_t nwords = job.bytes >> 2, n = (uint32_t)job.n;
	for (int i = t; i < (1 << DEC_K1); i += DECP_THREADS) s_lut1[i] = T->lut1[i];
	if (t < 2) s_flag[t] = 0;
	{	// the zero runs are never written: clear the band (16-byte stores; job.dst and job.n are multiples of 8 elements)
		uint4 *z = (uint4 *)job.dst;
		const uint4 zero = { 0u, 0u, 0u, 0u };
		for (uint32_t i = t; i < n / 8; i += DECP_THREADS) z[i] = zero;
	}
	uint32_t carry = 0, base_idx = 0;
	int err = 0;
	for (uint32_t seq0 = 0; ; seq0 += DECP_SEQ_WORDS) {
		if (seq0 >= nwords) { err = 3; break; }          // ran off the payload without meeting the end marker
		__syncthreads();
		for (int i = t; i < DECP_SEQ_WORDS + 4; i += DECP_THREADS) { const uint32_t w = seq0 + (uint32_t)i; s_words[i] = w < nwords ? bswap32(words[w]) : 0u; }
		__syncthreads();
		// only the subsequences that hold payload bits take part; the last of them stops at the end of the payload
		const uint64_t left = (uint64_t)(nwords - seq0) * 32u;
		const uint32_t seq_bits = left < (uint64_t)DECP_SEQ_BITS ? (uint32_t)left : (uint32_t)DECP_SEQ_BITS;
		const int nsub = (int)((seq_bits + DECP_SUB_BITS - 1) / DECP_SUB_BITS);
		const bool active = t < nsub;
		uint32_t start = t ? (uint32_t)t * DECP_SUB_BITS : carry;
		const uint32_t limit = (uint32_t)(t + 1) * DECP_SUB_BITS < seq_bits ? (uint32_t)(t + 1) * DECP_SUB_BITS : seq_bits;
		DecSub r = active ? dec_sub<false>(s_words, s_lut1, T, start, limit, nullptr, 0u, 0u, 0) : DecSub{ DECP_END, 0u };
		for (int round = 0; ; round++) {
			s_end[t] = r.end;
			__syncthreads();
			if (t == 0) s_flag[(round + 1) & 1] = 0;
			const uint32_t ns = t ? s_end[t - 1] : start;
			const bool changed = active && ns != start;
			if (changed) s_flag[round & 1] = 1;
			__syncthreads();
			if (!s_flag[round & 1]) break;
			if (changed) {
				start = ns;
				if (ns >= DECP_BAD) { r.end = ns; r.cnt = 0; }
				else r = dec_sub<false>(s_words, s_lut1, T, start, limit, nullptr, 0u, 0u, 0);
			}
		}
		uint32_t total;
		const uint32_t my_idx = base_idx + decp_excl_sum<NT>(r.cnt, s_wsum, &total);
		const uint32_t last = s_end[nsub - 1];
		if (last == DECP_BAD) { err = 1; break; }
		if (total > n - base_idx) { err = 2; break; }   // more coefficients than the band holds
		if (active && start < DECP_BAD && r.cnt) (void)dec_sub<true>(s_words, s_lut1, T, start, r.end, job.dst, my_idx, n, job.quant);
		if (last == DECP_END) break;
		if (seq_bits < (uint32_t)DECP_SEQ_BITS) { err = 3; break; }   // the payload ended without the band end marker
		carry = last - DECP_SEQ_BITS;
		base_idx += total;
	}
	if (err && t == 0) atomic_or_u32((uint32_t *)errors, 1u << err);
}
__global__ void __launch_bounds__(DECP_THREADS) k_dec_bands_par(const DecBandJob *jobs, const DecTables *T, int *errors) { dec_bands_par<DECP_THREADS, DECP_SUB_BITS>(jobs, T, errors); }
__global__ void __launch_bounds__(DECP_LL_THREADS) k_dec_bands_par_ll(const DecBandJob *jobs, const DecTables *T, int *errors) { dec_bands_par<DECP_LL_THREADS, DECP_LL_SUB_BITS>(jobs, T, errors); }

// ---------------------------------------------------------------------------------------------
// Sample parser on the GPU, for samples that already live in HBM (the batched round trip hands the encoder's output straight
// to the decoder; a device-resident reader would do the same): one lane per sample walks the tag/value stream exactly like
// the host parser (cfhd_bitstream.cpp parse_sample, after Codec/decoder.c:8861-9420 UpdateCodecState/DecodeSampleIntraFrame:
// optional tags are negated, 0x4000 chunks carry a payload to skip, 0x20xx chunks give the size of the band that follows) and
// fills this frame's rows of the band / lowpass job tables.  Anything that does not match the prepared geometry leaves the
// frame's jobs empty and raises the error flag.
// ---------------------------------------------------------------------------------------------
struct DecPlanBand { int width, height, pitch, offset; };
struct DecPlan {
	int width, display_height, encoded_format, num_channels, bands_per_frame;
	DecPlanBand low[4]; int low_bias[4];
	DecPlanBand high[4][3][4];
	int slot[4][3][4];            // launch order of the band jobs: slot-major, the largest bands first
};
enum { DEC_PARSE_THREADS = 64, DEC_ERR_PARSE = 0x100 };   // one wave per workgroup, one workgroup per sample

// One wave per sample: the 64 lanes fetch 256 consecutive bytes of the tag stream with one coalesced load, the (wave-uniform) walk
// reads its tags out of the lanes' registers; a band header and the size chunk in front of it usually come with one fetch.
struct DecTagReader {
	const uint8_t *d; uint64_t base; uint32_t w; int lane;
	__device__ __forceinline__ uint32_t word(uint64_t pos)
	{
		const uint64_t b = pos & ~(uint64_t)255;
		if (b != base) { w = *(const uint32_t *)(d + b + 4 * (size_t)lane); base = b; }
		return bswap32(wave_read(w, (int)((pos >> 2) & 63u)));
	}
};

__global__ void __launch_bounds__(DEC_PARSE_THREADS) k_dec_parse(const uint8_t *samples, size_t sample_stride, const uint32_t *sizes, int nframes, const DecPlan *P,
                                                                 int16_t *coeffs, size_t coeff_stride, DecBandJob *bandjobs, DecLowpassJob *lowjobs, int *errors,
                                                                 DecDiffJob *diffjobs /* [nframes * channels], may be null: progressive samples only */)
{
	const int f = blockIdx.x;                            // every lane walks the same tags; lane 0 writes the jobs
	const int lane = wave_lane();
	const bool writer = lane == 0;
	const uint8_t *d = samples + sample_stride * (size_t)f;
	int16_t *cbase = coeffs + coeff_stride * (size_t)f;
	const uint64_t size = sizes[f];
	const int nch = P->num_channels;
	if (writer) for (int c = 0; c < nch; c++) {
		lowjobs[f * nch + c] = DecLowpassJob{ d, cbase + P->low[c].offset, 0, 0, P->low[c].pitch, 0 };
		for (int lv = 0; lv < 3; lv++)
			for (int b = 1; b < 4; b++) {
				const DecPlanBand pb = P->high[c][lv][b];
				bandjobs[(size_t)P->slot[c][lv][b] * nframes + f] = DecBandJob{ d, 0u, cbase + pb.offset, pb.height * pb.pitch, 1 };
			}
	}
	DecTagReader rd; rd.d = d; rd.base = ~(uint64_t)0; rd.w = 0; rd.lane = lane;
	uint64_t pos = 0, pending_at = 0, peak_base = 0;
	uint32_t pending = 0, seen_low = 0, peak_offset = 0;
	int peak_level = 0;
	if (writer && diffjobs) for (int c = 0; c < nch; c++) diffjobs[f * nch + c] = DecDiffJob{ nullptr, 0, 0, 0, nullptr, 0u, 0 };
	uint64_t seen = 0;                                   // one bit per coded band: up to 4 channels x 9
	int channel = 0, lv = -1, band = 0, bw = 0, bh = 0, bq = 1, bflags = 0, lw = 0, lh = 0;
	int width = 0, height = 0, display_height = 0, num_channels = 0, encoded_format = 0;
	bool bad = size < 4;
	while (!bad && pos + 4 <= size) {
		const uint32_t word = rd.word(pos);
		int tag = (int)(int16_t)(word >> 16);
		const int value = (int)(word & 0xffffu);
		if (tag < 0) tag = -tag;
		pos += 4;
		if (tag & 0x4000) {
			const uint32_t bytes = (tag & 0x2000) ? ((((uint32_t)(tag & 0xff) << 16) | (uint32_t)value) * 4u) : (uint32_t)value * 4u;
			if (pos + bytes > size) { bad = true; break; }
			pos += bytes;
			continue;
		}
		if (tag & 0x2000) {
			if ((tag & 0xff00) == 0x2000) { pending = ((((uint32_t)(tag & 0xff)) << 16) | (uint32_t)value) * 4u; pending_at = pos; }
			continue;
		}
		switch (tag) {
		case 2: pos += 4u * (uint64_t)value; break;                                  // TAG_INDEX
		case 62: channel = value; if (channel >= 4) bad = true; break;               // TAG_CHANNEL
		case 12: num_channels = value; break;                                        // TAG_NUM_CHANNELS
		case 84: encoded_format = value; break;                                      // TAG_ENCODED_FORMAT
		case 20: width = value; break;                                               // TAG_FRAME_WIDTH
		case 21: height = value; break;                                              // TAG_FRAME_HEIGHT
		case 85: display_height = value; break;                                      // TAG_FRAME_DISPLAY_HEIGHT
		case 27: lw = value; break;                                                  // TAG_LOWPASS_WIDTH
		case 28: lh = value; break;                                                  // TAG_LOWPASS_HEIGHT
		case 4:                                                                      // TAG_MARKER
			if (value == 0x0f0f) {                                                   // MARK_COEFF_START: raw lowpass values inside the pending chunk
				const uint64_t end = pending_at + pending;
				const uint64_t bytes = (uint64_t)lw * (uint64_t)lh * 2u;
				if (pending == 0 || pos + bytes > end || end > size || channel >= nch) { bad = true; break; }
				const DecPlanBand ll = P->low[channel];
				if (lw != ll.width || lh != ll.height) { bad = true; break; }
				if (writer) lowjobs[f * nch + channel] = DecLowpassJob{ d + pos, cbase + ll.offset, ll.width, ll.height, ll.pitch, P->low_bias[channel] };
				seen_low |= 1u << channel;
				pos = end; pending = 0;
			}
			break;
		case 38: lv = value - 1; if (lv < 0 || lv >= 3) bad = true; break;          // TAG_WAVELET_NUMBER
		case 48: band = value; if (band < 1 || band > 3) bad = true; bflags = 0; break;   // TAG_BAND_NUMBER
		case 72: bflags = value; break;                                              // TAG_BAND_CODING_FLAGS
		case 75: peak_offset = (peak_offset & ~0xffffu) | (uint32_t)value; peak_base = pos; peak_level = 0; break;     // TAG_PEAK_TABLE_OFFSET_L (decoder.c:23978)
		case 76: peak_offset = (peak_offset & 0xffffu) | ((uint32_t)value << 16); peak_level = 0; break;               // TAG_PEAK_TABLE_OFFSET_H
		case 74: peak_level = value; break;                                          // TAG_PEAK_LEVEL
		case 49: bw = value; break;                                                  // TAG_BAND_WIDTH
		case 50: bh = value; break;                                                  // TAG_BAND_HEIGHT
		case 53: bq = value; break;                                                  // TAG_BAND_QUANTIZATION
		case 55: {                                                                   // TAG_BAND_HEADER: the code words follow, up to the band trailer
			const uint64_t end = pending_at + pending;
			if (lv < 0 || pending == 0 || end < pos + 4 || end > size || channel >= nch || band < 1) { bad = true; break; }
			const DecPlanBand pb = P->high[channel][lv][band];
			const int codebook = bflags & 0xf;
			const bool difference = (bflags >> 4) & 1;
			if (bw != pb.width || bh != pb.height || (pos & 3u) || codebook > 2 || ((codebook == 2 || difference) && !diffjobs)) { bad = true; break; }
			if (peak_level && peak_base + peak_offset + 2 > size) { bad = true; break; }
			if (writer) bandjobs[(size_t)P->slot[channel][lv][band] * nframes + f] = DecBandJob{ d + pos, (uint32_t)(end - 4 - pos), cbase + pb.offset, pb.height * pb.pitch, bq, 0u, codebook == 2 };
			if (writer && difference) diffjobs[f * nch + channel] = DecDiffJob{ cbase + pb.offset, pb.width, pb.height, pb.pitch, peak_level ? d + peak_base + peak_offset : nullptr,
			                                                                    peak_level ? (uint32_t)(size - (peak_base + peak_offset)) : 0u, peak_level };
			peak_level = 0;
			seen |= (uint64_t)1 << ((channel * 3 + lv) * 3 + band - 1);
			pos = end; pending = 0;
			break; }
		default: break;
		}
	}
	if (display_height == 0) display_height = height;
	const uint64_t want = ((uint64_t)1 << (nch * 9)) - 1u;
	if (bad || width != P->width || display_height != P->display_height || encoded_format != P->encoded_format || num_channels != nch
	    || seen != want || seen_low != (1u << nch) - 1u) {
		if (writer) for (int c = 0; c < nch; c++)
			for (int l = 0; l < 3; l++)
				for (int b = 1; b < 4; b++) bandjobs[(size_t)P->slot[c][l][b] * nframes + f].bytes = 0u;
		if (writer) atomic_or_u32((uint32_t *)errors, (uint32_t)DEC_ERR_PARSE);
	}
}

// Raw 16-bit big-endian lowpass coefficients + the reference decoder's bias (decoder.c:12240-12290, :12468-12545).
__global__ void __launch_bounds__(256) k_dec_lowpass(const DecLowpassJob *jobs)
{
	const DecLowpassJob job = jobs[blockIdx.y];
	const int count = job.width * job.height;
	for (int i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) {
		const int r = i / job.width, c = i - r * job.width;
		int v = (int)(int16_t)(((uint32_t)job.src[2 * i] << 8) | job.src[2 * i + 1]);
		if (job.width & 1) v = (int)(uint16_t)v;          // the odd-width path reads 16 unsigned bits (GetBits)
		v += job.bias;
		job.dst[(size_t)r * job.pitch + c] = (int16_t)(v > 0x7fff ? 0x7fff : v);
	}
}

} // namespace dev
} // namespace cfhd
