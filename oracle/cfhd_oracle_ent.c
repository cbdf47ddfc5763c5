/* oracle/cfhd_oracle_ent.c -- TEST INFRASTRUCTURE ONLY (see cfhd_oracle.h).
 *
 * Host-side neighbours of the hot path, restated plainly: quantizer table derivation and the
 * run-length / variable-length coding of one quantized band (code set 17, cubic companding).
 */
#include "cfhd_oracle.h"
#include "codebook_data.h"
#include <stdlib.h>
#include <string.h>

/* =================================== quantizer tables =================================== */

/* Codec/quantize.h:54-65 (quality tables), Codec/quantize.c:186-584 QuantizationSetQuality,
 * Codec/wavelet.c:7022 SetTransformScale, :1710 SetTransformPrescale,
 * Codec/quantize.c:2865-3350 SetTransformQuantization (TRANSFORM_TYPE_SPATIAL, fixed quality,
 * bit-rate limiter idle: first frame / FILMSCAN qualities, quantize.c:2994-3012). */
static const int LumaQ[4][17] = {
	{4, 4,5,5, 4,5,5, 9,8,8,8, 4,4,4, 4,4,4},
	{4, 8,8,12, 8,8,12, 9,12,12,16, 32,32,48, 32,32,48},
	{4, 6,6,8, 6,6,8, 5,8,8,12, 16,16,24, 16,16,24},
	{4, 4,4,6, 4,4,6, 5,8,8,8, 8,8,12, 8,8,12},
};
static const int ChromaQ[4][17] = {
	{4, 4,5,5, 4,5,5, 9,8,8,8, 8,8,8, 8,8,8},
	{4, 8,8,12, 8,8,12, 9,12,12,16, 32,32,48, 32,32,48},
	{4, 6,6,8, 6,6,8, 5,8,8,12, 16,16,32, 16,16,32},
	{4, 6,6,8, 6,6,8, 5,8,8,8, 8,8,16, 8,8,16},
};

void orc_quant_tables(int quality, int precision, int chroma_full_res, int num_channels, int progressive, orc_quant_t *q)
{
	int qL[17], qC[17], qLmax[17], qCmax[17];
	int factor = quality & 0xff;
	int detail = (quality & 0x0e0000) >> 17;
	int rgb_quality = (quality & 0x06000000) >> 25;
	int newQuality, overrate, lowfreqquant = 4, i, ch, k;
	int mpq = detail + 2;
	int FSratelimiter = 0;

	memset(q, 0, sizeof(*q));
	if (rgb_quality > 2) rgb_quality = 2;
	if (mpq > 8) mpq = 0;
	q->midpoint_prequant = mpq;
	q->num_channels = num_channels;
	if (quality & 0x1f00) factor = 5;
	newQuality = factor;
	if (newQuality == 5) FSratelimiter = 8; else if (newQuality == 6) FSratelimiter = 4;
	if (factor < 1 || factor > 10) factor = 0;
	if (factor > 3) factor = 3;
	overrate = factor; if (overrate >= 2) overrate--;
	for (i = 0; i < 17; i++) {
		qL[i] = LumaQ[factor][i]; qLmax[i] = LumaQ[overrate][i];
		qC[i] = chroma_full_res ? LumaQ[factor][i] : ChromaQ[factor][i];
		qCmax[i] = chroma_full_res ? LumaQ[overrate][i] : ChromaQ[overrate][i];
	}
	for (i = 0; i < 17; i++) { qLmax[i] = qL[i] + (qLmax[i] - qL[i]) / 2; qCmax[i] = qC[i] + (qCmax[i] - qC[i]) / 2; }
	if (precision >= 10) {
		int scale = 4 * 16, limiter = FSratelimiter > 16 ? 16 : FSratelimiter;
		if (newQuality == 4) { lowfreqquant = 3; scale = 3 * 16; }
		else if (newQuality >= 5 && newQuality <= 10) { lowfreqquant = 2; scale = 16 + limiter * 2; }
		if (newQuality >= 5 && scale >= 4) scale >>= 1;
		if (newQuality == 10 && scale >= 6) { scale *= 2; scale /= 3; }
		if (newQuality >= 4) for (i = 1; i < 7; i++) qL[i] = qC[i] = qLmax[i] = qCmax[i] = lowfreqquant;
		for (i = 8; i < 17; i++) {
			qL[i] = (qL[i] * scale) >> 4; if (qL[i] < 2) qL[i] = 2;
			qC[i] = (qC[i] * scale) >> 4; if (qC[i] < 2) qC[i] = 2;
			qLmax[i] = (qLmax[i] * 64) >> 4; if (qLmax[i] < 2) qLmax[i] = 2;
			qCmax[i] = (qCmax[i] * 64) >> 4; if (qCmax[i] < 2) qCmax[i] = 2;
		}
		qL[7] = qC[7] = qLmax[7] = qCmax[7] = 4;
	}
	if (precision == 12) {
		int chromagain = (rgb_quality == 0) ? 8 : (rgb_quality == 1 ? 6 : 4);
		if (newQuality >= 4) for (i = 1; i < 7; i++) qL[i] = qC[i] = qLmax[i] = qCmax[i] = lowfreqquant;
		for (i = 4; i < 7; i++) { qL[i] *= 4; qC[i] *= 4; qLmax[i] *= 4; qCmax[i] *= 4; }
		if (FSratelimiter > 16) { chromagain += FSratelimiter - 16; if (chromagain > 8) chromagain = 8; }
		for (i = 11; i < 17; i++) { qL[i] *= 4; qC[i] *= chromagain; qLmax[i] *= 4; qCmax[i] *= chromagain; }
	}
	if (!progressive) {
		if (factor == 2) {
			qLmax[12] = qL[12]; qLmax[13] = qL[13]; qLmax[15] = qL[15]; qLmax[16] = qL[16];
			qCmax[12] = qC[12]; qCmax[13] = qC[13]; qCmax[15] = qC[15]; qCmax[16] = qC[16];
		}
#define TWEAK(a) do { a[11] = a[11] * 3 / 2; a[12] = a[12] * 2 / 3; a[14] = a[14] * 3 / 2; a[15] = a[15] * 2 / 3; } while (0)
		TWEAK(qL); TWEAK(qC); TWEAK(qLmax); TWEAK(qCmax);
#undef TWEAK
	}
	/* goplength == 1 (intra frame) remap, quantize.c:548-565 */
	for (i = 0; i < 3; i++) { qL[7 + i] = qL[11 + i]; qC[7 + i] = qC[11 + i]; qLmax[7 + i] = qLmax[11 + i]; qCmax[7 + i] = qCmax[11 + i]; }

	/* prescale table: wavelet.c:1710-1782 (spatial transform) */
	if (precision == 10) { q->prescale[1] = 2; }
	else if (precision == 12) { q->prescale[1] = 2; q->prescale[2] = 2; }

	for (ch = 0; ch < num_channels; ch++) {
		const int *quant = ch ? qC : qL, *quantMAX = ch ? qCmax : qLmax;
		int scale[3][4];
		int subband = 1, index;
		/* SetTransformScale, spatial: wavelet 0 = {4,2,2,1}; each further level multiplies by the lowpass scale */
		scale[0][0] = 4; scale[0][1] = 2; scale[0][2] = 2; scale[0][3] = 1;
		for (k = 1; k < 3; k++) {
			int s = scale[k - 1][0];
			scale[k][0] = 4 * s; scale[k][1] = 2 * s; scale[k][2] = 2 * s; scale[k][3] = s;
		}
		for (k = 0; k < 3; k++) for (i = 0; i < 4; i++) q->scale[ch][k][i] = scale[k][i];
		/* highest wavelets first: subbands 1..6 */
		for (index = 2; index >= 1; index--) {
			q->quant[ch][index][0] = 1;
			for (i = 1; i < 4; i++) {
				int vscale = (quantMAX[subband] - quant[subband]) * 256 - 256 * quantMAX[subband] + 512 * quant[subband];
				int quantization = (vscale * scale[index][i]) >> 8;
				quantization >>= 2;
				if (mpq) { quantization *= mpq; quantization /= (mpq - 1) * 2; } else quantization /= 2;
				q->quant[ch][index][i] = quantization;
				subband++;
			}
		}
		q->quant[ch][0][0] = 1;
		for (i = 1; i < 4; i++) {
			int vscale = (quantMAX[subband] - quant[subband]) * 256 - 256 * quantMAX[subband] + 512 * quant[subband];
			int quantization = vscale >> 8;
			if (mpq) { quantization *= mpq; quantization /= (mpq - 1) * 2; } else quantization /= 2;
			q->quant[ch][0][i] = quantization;
			subband++;
		}
	}
}

/* =================================== entropy coding =================================== */

typedef struct { uint32_t bits; int size; int count; } rlc_t;
static rlc_t g_runtab[3072];
static uint32_t g_valtab[2048];     /* size<<27 | codeword, index = value & 0x7ff (Codec/vlc.h:71-74) */
static int g_cubic_inv[1025];
static int g_tables_ready = 0;

/* Codec/codebooks.c:1032 FillVleTable (cubic companding :1053-1078), :401 ComputeRunLengthCodeTable,
 * :499 FillRunLengthCodeTable. */
static void build_tables(void)
{
	rlc_t book[8]; int n = 0, i, j;
	if (g_tables_ready) return;
	for (i = 0; i < 1025; i++) g_cubic_inv[i] = 0;
	for (i = 1; i < 256; i++) {
		int mag = i + (int)(((int64_t)i * i * i * 768) >> 24);
		if (mag > 1023) mag = 1023;
		g_cubic_inv[mag] = i;
	}
	{ int last = 0; for (i = 0; i < 1025; i++) { if (g_cubic_inv[i]) last = g_cubic_inv[i]; else g_cubic_inv[i] = last; } }
	for (i = 0; i < 2048; i++) {
		int value = (i & 1024) ? (-1024 + (i & 1023)) : i;
		int mag = abs(value);
		uint32_t code; int size;
		mag = g_cubic_inv[mag];
		if (mag > 255) mag = 255;
		code = cfhd_cs17_mag_code[mag]; size = cfhd_cs17_mag_len[mag];
		if (value > 0) { code = (code << 1) | 0; size++; }
		else if (value < 0) { code = (code << 1) | 1; size++; }
		g_valtab[i] = ((uint32_t)size << 27) | code;
	}
	for (i = 0; i < CFHD_CS17_NUM_RUNS; i++) { book[n].bits = cfhd_cs17_run[i][0]; book[n].size = (int)cfhd_cs17_run[i][1]; book[n].count = (int)cfhd_cs17_run[i][2]; n++; }
	book[n].bits = cfhd_cs17_mag_code[0]; book[n].size = cfhd_cs17_mag_len[0]; book[n].count = 1; n++;
	for (i = 0; i < n; i++) for (j = i + 1; j < n; j++) if (book[i].count < book[j].count) { rlc_t t = book[i]; book[i] = book[j]; book[j] = t; }
	for (i = 0; i < 3072; i++) {
		uint32_t codeword = 0; int codesize = 0, remaining = i;
		for (j = 0; j < n; j++) {
			int rep, k;
			if (remaining == 0) break;
			rep = remaining / book[j].count;
			for (k = 0; k < rep; k++) {
				if (book[j].size > (31 - codesize)) {
					if (codesize) { remaining -= k * book[j].count; goto next; }
					else break;
				}
				codeword = (codeword << book[j].size) | book[j].bits;
				codesize += book[j].size;
			}
			remaining -= k * book[j].count;
		}
next:
		g_runtab[i].bits = codeword; g_runtab[i].size = codesize; g_runtab[i].count = i - remaining;
	}
	g_tables_ready = 1;
}

typedef struct { uint8_t *p; size_t cap, n; uint32_t acc; int free_bits; } bitw_t;
static void putbits(bitw_t *w, uint32_t bits, int nbits)
{
	/* Codec/bitstream.c:819 PutBits: MSB first into a 32-bit accumulator, stored big-endian */
	if (nbits == 0) return;
	if (nbits <= w->free_bits) {
		w->acc = (nbits == 32) ? bits : ((w->acc << nbits) | (bits & ((1u << nbits) - 1)));
		w->free_bits -= nbits;
	} else {
		int rest = nbits - w->free_bits;
		uint32_t hi = (w->free_bits == 0) ? 0 : ((bits >> rest) & ((1u << w->free_bits) - 1));
		w->acc = (w->free_bits == 0) ? w->acc : ((w->acc << w->free_bits) | hi);
		if (w->n + 4 <= w->cap) { w->p[w->n] = (uint8_t)(w->acc >> 24); w->p[w->n + 1] = (uint8_t)(w->acc >> 16); w->p[w->n + 2] = (uint8_t)(w->acc >> 8); w->p[w->n + 3] = (uint8_t)w->acc; }
		w->n += 4;
		w->acc = bits & ((1u << rest) - 1);
		w->free_bits = 32 - rest;
	}
	if (w->free_bits == 0) {
		if (w->n + 4 <= w->cap) { w->p[w->n] = (uint8_t)(w->acc >> 24); w->p[w->n + 1] = (uint8_t)(w->acc >> 16); w->p[w->n + 2] = (uint8_t)(w->acc >> 8); w->p[w->n + 3] = (uint8_t)w->acc; }
		w->n += 4; w->acc = 0; w->free_bits = 32;
	}
}

static void put_run(bitw_t *w, int count)
{
	while (count > 0) {
		int idx = count < 3072 ? count : 3071;
		putbits(w, g_runtab[idx].bits, g_runtab[idx].size);
		count -= g_runtab[idx].count;
	}
}

/* Codec/encoder.c:5386 EncodeQuantLongRuns (+ band end code :6538 and PadBitsTag in
 * PutVideoBandTrailer codec.c:1835). Zero runs continue across rows including the pitch gap (:5640). */
size_t orc_vlc_encode_band(const PIXEL16 *band, int width, int height, int pitch, uint8_t *out, size_t cap)
{
	bitw_t w = { out, cap, 0, 0, 32 };
	int row, count = 0, gap = pitch - width;
	build_tables();
	for (row = 0; row < height; row++) {
		const PIXEL16 *p = band + (size_t)row * pitch;
		int i;
		for (i = 0; i < width; i++) {
			int v = p[i];
			if (v == 0) { count++; continue; }
			put_run(&w, count); count = 0;
			if (v < 0) { if (v <= -1024) v = -1023; v += 2048; } else if (v >= 1024) v = 1023;
			putbits(&w, g_valtab[v] & 0x7FFFFFF, (int)(g_valtab[v] >> 27));
		}
		count += gap;
	}
	put_run(&w, count);
	putbits(&w, cfhd_cs17_band_end[0], (int)cfhd_cs17_band_end[1]);
	if (w.free_bits < 32) putbits(&w, 0, w.free_bits);
	return w.n;
}

/* Bit-serial decode of the same code (functionally what Codec/decoder.c:19534 DecodeBandFSM16sNoGap does
 * with its nibble FSM): magnitude -> cubic expansion (codebooks.c:1345-1378 ScaleFSM) -> * quant
 * (decoder.c:20597-20608 DeQuantFSM, 16-bit product). Returns 0 on success. */
int orc_vlc_decode_band(const uint8_t *in, size_t nbytes, int width, int height, int pitch, int quant, PIXEL16 *band)
{
	size_t nbits = nbytes * 8, pos = 0;
	size_t total = (size_t)height * pitch, idx = 0;    /* index in the padded raster */
	int r;
	(void)width;
	for (r = 0; r < height; r++) memset(band + (size_t)r * pitch, 0, (size_t)pitch * sizeof(PIXEL16));
	for (;;) {
		uint32_t code = 0; int len = 0, found = 0, i;
		while (!found) {
			if (pos >= nbits) return -1;
			code = (code << 1) | ((in[pos >> 3] >> (7 - (pos & 7))) & 1); pos++; len++;
			if (len > 26) return -2;
			if (len == (int)cfhd_cs17_band_end[1] && code == cfhd_cs17_band_end[0]) return 0;
			for (i = 0; i < CFHD_CS17_NUM_RUNS && !found; i++)
				if ((int)cfhd_cs17_run[i][1] == len && cfhd_cs17_run[i][0] == code) { idx += cfhd_cs17_run[i][2]; found = 2; }
			for (i = 0; i < 256 && !found; i++)
				if (cfhd_cs17_mag_len[i] == len && cfhd_cs17_mag_code[i] == code) {
					if (i == 0) { idx++; found = 2; }
					else {
						int sign, mag = i + (int)(((int64_t)i * i * i * 768) >> 24);
						if (pos >= nbits) return -1;
						sign = (in[pos >> 3] >> (7 - (pos & 7))) & 1; pos++;
						if (idx >= total) return -3;
						band[idx++] = (PIXEL16)((sign ? -mag : mag) * quant);
						found = 1;
					}
				}
		}
	}
}


/* =================================== a whole sample's coefficients =================================== */

/* The code words of code sets 17 and 18 are the same (Common/table17.inc / table18.inc: same lengths, same bits); what differs is the
 * meaning of a magnitude index: cubic expansion for 17 (codebooks.c:1345-1378), the index itself for 18.  A binary trie over the 264
 * code words (256 magnitudes, 7 zero runs, the band end marker): one step per payload bit, no search. */
typedef struct { int child[2]; int kind; int value; } trie_t;      /* kind 0: inner node, 1: magnitude index, 2: zero run, 3: band end */
static trie_t g_trie[8192]; static int g_trie_n = 0;
static void trie_add(uint32_t bits, int len, int kind, int value)
{
	int at = 0, i;
	for (i = len - 1; i >= 0; i--) {
		const int b = (int)((bits >> i) & 1u);
		if (!g_trie[at].child[b]) { g_trie[at].child[b] = g_trie_n; memset(&g_trie[g_trie_n], 0, sizeof(trie_t)); g_trie_n++; }
		at = g_trie[at].child[b];
	}
	g_trie[at].kind = kind; g_trie[at].value = value;
}
static void build_trie(void)
{
	int i;
	if (g_trie_n) return;
	memset(&g_trie[0], 0, sizeof(trie_t)); g_trie_n = 1;
	for (i = 0; i < 256; i++) trie_add(cfhd_cs17_mag_code[i], cfhd_cs17_mag_len[i], i == 0 ? 2 : 1, i == 0 ? 1 : i);   /* magnitude 0 = a single zero */
	for (i = 0; i < CFHD_CS17_NUM_RUNS; i++) trie_add(cfhd_cs17_run[i][0], (int)cfhd_cs17_run[i][1], 2, (int)cfhd_cs17_run[i][2]);
	trie_add(cfhd_cs17_band_end[0], (int)cfhd_cs17_band_end[1], 3, 0);
}

/* One coded band from bit 0 of `in` (Codec/decoder.c:19534 DecodeBandFSM16sNoGap / :19809 ...WithPeaks, restated bit-serially): code words until the
 * band end marker; value = sign * expand(index) * quant in 16-bit arithmetic (DeQuantFSM :20597-20608); with a peak table every value whose
 * magnitude exceeds `peak_level` is replaced by the next 16-bit little-endian word of the table (:19870-19873: the table holds finished
 * coefficients); with difference coding every row becomes its running sum afterwards (:20822-20836).  The raster is height x pitch and the
 * zero runs continue through the pitch padding (one long row, :19572).  Returns the number of payload bits consumed (marker included), < 0 on error. */
long orc_decode_band_bits(const uint8_t *in, size_t nbytes, int width, int height, int pitch, int quant, int codebook,
                          const uint8_t *peaks, size_t peak_bytes, int peak_level, int difference, PIXEL16 *band)
{
	const size_t nbits = nbytes * 8, total = (size_t)height * pitch;
	size_t pos = 0, idx = 0, peak_at = 0;
	int r, x;
	build_trie();
	for (r = 0; r < height; r++) memset(band + (size_t)r * pitch, 0, (size_t)pitch * sizeof(PIXEL16));
	for (;;) {
		int at = 0;
		while (g_trie[at].kind == 0) {
			int b;
			if (pos >= nbits) return -1;
			b = (in[pos >> 3] >> (7 - (pos & 7))) & 1; pos++;
			at = g_trie[at].child[b];
			if (!at) return -2;                               /* not a code word */
		}
		if (g_trie[at].kind == 3) break;
		if (g_trie[at].kind == 2) { idx += (size_t)g_trie[at].value; continue; }
		{
			const int i = g_trie[at].value;
			int sign, mag = codebook == 2 ? i : i + (int)(((int64_t)i * i * i * 768) >> 24);
			int v;
			if (pos >= nbits) return -1;
			sign = (in[pos >> 3] >> (7 - (pos & 7))) & 1; pos++;
			if (idx >= total) return -3;
			v = (PIXEL16)((sign ? -mag : mag) * quant);
			if (peak_level && peaks && (v > peak_level || v < -peak_level)) {
				if (2 * peak_at + 2 > peak_bytes) return -4;
				v = (PIXEL16)(uint16_t)(peaks[2 * peak_at] | (peaks[2 * peak_at + 1] << 8)); peak_at++;
			}
			band[idx++] = (PIXEL16)v;
		}
	}
	if (difference)
		for (r = 0; r < height; r++) { PIXEL16 *line = band + (size_t)r * pitch; for (x = 1; x < width; x++) line[x] = (PIXEL16)(line[x] + line[x - 1]); }
	return (long)pos;
}

/* The tag-value walk of an intra-frame sample as far as the coefficients (Codec/decoder.c:23300-24000 UpdateCodecState: one case per tag; codec.h:196-404 the tag
 * numbers; an optional tag is stored negated, :2340-2360; tags with bit 0x4000 carry a payload of `value` -- with bit 0x2000 (tag & 0xff) << 16 | value -- longwords
 * that the walk skips, tags with bit 0x2000 alone are size fields).  The lowpass band of a channel follows the coefficient marker 0x0F0F as 16-bit big-endian words
 * (:12230-12545 adds the output-dependent bias: the caller's business); a highpass band's code words follow its BAND_HEADER tag and end with the band end marker, the
 * walk goes on at the next 32-bit word (bitstream.c AlignBitsTag).  Unlike the product's parser this walk does not look at the size fields at all: where a band ends is
 * where its code words end, as in the reference.
 * dst[channel][wavelet 0..2][band 0..3] / pitch likewise: where the band's raster goes (band 0 of wavelet 2 = the lowpass band; NULL: not wanted, the band is walked
 * over and dropped).  dims[channel][wavelet][band][2] = the width and height the caller's rasters have: a sample that says otherwise is refused.
 * info[0..7] = width, height, display height, channels, precision, progressive flag, encoded format, bands decoded. */
int orc_decode_sample(const uint8_t *d, size_t size, PIXEL16 *const dst[4][3][4], const int pitch[4][3][4], const int dims[4][3][4][2], int32_t info[8])
{
	size_t pos = 0;
	int channel = 0, lv = -1, band = 0, bw = 0, bh = 0, bq = 1, bflags = 0, lw = 0, lh = 0, decoded = 0;
	size_t peak_base = 0; uint32_t peak_offset = 0; int peak_level = 0;
	memset(info, 0, 8 * sizeof(int32_t));
	info[5] = 1;
	while (pos + 4 <= size) {
		int tag = (int16_t)((d[pos] << 8) | d[pos + 1]);
		const int value = (d[pos + 2] << 8) | d[pos + 3];
		pos += 4;
		if (tag < 0) tag = -tag;
		if (tag & 0x4000) { pos += (size_t)((tag & 0x2000) ? (((uint32_t)(tag & 0xff) << 16) | (uint32_t)value) : (uint32_t)value) * 4; continue; }
		if (tag & 0x2000) continue;
		switch (tag) {
		case 2: pos += 4 * (size_t)value; break;                            /* CODEC_TAG_INDEX: the channel size table */
		case 62: channel = value; if (channel < 0 || channel > 3) return -3; break;          /* CODEC_TAG_CHANNEL */
		case 12: info[3] = value; break;                                    /* NUM_CHANNELS */
		case 20: info[0] = value; break; case 21: info[1] = value; break;   /* FRAME_WIDTH / FRAME_HEIGHT */
		case 85: info[2] = value; break;                                    /* FRAME_DISPLAY_HEIGHT */
		case 70: info[4] = value; break;                                    /* PRECISION */
		case 68: info[5] = value & 1; break;                                /* SAMPLE_FLAGS: bit 0 progressive */
		case 84: info[6] = value; break;                                    /* ENCODED_FORMAT */
		case 27: lw = value; break; case 28: lh = value; break;             /* LOWPASS_WIDTH / LOWPASS_HEIGHT */
		case 4:                                                             /* CODEC_TAG_MARKER */
			if (value == 0x0F0F) {                                          /* coefficient start: the raw lowpass band (decoder.c:23446-23470 -> DecodeSampleSubband(0)) */
				PIXEL16 *out = dst[channel][2][0];
				const size_t bytes = (size_t)lw * lh * 2;
				int r, x;
				if (pos + bytes > size) return -4;
				if (out) {
					if (lw != dims[channel][2][0][0] || lh != dims[channel][2][0][1]) return -5;
					for (r = 0; r < lh; r++) for (x = 0; x < lw; x++) { const uint8_t *p = d + pos + ((size_t)r * lw + x) * 2; out[(size_t)r * pitch[channel][2][0] + x] = (PIXEL16)(uint16_t)((p[0] << 8) | p[1]); }
				}
				pos += (bytes + 3) & ~(size_t)3;
				decoded++;
			}
			break;
		case 38: lv = value - 1; if (lv < 0 || lv > 2) return -6; break;    /* WAVELET_NUMBER */
		case 48: band = value; if (band < 1 || band > 3) return -7; bflags = 0; break;       /* BAND_NUMBER */
		case 72: bflags = value; break;                                     /* BAND_CODING_FLAGS: bits 0-3 code book, bit 4 difference coding (decoder.c:23970-23976) */
		case 75: peak_offset = (peak_offset & ~0xffffu) | (uint32_t)value; peak_base = pos; peak_level = 0; break;      /* PEAK_TABLE_OFFSET_L (:23978): base = the word behind this tuple */
		case 76: peak_offset = (peak_offset & 0xffffu) | ((uint32_t)value << 16); peak_level = 0; break;
		case 74: peak_level = value; break;                                 /* PEAK_LEVEL (:23991): base += offset */
		case 49: bw = value; break; case 50: bh = value; break;             /* BAND_WIDTH / BAND_HEIGHT */
		case 53: bq = value; break;                                         /* BAND_QUANTIZATION */
		case 55: {                                                          /* BAND_HEADER: the code words follow (decoder.c:23425) */
			PIXEL16 *out;
			long bits;
			static PIXEL16 *scratch = NULL; static size_t scratch_n = 0;
			const uint8_t *peaks = NULL; size_t peak_bytes = 0;
			int p;
			if (lv < 0) return -8;
			out = dst[channel][lv][band]; p = out ? pitch[channel][lv][band] : ((bw + 7) & ~7);
			if (out && (bw != dims[channel][lv][band][0] || bh != dims[channel][lv][band][1])) return -5;
			if (!out) { if ((size_t)bh * p > scratch_n) { free(scratch); scratch_n = (size_t)bh * p; scratch = (PIXEL16 *)malloc(scratch_n * sizeof(PIXEL16)); } out = scratch; }
			if (peak_level) { const size_t at = peak_base + peak_offset; if (at + 2 > size) return -9; peaks = d + at; peak_bytes = size - at; }
			bits = orc_decode_band_bits(d + pos, size - pos, bw, bh, p, bq, bflags & 0xf, peaks, peak_bytes, peak_level, (bflags >> 4) & 1, out);
			if (bits < 0) return -20 + (int)bits;
			pos += (((size_t)bits + 31) / 32) * 4;
			peak_level = 0;
			decoded++;
			break; }
		default: break;
		}
	}
	info[7] = decoded;
	return 0;
}


/* The same walk over the group sample of a two-frame GOP (Codec/decoder.c:11180 DecodeSampleGroup -> the same UpdateCodecState / DecodeSampleSubband per tag): six
 * wavelets per channel (WAVELET_NUMBER 1..6 -> 0..5: the two frame wavelets, the temporal wavelet, the spatial wavelet of the temporal highpass, the two of the temporal
 * lowpass), the lowpass band of the top wavelet (index 5) raw behind the coefficient marker, every other band behind its BAND_HEADER: run-length coded (BAND_ENCODING 3, code
 * set and difference coding from BAND_CODING_FLAGS, peak tables as for intra samples) or -- band 0 of wavelet 3, the lowpass band of the temporal highpass -- as signed
 * 16-bit big-endian words times the band's divisor (BAND_ENCODING 4 = BAND_ENCODING_16BIT, decoder.c:12790 DecodeBand16s), the walk going on at the next 32-bit word.
 * dst / pitch / dims[channel 0..2][wavelet 0..5][band 0..3]; info as orc_decode_sample (info[5]: the progressive flag as the reference defaults it for a group: 0 unless a
 * SAMPLE_FLAGS tag says otherwise, codec.c:263, decoder.c:13397; info[6]: frames in the group). */
int orc_decode_group(const uint8_t *d, size_t size, PIXEL16 *const dst[3][6][4], const int pitch[3][6][4], const int dims[3][6][4][2], int32_t info[8])
{
	size_t pos = 0;
	int channel = 0, wv = -1, band = 0, bw = 0, bh = 0, bq = 1, bflags = 0, benc = 3, bsub = 0, lw = 0, lh = 0, decoded = 0;
	size_t peak_base = 0; uint32_t peak_offset = 0; int peak_level = 0;
	memset(info, 0, 8 * sizeof(int32_t));
	while (pos + 4 <= size) {
		int tag = (int16_t)((d[pos] << 8) | d[pos + 1]);
		const int value = (d[pos + 2] << 8) | d[pos + 3];
		pos += 4;
		if (tag < 0) tag = -tag;
		if (tag & 0x4000) { pos += (size_t)((tag & 0x2000) ? (((uint32_t)(tag & 0xff) << 16) | (uint32_t)value) : (uint32_t)value) * 4; continue; }
		if (tag & 0x2000) continue;
		switch (tag) {
		case 2: pos += 4 * (size_t)value; break;                            /* CODEC_TAG_INDEX */
		case 62: channel = value; if (channel < 0 || channel > 2) return -3; break;
		case 12: info[3] = value; break;
		case 11: info[6] = value; break;                                    /* NUM_FRAMES */
		case 20: info[0] = value; break; case 21: info[1] = value; break;
		case 85: info[2] = value; break;
		case 70: info[4] = value; break;
		case 68: info[5] = value & 1; break;
		case 27: lw = value; break; case 28: lh = value; break;
		case 4:
			if (value == 0x0F0F) {
				PIXEL16 *out = dst[channel][5][0];
				const size_t bytes = (size_t)lw * lh * 2;
				int r, x;
				if (pos + bytes > size) return -4;
				if (out) {
					if (lw != dims[channel][5][0][0] || lh != dims[channel][5][0][1]) return -5;
					for (r = 0; r < lh; r++) for (x = 0; x < lw; x++) { const uint8_t *p = d + pos + ((size_t)r * lw + x) * 2; out[(size_t)r * pitch[channel][5][0] + x] = (PIXEL16)(uint16_t)((p[0] << 8) | p[1]); }
				}
				pos += (bytes + 3) & ~(size_t)3;
				decoded++;
			}
			break;
		case 38: wv = value - 1; if (wv < 0 || wv > 5) return -6; break;
		case 48: band = value; if (band < 0 || band > 3) return -7; bflags = 0; benc = 3; break;
		case 72: bflags = value; break;
		case 75: peak_offset = (peak_offset & ~0xffffu) | (uint32_t)value; peak_base = pos; peak_level = 0; break;
		case 76: peak_offset = (peak_offset & 0xffffu) | ((uint32_t)value << 16); peak_level = 0; break;
		case 74: peak_level = value; break;
		case 49: bw = value; break; case 50: bh = value; break;
		case 51: bsub = value; break;                                       /* BAND_SUBBAND */
		case 52: benc = value; break;                                       /* BAND_ENCODING */
		case 53: bq = value; break;
		case 55: {
			PIXEL16 *out;
			static PIXEL16 *scratch = NULL; static size_t scratch_n = 0;
			int p;
			if (wv < 0) return -8;
			if (bsub == 255) { if (wv != 2 || band != 1) return -10; break; }      /* the empty band of the temporal wavelet: a header and nothing behind it (decoder.c:11954) */
			out = dst[channel][wv][band]; p = out ? pitch[channel][wv][band] : ((bw + 7) & ~7);
			if (out && (bw != dims[channel][wv][band][0] || bh != dims[channel][wv][band][1])) return -5;
			if (!out) { if ((size_t)bh * p > scratch_n) { free(scratch); scratch_n = (size_t)bh * p; scratch = (PIXEL16 *)malloc(scratch_n * sizeof(PIXEL16)); } out = scratch; }
			if (benc == 4) {
				const size_t bytes = (size_t)bw * bh * 2;
				int r, x;
				if (pos + bytes > size) return -4;
				for (r = 0; r < bh; r++) for (x = 0; x < bw; x++) { const uint8_t *q = d + pos + ((size_t)r * bw + x) * 2; out[(size_t)r * p + x] = (PIXEL16)((int16_t)(uint16_t)((q[0] << 8) | q[1]) * bq); }
				pos += (bytes + 3) & ~(size_t)3;
				{   /* the encoder closes the raw band with the band end code word all the same (encoder.c:8219-8260): walked over as a band of no coefficients */
					const long e = orc_decode_band_bits(d + pos, size - pos, 0, 0, 0, 1, 1, NULL, 0, 0, 0, out);
					if (e < 0) return -30 + (int)e;
					pos += (((size_t)e + 31) / 32) * 4;
				}
			} else {
				const uint8_t *peaks = NULL; size_t peak_bytes = 0;
				long bits;
				if (peak_level) { const size_t at = peak_base + peak_offset; if (at + 2 > size) return -9; peaks = d + at; peak_bytes = size - at; }
				bits = orc_decode_band_bits(d + pos, size - pos, bw, bh, p, bq, bflags & 0xf, peaks, peak_bytes, peak_level, (bflags >> 4) & 1, out);
				if (bits < 0) return -20 + (int)bits;
				pos += (((size_t)bits + 31) / 32) * 4;
			}
			peak_level = 0;
			decoded++;
			break; }
		default: break;
		}
	}
	info[7] = decoded;
	return 0;
}
