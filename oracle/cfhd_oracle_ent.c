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
