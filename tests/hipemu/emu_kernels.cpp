// tests/hipemu/emu_kernels.cpp -- TEST INFRASTRUCTURE ONLY.
// Runs the product's HIP kernels (cineform-sdk_amd/csrc/cfhd_kernels.h, unmodified source) on the CPU through
// hip_emu.h so the `-m "not gpu"` suite can check their tiling / LDS / border logic against the oracle.
#include "hip_emu.h"
#define CFHD_ENT_FILL 64          // k_ent_layout: pieces of 16 words, so that the small test frames give holes of many pieces
dim3 threadIdx, blockIdx, blockDim, gridDim;
#include "cfhd_kernels.h"
#include <vector>

using namespace cfhd::dev;

static QuantParam make_q(int divisor, int mpq)
{
	QuantParam q; q.divisor = divisor; q.mid = 0; q.mult = 0;
	if (divisor > 1) {
		if (mpq >= 2 && mpq < 9) { q.mid = divisor / mpq; if (mpq == 2 && q.mid) q.mid--; }
		q.mult = ((1u << 16) / (unsigned)divisor) & 0xffffu;
	}
	return q;
}

extern "C" {

void emu_fwd_plane(const int16_t *in, int in_pitch, int width, int height, int prescale, const int *quant, int mpq,
                   int16_t *ll, int16_t *lh, int16_t *hl, int16_t *hh, int out_pitch)
{
	FwdPlaneJob job;
	job.in = in; job.in_pitch = in_pitch; job.width = width; job.height = height; job.prescale = prescale;
	job.out[0] = ll; job.out[1] = lh; job.out[2] = hl; job.out[3] = hh; job.out_pitch = out_pitch;
	job.xstride = 1; job.shift = 0; job.display_height = height; job.compand = 0;
	for (int b = 0; b < 4; b++) job.q[b] = make_q(quant[b], mpq);
	dim3 grid((width / 2 + TW - 1) / TW, (height / 2 + TH - 1) / TH, 1);
	hipemu::launch(grid, dim3(NTHREADS), [&] { k_fwd_plane(&job); });
}

// Level 1 of a 4:4:4(:4) format from interleaved 16-bit pixels: nch component planes, quant[c*4+b], out[c*4+b].
void emu_fwd_packed16(const uint16_t *in, int in_pitch_words, int width, int height, int display_height, int nch, int shift, const int *word_of_channel, int compand_channel,
                      const int *quant, int mpq, int16_t **out, int out_pitch)
{
	std::vector<FwdPlaneJob> jobs(nch);
	for (int c = 0; c < nch; c++) {
		FwdPlaneJob &job = jobs[c];
		job.in = (const int16_t *)(in + word_of_channel[c]); job.in_pitch = in_pitch_words; job.width = width; job.height = height; job.prescale = 0;
		job.xstride = nch; job.shift = shift; job.display_height = display_height; job.compand = c == compand_channel;
		for (int b = 0; b < 4; b++) { job.out[b] = out[c * 4 + b]; job.q[b] = make_q(quant[c * 4 + b], mpq); }
		job.out_pitch = out_pitch;
	}
	dim3 grid(((width / 2 + TW - 1) / TW) * nch, (height / 2 + TH - 1) / TH, 1);
	hipemu::launch(grid, dim3(NTHREADS), [&] { k_fwd_packed16(jobs.data(), nch); });
}

// The same level through k_fwd_packed16 (which = 0) or k_fwd_packed16_strip (which = 1): RG48 (wpp 3) / b64a (wpp 4, nch 4 or 3) with the
// word order EncodeBatch::fill_jobs uses.  The two kernels have to agree bit for bit.
void emu_fwd_packed16_shapes(int which, const uint16_t *in, int in_pitch_words, int width, int height, int display_height, int wpp, int nch, int shift, int compand_alpha,
                             const int *quant, int mpq, int16_t **out, int out_pitch)
{
	static const int rg48[4] = { 1, 0, 2, 3 }, b64a[4] = { 2, 1, 3, 0 };
	std::vector<FwdPlaneJob> jobs(nch);
	for (int c = 0; c < nch; c++) {
		FwdPlaneJob &job = jobs[c];
		memset(&job, 0, sizeof(job));
		job.in = (const int16_t *)(in + (wpp == 3 ? rg48 : b64a)[c]); job.in_pitch = in_pitch_words; job.width = width; job.height = height; job.prescale = 0;
		job.xstride = wpp; job.shift = shift; job.display_height = display_height; job.compand = compand_alpha && c == 3;
		for (int b = 0; b < 4; b++) { job.out[b] = out[c * 4 + b]; job.q[b] = make_q(quant[c * 4 + b], mpq); }
		job.out_pitch = out_pitch;
	}
	if (!which) {
		dim3 grid(((width / 2 + TW - 1) / TW) * nch, (height / 2 + TH - 1) / TH, 1);
		hipemu::launch(grid, dim3(NTHREADS), [&] { k_fwd_packed16(jobs.data(), nch); });
		return;
	}
	const int nseg = (width / 8 + PSTEP - 1) / PSTEP, nstrips = (height / 2 + PSR - 1) / PSR, waves = nseg * nstrips;
	const dim3 grid((waves + 3) / 4);
	if (wpp == 3) hipemu::launch(grid, dim3(NTHREADS), [&] { k_fwd_packed16_strip<3, 3>(jobs.data(), 1, nseg, nstrips); });
	else if (nch == 4) hipemu::launch(grid, dim3(NTHREADS), [&] { k_fwd_packed16_strip<4, 4>(jobs.data(), 1, nseg, nstrips); });
	else hipemu::launch(grid, dim3(NTHREADS), [&] { k_fwd_packed16_strip<4, 3>(jobs.data(), 1, nseg, nstrips); });
}

// Level 1 of a 4:2:2 frame from deep RGB pixels (RG48: wpp 3, r_word 0; b64a: wpp 4, r_word 1): the conversion happens in the loader of
// k_fwd_packed16 (layout 7), as EncodeBatch::fill_jobs sets it up.  quant[c*4+b], out[c*4+b]; out_pitch[c].
void emu_fwd_rgb16_to_yuv422(const uint16_t *in, int in_pitch_words, int wpp, int r_word, int width, int height, int display_height, int color_space,
                             const int *quant, int mpq, int16_t **out, const int *out_pitch)
{
	std::vector<FwdPlaneJob> jobs(3);
	for (int c = 0; c < 3; c++) {
		FwdPlaneJob &job = jobs[c];
		memset(&job, 0, sizeof(job));
		job.in = (const int16_t *)(in + r_word); job.in_pitch = in_pitch_words; job.width = c ? width / 2 : width; job.height = height; job.prescale = 0;
		job.xstride = wpp; job.shift = color_space; job.display_height = display_height; job.layout = 7; job.tail_from = c;
		for (int b = 0; b < 4; b++) { job.out[b] = out[c * 4 + b]; job.q[b] = make_q(quant[c * 4 + b], mpq); }
		job.out_pitch = out_pitch[c];
	}
	dim3 grid(((width / 2 + TW - 1) / TW) * 3, (height / 2 + TH - 1) / TH, 1);
	hipemu::launch(grid, dim3(NTHREADS), [&] { k_fwd_packed16(jobs.data(), 3); });
}

// Level 1 of a 4:2:2 frame from 8-bit pixels B, G, R(, A) (RG24 / BGRA: bottom row first, layout 8; BGRa: top row first, layout 9), as
// EncodeBatch::fill_jobs sets it up.
void emu_fwd_rgb8_to_yuv422(const uint8_t *in, int in_pitch, int bpp, int top_down, int width, int height, int display_height, int color_space,
                            const int *quant, int mpq, int16_t **out, const int *out_pitch)
{
	std::vector<FwdPlaneJob> jobs(3);
	for (int c = 0; c < 3; c++) {
		FwdPlaneJob &job = jobs[c];
		memset(&job, 0, sizeof(job));
		job.in = (const int16_t *)in; job.in_pitch = in_pitch; job.width = c ? width / 2 : width; job.height = height; job.prescale = 0;
		job.xstride = bpp; job.shift = color_space; job.display_height = display_height; job.layout = top_down ? 9 : 8; job.tail_from = c;
		for (int b = 0; b < 4; b++) { job.out[b] = out[c * 4 + b]; job.q[b] = make_q(quant[c * 4 + b], mpq); }
		job.out_pitch = out_pitch[c];
	}
	dim3 grid(((width / 2 + TW - 1) / TW) * 3, (height / 2 + TH - 1) / TH, 1);
	hipemu::launch(grid, dim3(NTHREADS), [&] { k_fwd_packed16(jobs.data(), 3); });
}

// Level 1 of a 4:2:2 frame from 16-bit words Y0 C1 Y1 C2 (YU64): the same kernel with per-channel first word, stride and width, as
// EncodeBatch::fill_jobs sets it up.  quant[c*4+b], out[c*4+b]; out_pitch[c].
void emu_fwd_yu64(const uint16_t *in, int in_pitch_words, int width, int height, int display_height, const int *quant, int mpq, int16_t **out, const int *out_pitch)
{
	std::vector<FwdPlaneJob> jobs(3);
	for (int c = 0; c < 3; c++) {
		FwdPlaneJob &job = jobs[c];
		job.in = (const int16_t *)(in + (c == 0 ? 0 : (c == 1 ? 1 : 3))); job.in_pitch = in_pitch_words; job.width = c ? width / 2 : width; job.height = height; job.prescale = 0;
		job.xstride = c ? 4 : 2; job.shift = 6; job.display_height = display_height; job.compand = 0;
		for (int b = 0; b < 4; b++) { job.out[b] = out[c * 4 + b]; job.q[b] = make_q(quant[c * 4 + b], mpq); }
		job.out_pitch = out_pitch[c];
	}
	dim3 grid(((width / 2 + TW - 1) / TW) * 3, (height / 2 + TH - 1) / TH, 1);
	hipemu::launch(grid, dim3(NTHREADS), [&] { k_fwd_packed16(jobs.data(), 3); });
}

// Level 1 of a 4:2:2 frame from v210 words: the loader picks the 10-bit fields (FwdPlaneJob::layout), as EncodeBatch::fill_jobs sets it up.
void emu_fwd_v210(const uint32_t *in, int in_pitch_bytes, int width, int height, int display_height, const int *quant, int mpq, int16_t **out, const int *out_pitch)
{
	std::vector<FwdPlaneJob> jobs(3);
	for (int c = 0; c < 3; c++) {
		FwdPlaneJob &job = jobs[c];
		job.in = (const int16_t *)in; job.in_pitch = in_pitch_bytes / 2; job.width = c ? width / 2 : width; job.height = height; job.prescale = 0;
		job.xstride = 3; job.shift = 6; job.display_height = display_height; job.compand = 0;
		job.layout = c + 1; job.tail_from = (width - width % 48) / 2;
		for (int b = 0; b < 4; b++) { job.out[b] = out[c * 4 + b]; job.q[b] = make_q(quant[c * 4 + b], mpq); }
		job.out_pitch = out_pitch[c];
	}
	dim3 grid(((width / 2 + TW - 1) / TW) * 3, (height / 2 + TH - 1) / TH, 1);
	hipemu::launch(grid, dim3(NTHREADS), [&] { k_fwd_packed16(jobs.data(), 3); });
}

// Level 1 of an RGB 4:4:4 frame from 8-bit B, G, R bytes, bottom row first (RG24): FwdPlaneJob::layout 4, as EncodeBatch::fill_jobs sets it up.
void emu_fwd_rg24(const uint8_t *in, int in_pitch_bytes, int width, int height, int display_height, const int *quant, int mpq, int16_t **out, int out_pitch)
{
	std::vector<FwdPlaneJob> jobs(3);
	for (int c = 0; c < 3; c++) {
		FwdPlaneJob &job = jobs[c];
		job.in = (const int16_t *)in; job.in_pitch = in_pitch_bytes; job.width = width; job.height = height; job.prescale = 0;
		job.xstride = 3; job.shift = 4; job.display_height = display_height; job.compand = 0;
		job.layout = 4; job.tail_from = c == 0 ? 1 : (c == 1 ? 2 : 0);
		for (int b = 0; b < 4; b++) { job.out[b] = out[c * 4 + b]; job.q[b] = make_q(quant[c * 4 + b], mpq); }
		job.out_pitch = out_pitch;
	}
	dim3 grid(((width / 2 + TW - 1) / TW) * 3, (height / 2 + TH - 1) / TH, 1);
	hipemu::launch(grid, dim3(NTHREADS), [&] { k_fwd_packed16(jobs.data(), 3); });
}

// Level 1 of an RGBA 4:4:4:4 frame from 8-bit B, G, R, A bytes (BGRA: bottom row first, layout 4; BGRa: top row first, layout 5); the alpha plane is curved.
void emu_fwd_rgba8(const uint8_t *in, int in_pitch_bytes, int top_down, int width, int height, int display_height, const int *quant, int mpq, int16_t **out, int out_pitch)
{
	std::vector<FwdPlaneJob> jobs(4);
	for (int c = 0; c < 4; c++) {
		FwdPlaneJob &job = jobs[c];
		memset(&job, 0, sizeof(job));
		job.in = (const int16_t *)in; job.in_pitch = in_pitch_bytes; job.width = width; job.height = height; job.prescale = 0;
		job.xstride = 4; job.shift = 4; job.display_height = display_height; job.compand = c == 3;
		job.layout = top_down ? 5 : 4; job.tail_from = c == 0 ? 1 : (c == 1 ? 2 : (c == 2 ? 0 : 3));
		for (int b = 0; b < 4; b++) { job.out[b] = out[c * 4 + b]; job.q[b] = make_q(quant[c * 4 + b], mpq); }
		job.out_pitch = out_pitch;
	}
	dim3 grid(((width / 2 + TW - 1) / TW) * 4, (height / 2 + TH - 1) / TH, 1);
	hipemu::launch(grid, dim3(NTHREADS), [&] { k_fwd_packed16(jobs.data(), 4); });
}

// Level 1 of an RGB 4:4:4 frame from 10-bit fields of one 32-bit word per pixel (r210 ...): FwdPlaneJob::layout 6.  shifts[c]: bit position of plane c.
void emu_fwd_rgb10(const uint32_t *in, int in_pitch_bytes, int width, int height, int display_height, int big_endian, const int *shifts, const int *quant, int mpq,
                   int16_t **out, int out_pitch)
{
	std::vector<FwdPlaneJob> jobs(3);
	for (int c = 0; c < 3; c++) {
		FwdPlaneJob &job = jobs[c];
		job.in = (const int16_t *)in; job.in_pitch = in_pitch_bytes / 4; job.width = width; job.height = height; job.prescale = 0;
		job.xstride = big_endian; job.shift = 0; job.display_height = display_height; job.compand = 0;
		job.layout = 6; job.tail_from = shifts[c];
		for (int b = 0; b < 4; b++) { job.out[b] = out[c * 4 + b]; job.q[b] = make_q(quant[c * 4 + b], mpq); }
		job.out_pitch = out_pitch;
	}
	dim3 grid(((width / 2 + TW - 1) / TW) * 3, (height / 2 + TH - 1) / TH, 1);
	hipemu::launch(grid, dim3(NTHREADS), [&] { k_fwd_packed16(jobs.data(), 3); });
}

// Last level of a 4:4:4(:4) format to interleaved 16-bit pixels: bands[c*4+b].
static int g_inv_packed16_strip = 0;
void emu_inv_packed16(int16_t **bands, int band_pitch, int w, int h, int display_height, int nch, int precision, const int *word_of_channel,
                      uint16_t *out, int out_pitch_words, int alpha_channel)
{
	std::vector<InvPlaneJob> jobs(nch);
	for (int c = 0; c < nch; c++) {
		InvPlaneJob &job = jobs[c];
		for (int b = 0; b < 4; b++) job.band[b] = bands[c * 4 + b];
		job.band_pitch = band_pitch; job.width = w; job.height = h; job.descale = 0;
		job.out = (int16_t *)(out + word_of_channel[c]); job.out_pitch = out_pitch_words; job.xstride = nch; job.precision = precision; job.display_height = display_height;
		job.alpha = c == alpha_channel;
	}
	if (g_inv_packed16_strip) {                           // (word_of_channel has to be the RG48 / b64a order the kernel is built for)
		const int nseg = (w / 4 + PSTEP - 1) / PSTEP, nstrips = (h + QSR - 1) / QSR, waves = nseg * nstrips;
		if (nch == 4) hipemu::launch(dim3((waves + 3) / 4), dim3(NTHREADS), [&] { k_inv_packed16_strip<4>(jobs.data(), 1, nseg, nstrips); });
		else hipemu::launch(dim3((waves + 3) / 4), dim3(NTHREADS), [&] { k_inv_packed16_strip<3>(jobs.data(), 1, nseg, nstrips); });
		return;
	}
	dim3 grid((w + ITW - 1) / ITW, (h + ITH - 1) / ITH, 1);
	hipemu::launch(grid, dim3(NTHREADS), [&] { k_inv_packed16(jobs.data(), nch, nch, 0u); });
}
void emu_inv_packed16_use_strip(int on) { g_inv_packed16_strip = on; }

// YU64 rows -> RG24 (k_yu64_to_rgb24, the second step of decoding 4:2:2 samples to RG24), as DecodeBatch launches it.
void emu_yu64_to_rgb24(const uint16_t *yu64, int in_pitch_words, int width, int rows, int matrix, uint32_t seed, uint8_t *out, int out_pitch)
{
	const int pairs = width / 2;
	hipemu::launch(dim3((pairs + NTHREADS - 1) / NTHREADS, rows, 1), dim3(NTHREADS), [&] { k_yu64_to_rgb24(yu64, in_pitch_words, 0, out, out_pitch, 0, pairs, rows, matrix, seed); });
}

// The last level of an RGB 4:4:4 sample to b64a: three planes into words 1..3 of four-word pixels, word 0 = 0xfff0 (InvPlaneJob::alpha_const), as
// DecodeBatch::prepare sets it up.
void emu_inv_b64a_of_444(int16_t **bands, int band_pitch, int w, int h, int display_height, uint16_t *out, int out_pitch_words)
{
	std::vector<InvPlaneJob> jobs(3);
	const int word_of_channel[3] = { 2, 1, 3 };            // planes G, R, B -> words A R G B
	for (int c = 0; c < 3; c++) {
		InvPlaneJob &job = jobs[c];
		memset(&job, 0, sizeof(job));
		for (int b = 0; b < 4; b++) job.band[b] = bands[c * 4 + b];
		job.band_pitch = band_pitch; job.width = w; job.height = h; job.descale = 0;
		job.out = (int16_t *)(out + word_of_channel[c]); job.out_pitch = out_pitch_words; job.xstride = 4; job.precision = 12; job.display_height = display_height;
		job.alpha_const = 0xfff0;
	}
	dim3 grid((w + ITW - 1) / ITW, (h + ITH - 1) / ITH, 1);
	hipemu::launch(grid, dim3(NTHREADS), [&] { k_inv_packed16(jobs.data(), 3, 4, 0u); });
}

// The last level of a 4:2:2 sample to YU64 (words Y0 C1 Y1 C2): k_inv_packed16 with per-plane widths and word strides, as DecodeBatch::prepare sets it up.
// bands[c*4+b], band_pitch[c]; luma band w x h.
void emu_inv_yu64(int16_t **bands, const int *band_pitch, int w, int h, int display_height, int precision, uint16_t *out, int out_pitch_words)
{
	std::vector<InvPlaneJob> jobs(3);
	for (int c = 0; c < 3; c++) {
		InvPlaneJob &job = jobs[c];
		memset(&job, 0, sizeof(job));
		for (int b = 0; b < 4; b++) job.band[b] = bands[c * 4 + b];
		job.band_pitch = band_pitch[c]; job.width = c ? w / 2 : w; job.height = h; job.descale = 0;
		job.out = (int16_t *)(out + (c == 0 ? 0 : (c == 1 ? 1 : 3))); job.out_pitch = out_pitch_words; job.xstride = c ? 4 : 2; job.precision = precision; job.display_height = display_height;
	}
	dim3 grid((w + ITW - 1) / ITW, (h + ITH - 1) / ITH, 1);
	hipemu::launch(grid, dim3(NTHREADS), [&] { k_inv_packed16(jobs.data(), 3, 2, 0u); });
}

// The last level of an RGB 4:4:4 sample to 8-bit pixels B, G, R(, A): k_inv_packed16 in its byte mode, as DecodeBatch::prepare sets it up.
void emu_inv_rgb8(int16_t **bands, int band_pitch, int w, int h, int display_height, int precision, int bytes_per_pixel, int bottom_up, uint32_t seed,
                  uint8_t *out, int out_pitch_bytes)
{
	std::vector<InvPlaneJob> jobs(3);
	for (int c = 0; c < 3; c++) {
		InvPlaneJob &job = jobs[c];
		memset(&job, 0, sizeof(job));
		for (int b = 0; b < 4; b++) job.band[b] = bands[c * 4 + b];
		job.band_pitch = band_pitch; job.width = w; job.height = h; job.descale = 0;
		job.out = (int16_t *)(out + (c == 0 ? 1 : (c == 1 ? 2 : 0))); job.out_pitch = out_pitch_bytes; job.xstride = bytes_per_pixel; job.precision = precision; job.display_height = display_height;
		job.bytes8 = 1; job.bottom_up = bottom_up; job.dither_seed = seed;
	}
	dim3 grid((w + ITW - 1) / ITW, (h + ITH - 1) / ITH, 1);
	hipemu::launch(grid, dim3(NTHREADS), [&] { k_inv_packed16(jobs.data(), 3, bytes_per_pixel, 0x1234u); });
}

// The last level of an RGBA 4:4:4:4 sample to BGRA / BGRa: the byte mode without dither, alpha from the fourth plane (InvPlaneJob::bytes8 == 2).
void emu_inv_rgba8(int16_t **bands, int band_pitch, int w, int h, int display_height, int precision, int bottom_up, uint8_t *out, int out_pitch_bytes)
{
	std::vector<InvPlaneJob> jobs(4);
	for (int c = 0; c < 4; c++) {
		InvPlaneJob &job = jobs[c];
		memset(&job, 0, sizeof(job));
		for (int b = 0; b < 4; b++) job.band[b] = bands[c * 4 + b];
		job.band_pitch = band_pitch; job.width = w; job.height = h; job.descale = 0;
		job.out = (int16_t *)(out + (c == 0 ? 1 : (c == 1 ? 2 : (c == 2 ? 0 : 3)))); job.out_pitch = out_pitch_bytes; job.xstride = 4; job.precision = precision; job.display_height = display_height;
		job.bytes8 = 2; job.bottom_up = bottom_up; job.alpha = c == 3;
	}
	dim3 grid((w + ITW - 1) / ITW, (h + ITH - 1) / ITH, 1);
	hipemu::launch(grid, dim3(NTHREADS), [&] { k_inv_packed16(jobs.data(), 4, 4, 0x1234u); });
}

// The last level of an RGB 4:4:4 sample to 10-bit RGB words (r210 / DPX0 / AB10 / AR10): k_inv_rgb10, as DecodeBatch::prepare sets it up.
void emu_inv_rgb10(int16_t **bands, int band_pitch, int w, int h, int display_height, int shift_r, int shift_g, int shift_b, int big_endian, uint32_t *out, int out_pitch_words)
{
	std::vector<InvPlaneJob> jobs(3);
	const int shifts[3] = { shift_g, shift_r, shift_b };      // planes G, R, B
	for (int c = 0; c < 3; c++) {
		InvPlaneJob &job = jobs[c];
		memset(&job, 0, sizeof(job));
		for (int b = 0; b < 4; b++) job.band[b] = bands[c * 4 + b];
		job.band_pitch = band_pitch; job.width = w; job.height = h; job.descale = 0;
		job.out = (int16_t *)out; job.out_pitch = out_pitch_words; job.xstride = 3; job.precision = 12; job.display_height = display_height;
		job.bit_shift = shifts[c]; job.big_endian = big_endian;
	}
	dim3 grid((w + ITW - 1) / ITW, (h + ITH - 1) / ITH, 1);
	hipemu::launch(grid, dim3(NTHREADS), [&] { k_inv_rgb10(jobs.data()); });
}

void emu_fwd_frame_yuv422(const uint8_t *in, int in_pitch, int width, int height, int display_height, int uyvy, int shift,
                          const int *quant /*[3][4]*/, int mpq, int16_t **out /*[3][4]*/, const int *out_pitch)
{
	FwdFrameJob job;
	job.in = in; job.in_pitch = in_pitch; job.width = width; job.height = height; job.display_height = display_height;
	job.uyvy = uyvy; job.shift = shift;
	for (int c = 0; c < 3; c++) {
		job.out_pitch[c] = out_pitch[c];
		for (int b = 0; b < 4; b++) { job.out[c][b] = out[c * 4 + b]; job.q[c][b] = make_q(quant[c * 4 + b], mpq); }
		// the difference-coded band is quantized inside the horizontal filter with midpoint = divisor / prequant (no decrement)
		if (quant[c * 4 + 2] > 1 && mpq >= 2 && mpq < 9) job.q[c][2].mid = quant[c * 4 + 2] / mpq;
	}
	dim3 grid((width / 2 + FTW - 1) / FTW, (height / 2 + FRW - 1) / FRW, 1);
	hipemu::launch(grid, dim3(NTHREADS), [&] { k_fwd_frame_yuv422(&job); });
}

void emu_unpack_byr4(const uint16_t *in, int in_pitch_words, int width, int height, int display_height, const uint16_t *curve, int order, int precision,
                     int16_t **out /*[4]*/, int out_pitch)
{
	BayerJob job;
	memset(&job, 0, sizeof(job));
	job.in = in; job.in_pitch = in_pitch_words; job.width = width; job.height = height; job.display_height = display_height;
	job.curve = curve; job.order = order; job.precision = precision; job.out_pitch = out_pitch;
	for (int c = 0; c < 4; c++) job.out[c] = out[c];
	hipemu::launch(dim3((width / 2 + NTHREADS - 1) / NTHREADS, height, 1), dim3(NTHREADS), [&] { k_unpack_byr4(&job); });
}

// Bayer level 1 without the planes: k_fwd_packed16 computing every component plane in its loader (FwdPlaneJob::layout 10 BYR4 / 11 BYR5), as
// EncodeBatch::fill_jobs sets it up.  width / height: the component planes (coded height); quant[c*4+b], out[c*4+b].
void emu_fwd_bayer(const void *in, int in_pitch_words, int width, int height, int display_height, int packed12, const uint16_t *curve, int order,
                   const int *quant, int mpq, int16_t **out, int out_pitch)
{
	std::vector<FwdPlaneJob> jobs(4);
	for (int c = 0; c < 4; c++) {
		FwdPlaneJob &job = jobs[c];
		memset(&job, 0, sizeof(job));
		job.in = (const int16_t *)in; job.in_pitch = in_pitch_words; job.width = width; job.height = height; job.prescale = 0;
		job.layout = packed12 ? 11 : 10; job.tail_from = c; job.xstride = order; job.shift = 12; job.display_height = display_height; job.curve = curve;
		for (int b = 0; b < 4; b++) { job.out[b] = out[c * 4 + b]; job.q[b] = make_q(quant[c * 4 + b], mpq); }
		job.out_pitch = out_pitch;
	}
	dim3 grid(((width / 2 + TW - 1) / TW) * 4, (height / 2 + TH - 1) / TH, 1);
	hipemu::launch(grid, dim3(NTHREADS), [&] { k_fwd_packed16(jobs.data(), 4); });
}

// BYR5 input: the same kernel reading the packed 12-bit rows (BayerJob::packed12), as EncodeBatch::fill_jobs sets it up.
void emu_unpack_byr5(const uint8_t *in, int width, int height, int display_height, int order, int16_t **out /*[4]*/, int out_pitch)
{
	BayerJob job;
	memset(&job, 0, sizeof(job));
	job.in = (const uint16_t *)in; job.in_pitch = width * 3; job.width = width; job.height = height; job.display_height = display_height;
	job.order = order; job.precision = 12; job.out_pitch = out_pitch; job.packed12 = 1;
	for (int c = 0; c < 4; c++) job.out[c] = out[c];
	hipemu::launch(dim3((width / 2 + NTHREADS - 1) / NTHREADS, height, 1), dim3(NTHREADS), [&] { k_unpack_byr4(&job); });
}

void emu_fwd_yuv422(const uint8_t *in, int in_pitch, int width, int height, int display_height, int uyvy, int shift,
                    const int *quant /*[3][4]*/, int mpq, int16_t **out /*[3][4]*/, const int *out_pitch)
{
	FwdYuvJob job;
	job.in = in; job.in_pitch = in_pitch; job.width = width; job.height = height; job.display_height = display_height;
	job.uyvy = uyvy; job.shift = shift;
	for (int c = 0; c < 3; c++) { job.out_pitch[c] = out_pitch[c]; for (int b = 0; b < 4; b++) { job.out[c][b] = out[c * 4 + b]; job.q[c][b] = make_q(quant[c * 4 + b], mpq); } }
	dim3 grid((width / 2 + TW - 1) / TW, (height / 2 + TH - 1) / TH, 1);
	hipemu::launch(grid, dim3(NTHREADS), [&] { k_fwd_yuv422(&job); });
}

int emu_fwd_yuv422_strip(const uint8_t *in, int in_pitch, int width, int height, int display_height, int uyvy, int shift,
                         const int *quant /*[3][4]*/, int mpq, int16_t **out /*[3][4]*/, const int *out_pitch)
{
	if (width % 32) return -1;
	FwdYuvJob job;
	job.in = in; job.in_pitch = in_pitch; job.width = width; job.height = height; job.display_height = display_height;
	job.uyvy = uyvy; job.shift = shift;
	for (int c = 0; c < 3; c++) { job.out_pitch[c] = out_pitch[c]; for (int b = 0; b < 4; b++) { job.out[c][b] = out[c * 4 + b]; job.q[c][b] = make_q(quant[c * 4 + b], mpq); } }
	hipemu::launch(dim3((width / 16 + SSEG - 1) / SSEG, (height / 2 + SRF - 1) / SRF, 1), dim3(NTHREADS), [&] { k_fwd_yuv422_strip(&job); });
	return 0;
}

void emu_inv_plane_ex(const int16_t *ll, const int16_t *lh, const int16_t *hl, const int16_t *hh, int band_pitch, int w, int h, int descale,
                      int16_t *out, int out_pitch, int ll_bottom_row_high);
void emu_inv_plane(const int16_t *ll, const int16_t *lh, const int16_t *hl, const int16_t *hh, int band_pitch, int w, int h, int descale,
                   int16_t *out, int out_pitch)
{
	emu_inv_plane_ex(ll, lh, hl, hh, band_pitch, w, h, descale, out, out_pitch, 0);
}
// ll_bottom_row_high: the wavelets of a two-frame group that the reference sends through InvertSpatialQuantOverflowProtected16s (InvPlaneJob::ll_bottom_row_high)
void emu_inv_plane_ex(const int16_t *ll, const int16_t *lh, const int16_t *hl, const int16_t *hh, int band_pitch, int w, int h, int descale,
                      int16_t *out, int out_pitch, int ll_bottom_row_high)
{
	InvPlaneJob job = {};
	job.ll_bottom_row_high = ll_bottom_row_high;
	job.band[0] = ll; job.band[1] = lh; job.band[2] = hl; job.band[3] = hh; job.band_pitch = band_pitch;
	job.width = w; job.height = h; job.descale = descale; job.out = out; job.out_pitch = out_pitch;
	job.xstride = 1; job.precision = 0; job.display_height = 2 * h;
	dim3 grid((w + ITW - 1) / ITW, (h + ITH - 1) / ITH, 1);
	hipemu::launch(grid, dim3(NTHREADS), [&] { k_inv_plane(&job); });
}

void emu_inv_yuv422(int16_t **bands /*[3][4]*/, const int *band_pitch, int w, int h, int display_height, int uyvy, int shift,
                    unsigned dither_seed, uint8_t *out, int out_pitch)
{
	InvYuvJob job;
	for (int c = 0; c < 3; c++) { job.band_pitch[c] = band_pitch[c]; for (int b = 0; b < 4; b++) job.band[c][b] = bands[c * 4 + b]; }
	job.width = w; job.height = h; job.display_height = display_height; job.uyvy = uyvy; job.shift = shift;
	job.dither_seed = dither_seed; job.out = out; job.out_pitch = out_pitch;
	dim3 grid((w + ITW - 1) / ITW, (h + ITH - 1) / ITH, 1);
	hipemu::launch(grid, dim3(NTHREADS), [&] { k_inv_yuv422(&job, 0u); });
}

// interlaced frames: the inverse frame transform (any width)
void emu_inv_frame_yuv422(int16_t **bands /*[3][4]*/, const int *band_pitch, int w, int h, int display_height, int uyvy, int shift,
                          unsigned dither_seed, uint8_t *out, int out_pitch)
{
	InvYuvJob job;
	for (int c = 0; c < 3; c++) { job.band_pitch[c] = band_pitch[c]; for (int b = 0; b < 4; b++) job.band[c][b] = bands[c * 4 + b]; }
	job.width = w; job.height = h; job.display_height = display_height; job.uyvy = uyvy; job.shift = shift;
	job.dither_seed = dither_seed; job.out = out; job.out_pitch = out_pitch;
	hipemu::launch(dim3((w / 2 + NTHREADS - 1) / NTHREADS, h, 1), dim3(NTHREADS), [&] { k_inv_frame_yuv422(&job, 0u); });
}

// the same with four band columns per thread (luma band width a multiple of 4)
int emu_inv_frame_yuv422_quad(int16_t **bands /*[3][4]*/, const int *band_pitch, int w, int h, int display_height, int uyvy, int shift,
                              unsigned dither_seed, uint8_t *out, int out_pitch)
{
	if (w % 4) return -1;
	InvYuvJob job;
	for (int c = 0; c < 3; c++) { job.band_pitch[c] = band_pitch[c]; for (int b = 0; b < 4; b++) job.band[c][b] = bands[c * 4 + b]; }
	job.width = w; job.height = h; job.display_height = display_height; job.uyvy = uyvy; job.shift = shift;
	job.dither_seed = dither_seed; job.out = out; job.out_pitch = out_pitch;
	hipemu::launch(dim3((w / 4 + NTHREADS - 1) / NTHREADS, h, 1), dim3(NTHREADS), [&] { k_inv_frame_yuv422_quad(&job, 0u); });
	return 0;
}

// the register-strip variant of the same level (luma band width a multiple of 16; segments of 124 blocks of 8 columns)
int emu_inv_yuv422_strip(int16_t **bands /*[3][4]*/, const int *band_pitch, int w, int h, int display_height, int uyvy, int shift,
                         unsigned dither_seed, uint8_t *out, int out_pitch)
{
	if (w % 16) return -1;
	InvYuvJob job;
	for (int c = 0; c < 3; c++) { job.band_pitch[c] = band_pitch[c]; for (int b = 0; b < 4; b++) job.band[c][b] = bands[c * 4 + b]; }
	job.width = w; job.height = h; job.display_height = display_height; job.uyvy = uyvy; job.shift = shift;
	job.dither_seed = dither_seed; job.out = out; job.out_pitch = out_pitch;
	hipemu::launch(dim3((w / SBLK + SSEG - 1) / SSEG, (h + SR - 1) / SR, 1), dim3(NTHREADS), [&] { k_inv_yuv422_strip(&job, 0u); });
	return 0;
}

// ---- round 4 ----
// the interlaced last level in the register-strip organisation (luma band width a multiple of 16); lists: LH / HH of every channel given as block lists
// (masks: the frame's chunk masks, mask_base[c * 4 + b] as cfhd_core.h dec_block_list_layout numbers them), HL dense
int emu_inv_frame_yuv422_strip(int16_t **bands /*[3][4]*/, const int *band_pitch, int w, int h, int display_height, int uyvy, int shift,
                               unsigned dither_seed, uint8_t *out, int out_pitch, const unsigned long long *masks, const int *mask_base)
{
	if (w % 16) return -1;
	InvYuvJob job;
	memset(&job, 0, sizeof(job));
	for (int c = 0; c < 3; c++) { job.band_pitch[c] = band_pitch[c]; for (int b = 0; b < 4; b++) job.band[c][b] = bands[c * 4 + b]; }
	job.width = w; job.height = h; job.display_height = display_height; job.uyvy = uyvy; job.shift = shift;
	job.dither_seed = dither_seed; job.out = out; job.out_pitch = out_pitch;
	job.masks = masks;
	if (masks) for (int c = 0; c < 3; c++) for (int b = 0; b < 4; b++) job.mask_base[c][b] = mask_base[c * 4 + b];
	const dim3 grid((w / SBLK + SSEG - 1) / SSEG, (h + SRI - 1) / SRI, 1);
	if (masks) hipemu::launch(grid, dim3(NTHREADS), [&] { k_inv_frame_yuv422_strip_blocks(&job, 0u); });
	else hipemu::launch(grid, dim3(NTHREADS), [&] { k_inv_frame_yuv422_strip(&job, 0u); });
	return 0;
}

// the progressive last level from block lists (all three highpass bands of every channel)
int emu_inv_yuv422_strip_blocks(int16_t **bands /*[3][4]*/, const int *band_pitch, int w, int h, int display_height, int uyvy, int shift,
                                unsigned dither_seed, uint8_t *out, int out_pitch, const unsigned long long *masks, const int *mask_base)
{
	if (w % 16) return -1;
	InvYuvJob job;
	memset(&job, 0, sizeof(job));
	for (int c = 0; c < 3; c++) { job.band_pitch[c] = band_pitch[c]; for (int b = 0; b < 4; b++) job.band[c][b] = bands[c * 4 + b]; }
	job.width = w; job.height = h; job.display_height = display_height; job.uyvy = uyvy; job.shift = shift;
	job.dither_seed = dither_seed; job.out = out; job.out_pitch = out_pitch;
	job.masks = masks;
	for (int c = 0; c < 3; c++) for (int b = 0; b < 4; b++) job.mask_base[c][b] = mask_base[c * 4 + b];
	hipemu::launch(dim3((w / SBLK + SSEG - 1) / SSEG, (h + SR - 1) / SR, 1), dim3(NTHREADS), [&] { k_inv_yuv422_strip_blocks(&job, 0u); });
	return 0;
}

// the interlaced level 1 in the register-strip organisation (width a multiple of 32)
int emu_fwd_frame_yuv422_strip(const uint8_t *in, int in_pitch, int width, int height, int display_height, int uyvy, int shift,
                               const int *quant /*[3][4]*/, int mpq, int16_t **out /*[3][4]*/, const int *out_pitch)
{
	if (width % 32) return -1;
	FwdFrameJob job;
	memset(&job, 0, sizeof(job));
	job.in = in; job.in_pitch = in_pitch; job.width = width; job.height = height; job.display_height = display_height;
	job.uyvy = uyvy; job.shift = shift;
	for (int c = 0; c < 3; c++) {
		job.out_pitch[c] = out_pitch[c];
		for (int b = 0; b < 4; b++) { job.out[c][b] = out[c * 4 + b]; job.q[c][b] = make_q(quant[c * 4 + b], mpq); }
		if (quant[c * 4 + 2] > 1 && mpq >= 2 && mpq < 9) job.q[c][2].mid = quant[c * 4 + 2] / mpq;
	}
	hipemu::launch(dim3((width / 16 + SSEG - 1) / SSEG, (height / 2 + SRI - 1) / SRI, 1), dim3(NTHREADS), [&] { k_fwd_frame_yuv422_strip(&job); });
	return 0;
}

// Bayer level 1 straight from the BYR4 mosaic, all four component planes from one pass (k_fwd_bayer_strip); width / height: the component planes (coded height)
int emu_fwd_bayer_strip(const uint16_t *in, int in_pitch_words, int width, int height, int display_height, const uint16_t *curve, int order,
                        const int *quant, int mpq, int16_t **out, int out_pitch)
{
	if (width % 8) return -1;
	std::vector<FwdPlaneJob> jobs(4);
	for (int c = 0; c < 4; c++) {
		FwdPlaneJob &job = jobs[c];
		memset(&job, 0, sizeof(job));
		job.width = width; job.height = height; job.prescale = 0; job.display_height = height;
		for (int b = 0; b < 4; b++) { job.out[b] = out[c * 4 + b]; job.q[b] = make_q(quant[c * 4 + b], mpq); }
		job.out_pitch = out_pitch;
	}
	BayerJob bj;
	memset(&bj, 0, sizeof(bj));
	bj.in = in; bj.in_pitch = in_pitch_words; bj.width = width; bj.height = height; bj.display_height = display_height; bj.curve = curve; bj.order = order; bj.precision = 12;
	const int nseg = (width / 8 + PSTEP - 1) / PSTEP, nstrips = (height / 2 + PSR - 1) / PSR, waves = nseg * nstrips;
	hipemu::launch(dim3((waves + 3) / 4), dim3(NTHREADS), [&] { k_fwd_bayer_strip(jobs.data(), &bj, 1, nseg, nstrips); });
	return 0;
}

void emu_half_packed16(int16_t **ll, int nch, int pitch, int width, int rows, const int *word, int shift, int alpha, uint16_t *out, int out_pitch_bytes)
{
	HalfPackedJob job;
	for (int c = 0; c < 4; c++) { job.ll[c] = c < nch ? ll[c] : nullptr; job.word[c] = c < nch ? word[c] : 0; }
	job.pitch = pitch; job.width = width; job.rows = rows; job.nch = nch; job.shift = shift; job.alpha = alpha; job.out = out; job.out_pitch = out_pitch_bytes;
	hipemu::launch(dim3((width / 8 + NTHREADS - 1) / NTHREADS, rows, 1), dim3(NTHREADS), [&] { k_half_packed16(&job); });
}

void emu_half_yuv422(int16_t **ll /*[3]: Y, V, U*/, const int *pitch, int width, int rows, int uyvy, uint8_t *out, int out_pitch)
{
	HalfYuvJob job;
	for (int c = 0; c < 3; c++) { job.ll[c] = ll[c]; job.pitch[c] = pitch[c]; }
	job.width = width; job.rows = rows; job.uyvy = uyvy; job.out = out; job.out_pitch = out_pitch;
	hipemu::launch(dim3((width / 8 + NTHREADS - 1) / NTHREADS, rows, 1), dim3(NTHREADS), [&] { k_half_yuv422(&job); });
}

// nplanes planes of the same geometry (plane k: in[k] -> out[4k..4k+3]) through one launch of k_fwd_plane_strip
int emu_fwd_plane_strip(int16_t **in, int nplanes, int in_pitch, int width, int height, int prescale, const int *quant, int mpq, int16_t **out, int out_pitch)
{
	if (width % (2 * SBLK)) return -1;
	std::vector<FwdPlaneJob> jobs(nplanes);
	for (int k = 0; k < nplanes; k++) {
		FwdPlaneJob &job = jobs[k];
		job.in = in[k]; job.in_pitch = in_pitch; job.width = width; job.height = height; job.prescale = prescale;
		for (int b = 0; b < 4; b++) { job.out[b] = out[4 * k + b]; job.q[b] = make_q(quant[b], mpq); }
		job.out_pitch = out_pitch; job.xstride = 1; job.shift = 0; job.display_height = height; job.compand = 0;
	}
	int glog = 0; while ((1 << glog) < width / (2 * SBLK)) glog++;
	const int nblk = width / (2 * SBLK), nseg = nblk > 64 ? (nblk + PLSTEP - 1) / PLSTEP : 1;      // (as cfhd_device.hip for_channel_runs)
	if (nseg > 1) glog = 6;
	const int nstrips = (height / 2 + SRP - 1) / SRP, per_wave = nseg > 1 ? 1 : 64 >> glog, waves = ((nplanes + per_wave - 1) / per_wave) * nstrips * nseg;
	hipemu::launch(dim3((waves + 3) / 4), dim3(NTHREADS), [&] { k_fwd_plane_strip(jobs.data(), nplanes, 1, 0, 1, glog, nstrips, width, height, nseg); });
	return 0;
}

// nplanes independent planes of the same geometry (plane k: bands[4k..4k+3] -> out[k]) through one launch of the strip kernel, as the
// product batches the channel planes of many frames: the wave packing (64 >> glog planes per wave) is part of what is tested.
int emu_inv_plane_strip(int16_t **bands, int nplanes, int band_pitch, int w, int h, int descale, int16_t **out, int out_pitch)
{
	if (w % SBLK) return -1;
	std::vector<InvPlaneJob> jobs(nplanes);
	for (int k = 0; k < nplanes; k++) {
		InvPlaneJob &job = jobs[k];
		for (int b = 0; b < 4; b++) job.band[b] = bands[4 * k + b];
		job.band_pitch = band_pitch; job.width = w; job.height = h; job.descale = descale; job.out = out[k]; job.out_pitch = out_pitch;
		job.xstride = 1; job.precision = 0; job.display_height = 2 * h; job.alpha = 0;
	}
	int glog = 0; while ((1 << glog) < w / SBLK) glog++;
	const int nblk = w / SBLK, nseg = nblk > 64 ? (nblk + PLSTEP - 1) / PLSTEP : 1;
	if (nseg > 1) glog = 6;
	const int nstrips = (h + SRP - 1) / SRP, per_wave = nseg > 1 ? 1 : 64 >> glog, waves = ((nplanes + per_wave - 1) / per_wave) * nstrips * nseg;
	hipemu::launch(dim3((waves + 3) / 4), dim3(NTHREADS), [&] { k_inv_plane_strip(jobs.data(), nplanes, 1, 0, 1, glog, nstrips, w, h, nseg); });
	return 0;
}

} // extern "C"

// ------------------------------------------------------------------------------------------------------------
// GPU entropy stage under emulation: same tables / job construction as the device driver (cfhd_entropy_jobs.h).
// Needs the product's host objects for entropy_tables() / build_sample_template(): tests link them in.
// ------------------------------------------------------------------------------------------------------------
#include "cfhd_entropy_jobs.h"

// interlaced != 0: the plan of a field-coded frame (code set 18 for the difference band, coded with peaks: k_ent_peaks fills the table behind the band); returns -100
// when a band has more peaks than the stage's positions hold.
extern "C" long emu_entropy_encode2(int width, int height, int pixel_kind, int quality, unsigned frame_number, int16_t *coeffs /* product pyramid layout */,
                                    const uint8_t *meta, size_t meta_size, uint8_t *out, size_t cap, int interlaced)
{
	using namespace cfhd;
	FramePlan plan;
	if (!build_frame_plan(&plan, width, height, pixel_kind, ENC_YUV422)) return -1;
	plan.interlaced = interlaced != 0;
	QuantState st = {0, -1, 0};
	derive_quantization(&plan, quality, !interlaced, 0.0f, &st);
	SampleHeaderInfo hdr = { frame_number, pixel_kind == PIX_2VUY ? 1 : 2, 2, quality, !interlaced, meta, meta_size, nullptr, 0 };
	SampleTemplate t;
	build_sample_template(plan, hdr, &t);
	EntHostJobs jobs;
	if (!ent_build_band_jobs(plan, t, 1, coeffs, plan.coeff_elems, &jobs)) return -2;
	std::vector<uint8_t> block(kEntTmplStride, 0);
	if (!ent_fill_frame_block(plan, t, 0, jobs, coeffs, block.data())) return -3;
	uint32_t size = 0;
	uint32_t peak_flag = 0;
	dev::EntFrameJob fj = ent_frame_job(t, block.data(), out, (uint32_t)cap, &size, &peak_flag);
	static dev::EntTables tables[2]; static bool ready = false;
	if (!ready) { ent_build_tables(&tables[0], 1); ent_build_tables(&tables[1], 2); ready = true; }
	std::vector<dev::EntSegState> segs(jobs.segjobs.size());
	std::vector<dev::EntBandState> bstate(jobs.bands.size());
	const int nseg = (int)jobs.segjobs.size(), nb = (int)jobs.bands.size();
	const dev::EntBatchGeom geom = { nseg, nb, 0 };
	std::vector<uint32_t> tokens((size_t)nseg * dev::ENT_TOK_STRIDE, 0xdeadbeefu);       // k_ent_count's token lists for k_ent_emit
	hipemu::launch(dim3((nseg + dev::ENT_WAVES * dev::ENT_COUNT_SEGS - 1) / (dev::ENT_WAVES * dev::ENT_COUNT_SEGS)), dim3(dev::ENT_THREADS), [&] { dev::k_ent_count(jobs.segjobs.data(), geom, nseg, segs.data(), tables, &peak_flag, tokens.data(), 0, nseg); });
	hipemu::launch(dim3(nb), dim3(dev::ENT_THREADS), [&] { dev::k_ent_scan(jobs.bands.data(), segs.data(), bstate.data(), tables); });
	hipemu::launch(dim3(1, 3), dim3(dev::ENT_THREADS), [&] { dev::k_ent_layout(&fj, jobs.bands.data(), segs.data(), bstate.data(), tables); });
	{
		dev::EntPeakHoles which; which.n = 0;
		for (size_t h = 0; h < t.holes.size() && which.n < 7; h++) if (t.holes[h].kind == 2) which.hole[which.n++] = (int)h;
		if (which.n) hipemu::launch(dim3(2, (unsigned)which.n, 1), dim3(dev::ENT_THREADS), [&] { dev::k_ent_peaks(&fj, which, jobs.bands.data(), jobs.segjobs.data(), geom, segs.data(), bstate.data()); });
	}
	hipemu::launch(dim3((nseg + dev::ENT_WAVES * dev::ENT_EMIT_SEGS - 1) / (dev::ENT_WAVES * dev::ENT_EMIT_SEGS)), dim3(dev::ENT_THREADS), [&] { dev::k_ent_emit(nseg, segs.data(), tables, tokens.data()); });
	return (peak_flag & 2u) ? -100 : (long)size;
}

extern "C" long emu_entropy_encode(int width, int height, int pixel_kind, int quality, unsigned frame_number, int16_t *coeffs, const uint8_t *meta, size_t meta_size,
                                   uint8_t *out, size_t cap)
{
	return emu_entropy_encode2(width, height, pixel_kind, quality, frame_number, coeffs, meta, meta_size, out, cap, 0);
}

// GPU entropy decoder under emulation: parse on the host (product parser), decode every band with k_dec_bands / k_dec_lowpass.
extern "C" int emu_entropy_decode(const uint8_t *sample, size_t size, int pixel_kind, int16_t *coeffs, size_t coeff_elems, int parallel)
{
	using namespace cfhd;
	ParsedSample ps;
	if (parse_sample(sample, size, &ps) != 0) return -1;
	FramePlan plan;
	if (!build_frame_plan(&plan, ps.width, ps.display_height, pixel_kind, ps.encoded_format)) return -2;
	plan.precision = ps.precision;
	if (coeff_elems < plan.coeff_elems) return -3;
	if (!parallel) memset(coeffs, 0, (size_t)plan.coeff_elems * 2);      // k_dec_bands_par clears its bands itself
	std::vector<dev::DecBandJob> bands; std::vector<dev::DecLowpassJob> lows;
	if (!dec_build_jobs(ps, plan, sample, coeffs, pixel_kind, &bands, &lows)) return -4;
	static std::vector<uint32_t> tables;
	if (tables.empty()) tables = build_dec_tables(1);
	int errors = 0;
	const int nb = (int)bands.size();
	if (parallel == 2) {
		// device-resident path: two copies of the sample, parsed by k_dec_parse (no host parser output reaches the kernels)
		const size_t stride = (size + 256 + 255) & ~(size_t)255;
		std::vector<uint8_t> raw(stride * 2 + 256, 0);
		uint8_t *two = (uint8_t *)(((uintptr_t)raw.data() + 255) & ~(uintptr_t)255);
		memcpy(two, sample, size); memcpy(two + stride, sample, size);
		const uint32_t sizes[2] = { (uint32_t)size, (uint32_t)size };
		std::vector<int16_t> pyr((size_t)plan.coeff_elems * 2, 77);
		dev::DecPlan dp; dec_build_plan(plan, pixel_kind, &dp);
		std::vector<dev::DecBandJob> bj((size_t)dp.bands_per_frame * 2); std::vector<dev::DecLowpassJob> lj((size_t)plan.num_channels * 2);
		hipemu::launch(dim3(2), dim3(dev::DEC_PARSE_THREADS), [&] { dev::k_dec_parse(two, stride, sizes, 2, &dp, pyr.data(), plan.coeff_elems, bj.data(), lj.data(), &errors, nullptr); });
		if (errors) return -20 - errors;
		hipemu::launch(dim3((unsigned)bj.size()), dim3(dev::DECP_THREADS), [&] { dev::k_dec_bands_par(bj.data(), (const dev::DecTables *)tables.data(), &errors); });
		hipemu::launch(dim3(4, (unsigned)lj.size()), dim3(256), [&] { dev::k_dec_lowpass(lj.data()); });
		if (errors) return -10 - errors;
		memcpy(coeffs, pyr.data() + plan.coeff_elems, (size_t)plan.coeff_elems * 2);     // second frame out; both must agree where bands live
		for (size_t k = 0; k < bands.size(); k++) {
			const size_t off = (size_t)(bands[k].dst - coeffs);
			if (memcmp(pyr.data() + off, pyr.data() + plan.coeff_elems + off, (size_t)bands[k].n * 2)) return -30;
		}
		return 0;
	}
	if (parallel == 3)       // the low-latency shape: 512 threads on 128-bit subsequences
		hipemu::launch(dim3(nb), dim3(dev::DECP_LL_THREADS), [&] { dev::k_dec_bands_par_ll(bands.data(), (const dev::DecTables *)tables.data(), &errors); });
	else if (parallel)
		hipemu::launch(dim3(nb), dim3(dev::DECP_THREADS), [&] { dev::k_dec_bands_par(bands.data(), (const dev::DecTables *)tables.data(), &errors); });
	else
		hipemu::launch(dim3((nb + dev::DEC_THREADS - 1) / dev::DEC_THREADS), dim3(dev::DEC_THREADS),
		               [&] { dev::k_dec_bands(bands.data(), nb, (const dev::DecTables *)tables.data(), &errors); });
	hipemu::launch(dim3(4, (unsigned)lows.size()), dim3(256), [&] { dev::k_dec_lowpass(lows.data()); });
	return errors ? -10 - errors : 0;
}

// The round-2 decoder (cfhd_dec_kernels.h) under emulation.  mode 0: host parser, speculation on; 1: speculation off (every chunk but the first
// assumes a wrong start, so k_dec_chain has to index them again: the repair path); 2: two copies of the sample in "device" memory, parsed by
// k_dec_parse and numbered by k_dec_plan.  grid: workgroups of the grid-stride kernels (small grids exercise the stride loops).
uint32_t g_dx_stats[16];
extern "C" uint32_t *emu_dx_stats() { return g_dx_stats; }
extern "C" int emu_entropy_decode_dx(const uint8_t *sample, size_t size, int pixel_kind, int16_t *coeffs, size_t coeff_elems, int mode, int grid)
{
	memset(g_dx_stats, 0, sizeof(g_dx_stats));
	using namespace cfhd;
	const bool full_tiles = (mode & 8) != 0, one_wave = (mode & 16) != 0;        // + 16: k_dec_tiles with one wave per workgroup (a tile then takes several rounds of 64 pieces)
	mode &= 7;
	ParsedSample ps;
	if (parse_sample(sample, size, &ps) != 0) return -1;
	FramePlan plan;
	if (!build_frame_plan(&plan, ps.width, ps.display_height, pixel_kind, ps.encoded_format)) return -2;
	plan.precision = ps.precision;
	if (coeff_elems < plan.coeff_elems) return -3;
	static dev::DecIdxTables tables; static bool ready = false;
	if (!ready) { if (!build_dec_index_tables(1, &tables)) return -5; ready = true; }
	dev::DecPlan dp; dec_build_plan(plan, pixel_kind, &dp);
	const int nframes = mode == 2 ? 2 : 1, nch = plan.num_channels, njobs = dp.bands_per_frame * nframes;
	std::vector<dev::DecBandJob> jobs((size_t)njobs);
	std::vector<dev::DecLowpassJob> lows((size_t)nch * nframes);
	std::vector<dev::DecDiffJob> diffs((size_t)nch * nframes);            // interlaced samples: the difference-coded band of every channel
	std::vector<int16_t> pyr((size_t)plan.coeff_elems * nframes, 77);       // the tile kernel must write every coefficient of every band itself
	const size_t stride = (size + 256 + 255) & ~(size_t)255;
	std::vector<uint8_t> raw(stride * 2 + 512, 0);
	uint8_t *two = (uint8_t *)(((uintptr_t)raw.data() + 255) & ~(uintptr_t)255);
	memcpy(two, sample, size); memcpy(two + stride, sample, size);
	int errors = 0;
	const uint32_t max_chunks = (uint32_t)(nframes * (size / dev::DX_CHUNK_BYTES + dp.bands_per_frame + 1));
	std::vector<dev::DxChunkDesc> chunk_job(max_chunks); std::vector<uint32_t> counters(8, 0);
	if (mode == 2) {
		const uint32_t sizes[2] = { (uint32_t)size, (uint32_t)size };
		hipemu::launch(dim3(2), dim3(dev::DEC_PARSE_THREADS), [&] { dev::k_dec_parse(two, stride, sizes, 2, &dp, pyr.data(), plan.coeff_elems, jobs.data(), lows.data(), &errors, diffs.data()); });
		if (errors) return -20 - errors;
		hipemu::launch(dim3(1), dim3(1024), [&] { dev::k_dec_plan(jobs.data(), njobs, max_chunks, counters.data(), &errors); });
		hipemu::launch(dim3((unsigned)(njobs + dev::DX_WAVES - 1) / dev::DX_WAVES), dim3(dev::DX_THREADS), [&] { dev::k_dec_plan_fill(jobs.data(), njobs, chunk_job.data(), counters.data()); });
		if (errors) return -40 - errors;
	} else {
		if (!dx_build_jobs(ps, plan, dp, two, pyr.data(), pixel_kind, 0, 1, jobs.data(), lows.data(), false, diffs.data())) return -4;
		std::vector<dev::DxChunkDesc> cj;
		counters[0] = dx_number_chunks(jobs.data(), njobs, &cj);
		if (counters[0] > max_chunks) return -6;
		std::copy(cj.begin(), cj.end(), chunk_job.begin());
	}
	const uint32_t nchunks = counters[0];
	std::vector<uint32_t> entries((size_t)nchunks * dev::DX_ENTRY_STRIDE + 16, 0xdeadbeefu), chunk_base(nchunks + 1, 0xdeadbeefu);
	std::vector<dev::DxChunkRec> recs(nchunks + 1);
	std::vector<dev::DxBandSum> sums((size_t)njobs);
	// tiles of at most 1536 coefficients (the small test frames then have bands of many tiles, and tiles met by more pieces than a workgroup has threads), or -- mode + 8 -- of the product's size
	const uint32_t tile_max = full_tiles ? (uint32_t)dev::DX_TILE : 1536u;
	const dev::DxTilePlan tp = dx_tile_plan(plan, dp, nframes, false, false, false, tile_max);
	std::vector<dev::DxChunkAlt> alts(nchunks + 1);
	std::vector<dev::DxReindex> reindex((size_t)nchunks + 1);
	std::vector<uint32_t> repair_list((size_t)njobs + 1);
	counters[1] = 0; counters[2] = 0; counters[3] = 0; counters[4] = 0;
	const uint32_t alt_slots = 3;                       // room for the candidates of one or two chunks only: both ways of k_dec_reindex (copy, index again) are exercised
	std::vector<uint32_t> alt_entries((size_t)alt_slots * dev::DX_ENTRY_STRIDE + 16, 0xdeadbeefu);
	hipemu::launch(dim3((unsigned)grid), dim3(dev::DX_THREADS), [&] { dev::k_dec_index(chunk_job.data(), counters.data(), &tables, entries.data(), recs.data(), alts.data(), mode != 1, g_dx_stats, alt_entries.data(), alt_slots, &counters[3], &counters[4]); });
	hipemu::launch(dim3((unsigned)(njobs + dev::DX_WAVES - 1) / dev::DX_WAVES), dim3(dev::DX_THREADS), [&] { dev::k_dec_chain(jobs.data(), njobs, recs.data(), alts.data(), chunk_base.data(), sums.data(), &errors, repair_list.data(), reindex.data(), counters.data()); });
	hipemu::launch(dim3(2), dim3(dev::DX_THREADS), [&] { dev::k_dec_repair(jobs.data(), &tables, entries.data(), recs.data(), alts.data(), chunk_base.data(), sums.data(), &errors, repair_list.data(), reindex.data(), counters.data(), g_dx_stats); });
	hipemu::launch(dim3(3), dim3(dev::DX_THREADS), [&] { dev::k_dec_reindex(jobs.data(), &tables, entries.data(), reindex.data(), counters.data(), g_dx_stats, alts.data(), alt_entries.data()); });
	g_dx_stats[12] = counters[1]; g_dx_stats[13] = counters[2];
	std::vector<dev::DxTileDesc> tile_desc(tp.total + 1);
	auto tile_pass = [&](const dev::DxTilePlan &p_, unsigned long long *m_, uint32_t per_) {
		memset(tile_desc.data(), 0xdb, tile_desc.size() * sizeof(dev::DxTileDesc));
		hipemu::launch(dim3((p_.total + dev::DX_TILE_INDEX_TILES - 1) / dev::DX_TILE_INDEX_TILES), dim3(dev::DX_THREADS), [&] { dev::k_dec_tile_index(jobs.data(), p_, entries.data(), chunk_base.data(), sums.data(), tile_desc.data(), m_, per_); });
		if (one_wave) hipemu::launch(dim3((unsigned)grid), dim3(64), [&] { dev::k_dec_tiles<64>(tile_desc.data(), p_.first, p_.total, &tables, entries.data(), chunk_base.data()); });
		else hipemu::launch(dim3((unsigned)grid), dim3(dev::DX_TILE_THREADS), [&] { dev::k_dec_tiles<dev::DX_TILE_THREADS>(tile_desc.data(), p_.first, p_.total, &tables, entries.data(), chunk_base.data()); });
	};
	tile_pass(tp, nullptr, 0u);
	if (!plan.interlaced && plan.encoded_format == ENC_YUV422) {
		// the same tile pass with the level-1 highpass bands as block lists (what k_inv_yuv422_strip_blocks gathers): expanded again they must be the dense bands
		const std::vector<int16_t> dense(pyr);
		int mask_base[kMaxChannels][kNumBands];
		const size_t per_frame = (size_t)dec_block_list_layout(plan, mask_base);
		std::vector<unsigned long long> masks(per_frame * (size_t)nframes + 8, 0x5555555555555555ull);
		const dev::DxTilePlan tl = dx_tile_plan(plan, dp, nframes, false, true, false, tile_max);
		for (int16_t &v : pyr) v = 0x0bad;             // (stale places must never be read back)
		tile_pass(tl, masks.data(), (uint32_t)per_frame);
		for (int f = 0; f < nframes; f++)
			for (int c = 0; c < plan.num_channels; c++)
				for (int b = 1; b < 4; b++) {
					const BandDesc &bd = plan.ch[c].band[0][b];
					const int n = bd.height * bd.pitch;
					const int16_t *lists = pyr.data() + (size_t)f * plan.coeff_elems + bd.offset, *want = dense.data() + (size_t)f * plan.coeff_elems + bd.offset;
					for (int blk = 0; blk < n / 8; blk++) {
						const unsigned long long m = masks[(size_t)f * per_frame + (size_t)mask_base[c][b] + (size_t)(blk >> 6)];
						const int bit = blk & 63;
						int16_t got[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
						if ((m >> bit) & 1ull) memcpy(got, lists + ((size_t)(blk & ~63) + (size_t)__builtin_popcountll(m & ((1ull << bit) - 1ull))) * 8, 16);
						if (memcmp(got, want + (size_t)blk * 8, 16)) return -50;
					}
				}
		// the bands that stay dense (levels 2 and 3) must be what they were
		for (int f = 0; f < nframes; f++)
			for (int c = 0; c < plan.num_channels; c++)
				for (int lv = 1; lv < kNumLevels; lv++)
					for (int b = 1; b < 4; b++) {
						const BandDesc &bd = plan.ch[c].band[lv][b];
						if (memcmp(pyr.data() + (size_t)f * plan.coeff_elems + bd.offset, dense.data() + (size_t)f * plan.coeff_elems + bd.offset, (size_t)bd.height * bd.pitch * 2)) return -51;
					}
		pyr = dense;
	}
	// (both shapes of the un-differencing: mode 1 = the workgroup-per-band kernel for every band, else a wave per row for the bands without a peak table)
	if (mode != 1) hipemu::launch(dim3((unsigned)diffs.size(), 3), dim3(64 * dev::DXR_WAVES), [&] { dev::k_dec_undiff_rows(diffs.data()); });
	hipemu::launch(dim3((unsigned)diffs.size(), 5), dim3(dev::DXU_THREADS), [&] { dev::k_dec_undiff(diffs.data(), &errors, mode != 1 ? 1 : 0); });
	hipemu::launch(dim3(4, (unsigned)lows.size()), dim3(256), [&] { dev::k_dec_lowpass(lows.data()); });
	if (errors) return -10 - errors;
	memcpy(coeffs, pyr.data() + (size_t)(nframes - 1) * plan.coeff_elems, (size_t)plan.coeff_elems * 2);
	if (nframes == 2 && memcmp(pyr.data(), pyr.data() + plan.coeff_elems, (size_t)plan.final_elems * 2)) return -30;
	return 0;
}

// the running sums of a difference-coded band (no peak table): a wave per row (round 4)
extern "C" void emu_dec_undiff_rows(int16_t *band, int width, int height, int pitch)
{
	cfhd::dev::DecDiffJob job = { band, width, height, pitch, nullptr, 0u, 0 };
	hipemu::launch(dim3(1, 3), dim3(64 * cfhd::dev::DXR_WAVES), [&] { cfhd::dev::k_dec_undiff_rows(&job); });
}
