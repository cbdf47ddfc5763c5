// tests/hooks/cfhd_hooks.cpp -- TEST INFRASTRUCTURE ONLY.  Host-only entry points (prefix cfhd_amd_) with which the CPU-side tests
// exercise the product's own syntax layer (plan geometry, quantizer derivation, sample writer/parser, host VLC) without a GPU.
// Built by tests/cfhd_testlib.py into tests/_build/libcfhd_hooks.so together with the product's host-only sources
// (cfhd_tables.cpp, cfhd_bitstream.cpp, cfhd_metadata.cpp); not linked into libcfhd_amd.so.
#include "cfhd_core.h"
#include "cfhd_bitstream.h"
#include "cfhd_gop.h"
#include <string.h>
#include <vector>

using namespace cfhd;

extern "C" {

// out[0]=coeff_elems, [1]=final_elems, [2]=num_channels, [3]=precision, [4]=midpoint_prequant, [5..7]=prescale,
// then per channel, per level, per band: width,height,pitch,offset,quant,scale (6 ints).
int cfhd_amd_plan_info(int width, int height, int pixel_kind, int encoded_format, int quality, int progressive, int *out)
{
	FramePlan plan;
	if (!build_frame_plan(&plan, width, height, pixel_kind, encoded_format)) return -1;
	QuantState st = {0, -1, 0};
	derive_quantization(&plan, quality, progressive != 0, 0.0f, &st);
	int n = 0;
	out[n++] = (int)plan.coeff_elems; out[n++] = (int)plan.final_elems; out[n++] = plan.num_channels; out[n++] = plan.precision;
	out[n++] = plan.midpoint_prequant; out[n++] = plan.prescale[0]; out[n++] = plan.prescale[1]; out[n++] = plan.prescale[2];
	for (int c = 0; c < plan.num_channels; c++)
		for (int lv = 0; lv < kNumLevels; lv++)
			for (int b = 0; b < kNumBands; b++) {
				const BandDesc &d = plan.ch[c].band[lv][b];
				out[n++] = d.width; out[n++] = d.height; out[n++] = d.pitch; out[n++] = (int)d.offset; out[n++] = d.quant; out[n++] = d.scale;
			}
	return n;
}

// Two-frame group (cfhd_gop.h): plan geometry + quantizer as the product derives them.  out: coeff_elems, then per channel, per wavelet (6):
// type, level, nbands, width, height, pitch, prescale, then per band (4): offset, quant, scale.  Returns the number of ints, -1 when the
// geometry / quality is not served.
int cfhd_amd_gop_plan_info(int width, int height, int pixel_kind, int quality, long long *out)
{
	GopPlan plan; QuantState st = {0, -1, 0};
	const bool interlaced = (pixel_kind & 0x100) != 0; pixel_kind &= 0xff;      // (bit 8 of the pixel kind: CFHD_ENCODING_FLAGS_YUV_INTERLACED on top of the group flag)
	if (!build_gop_plan(&plan, width, height, pixel_kind, interlaced) || !derive_gop_quantization(&plan, quality, &st)) return -1;
	int n = 0;
	out[n++] = (long long)plan.coeff_elems; out[n++] = plan.height; out[n++] = plan.midpoint_prequant;
	for (int c = 0; c < 3; c++)
		for (int k = 0; k < kGopWavelets; k++) {
			const GopWavelet &w = plan.ch[c].w[k];
			out[n++] = w.type; out[n++] = w.level; out[n++] = w.nbands; out[n++] = w.width; out[n++] = w.height; out[n++] = w.pitch; out[n++] = w.prescale;
			for (int b = 0; b < 4; b++) { out[n++] = (long long)w.offset[b]; out[n++] = w.quant[b]; out[n++] = w.scale[b]; }
		}
	return n;
}
// The group sample / sequence header / P-frame sample from a group pyramid (host writer).  kind 0 group, 1 sequence header, 2 P-frame.
size_t cfhd_amd_write_gop_host(int kind, int width, int height, int pixel_kind, int quality, unsigned frame_number, const int16_t *coeffs,
                               const uint8_t *meta_global, size_t meta_global_size, uint8_t *out, size_t cap)
{
	GopPlan plan; QuantState st = {0, -1, 0};
	const bool interlaced = (pixel_kind & 0x100) != 0; pixel_kind &= 0xff;
	if (!build_gop_plan(&plan, width, height, pixel_kind, interlaced) || !derive_gop_quantization(&plan, quality, &st)) return 0;
	const int input_format = pixel_kind == PIX_2VUY ? 1 : 2;
	if (kind == 1) return write_sequence_header(plan, input_format, out, cap);
	if (kind == 2) return write_pframe_sample(plan, frame_number, out, cap);
	SampleHeaderInfo hdr = { frame_number, input_format, 2, quality, !interlaced, meta_global, meta_global_size, nullptr, 0 };
	return write_group_sample(plan, hdr, coeffs, out, cap);
}

// The dequantized group pyramid of a group sample (host parser + host VLC decoder; lowpass bias as the reference's decoder applies it to groups
// for 8-bit output).  0 on success.
int cfhd_amd_decode_group_host(const uint8_t *sample, size_t size, int pixel_kind, int16_t *coeffs, size_t coeff_elems)
{
	ParsedGroup pg;
	if (parse_group_sample(sample, size, &pg) != 0) return -1;
	GopPlan gp;
	if (!build_gop_plan(&gp, pg.width, pg.display_height ? pg.display_height : pg.height, pixel_kind) || gp.coeff_elems != coeff_elems) return -2;
	memset(coeffs, 0, coeff_elems * 2);
	for (int c = 0; c < 3; c++) {
		const GopWavelet &top = gp.ch[c].w[5];
		const ParsedBand &lp = pg.lowpass[c];
		if (!lp.present || lp.width != top.width || lp.height != top.height) return -3;
		const int bias = 2 * lowpass_bias(10, top.width, pixel_kind);
		for (int r = 0; r < top.height; r++)
			for (int x = 0; x < top.width; x++) {
				const uint8_t *p = sample + lp.offset + ((size_t)r * top.width + x) * 2;
				int v = (int16_t)((p[0] << 8) | p[1]); v += bias;
				coeffs[top.offset[0] + (size_t)r * top.pitch + x] = (int16_t)(v > 0x7fff ? 0x7fff : v);
			}
		static const int coded[5] = { 5, 4, 3, 1, 0 };
		for (int k : coded) {
			const GopWavelet &wv = gp.ch[c].w[k];
			for (int b = (k == 3 ? 0 : 1); b < 4; b++) {
				const ParsedBand &pb = pg.band[c][k][b];
				if (!pb.present || pb.width != wv.width || pb.height != wv.height) return -4;
				int16_t *dst = coeffs + wv.offset[b];
				if (pb.codebook < 0) {
					for (int r = 0; r < wv.height; r++)
						for (int x = 0; x < wv.width; x++) { const uint8_t *p = sample + pb.offset + ((size_t)r * wv.width + x) * 2; dst[(size_t)r * wv.pitch + x] = (int16_t)(((p[0] << 8) | p[1]) * pb.quant); }
				} else if (vlc_decode_band(sample + pb.offset, pb.bytes, wv.width, wv.height, wv.pitch, pb.quant, pb.codebook, dst)) return -5;
				// interlaced groups: peak values, then every row becomes its running sum (decoder.c:19809, :20822)
				if (pb.difference) finish_difference_band(dst, wv.width, wv.height, wv.pitch, pb.peak_level ? sample + pb.peak_offset : nullptr, pb.peak_level ? size - pb.peak_offset : 0, pb.peak_level);
			}
		}
	}
	return 0;
}

// The device an encoder-pool worker / decoder handle is dealt (cfhd_core.h unit_device): host logic, no GPU involved.
int cfhd_amd_unit_device(int i, int ndevices, const char *pinned_env, const char *list_env) { return unit_device(i, ndevices, pinned_env, list_env); }

// Quantizer tables of a sequence of frames under rate feedback: frame f is derived with lastgopbitcount = 8 * sample_bytes[f - 1]
// (0 for the first), the state carried from frame to frame as the encoder carries it.  out: per frame, per channel, the 9 highpass
// divisors in coding order (level 3 LH HL HH, level 2, level 1).
int cfhd_amd_quant_sequence(int width, int height, int pixel_kind, int encoded_format, int quality, int progressive,
                            const long long *sample_bytes, int nframes, int *out)
{
	FramePlan plan;
	if (!build_frame_plan(&plan, width, height, pixel_kind, encoded_format)) return -1;
	plan.interlaced = !progressive;
	QuantState st = {0, -1, 0};
	int n = 0;
	for (int f = 0; f < nframes; f++) {
		st.lastgopbitcount = f ? (int64_t)sample_bytes[f - 1] * 8 : 0;
		derive_quantization(&plan, quality, progressive != 0, 0.0f, &st);
		for (int c = 0; c < plan.num_channels; c++)
			for (int lv = kNumLevels - 1; lv >= 0; lv--)
				for (int b = 1; b < kNumBands; b++) out[n++] = plan.ch[c].band[lv][b].quant;
	}
	return n;
}

// Highpass divisors a sample's band headers carry, same order as above (9 per channel).  Returns the count or < 0.
int cfhd_amd_sample_quants(const uint8_t *sample, size_t size, int *out)
{
	ParsedSample ps;
	if (parse_sample(sample, size, &ps) != 0) return -1;
	int n = 0;
	for (int c = 0; c < ps.num_channels; c++)
		for (int lv = kNumLevels - 1; lv >= 0; lv--)
			for (int b = 1; b < kNumBands; b++) out[n++] = ps.high[c][lv][b].quant;
	return n;
}

// Assemble a sample on the host from a coefficient pyramid laid out per the plan (host VLC).
size_t cfhd_amd_write_sample_host(int width, int height, int pixel_kind, int encoded_format, int quality, int progressive,
                                  int input_format, int color_space, unsigned frame_number,
                                  const int16_t *coeffs, const uint8_t *meta_global, size_t meta_global_size,
                                  const uint8_t *meta_local, size_t meta_local_size, uint8_t *out, size_t cap)
{
	FramePlan plan;
	if (!build_frame_plan(&plan, width, height, pixel_kind, encoded_format)) return 0;
	QuantState st = {0, -1, 0};
	derive_quantization(&plan, quality, progressive != 0, 0.0f, &st);
	SampleHeaderInfo hdr = { frame_number, input_format, color_space, quality, progressive != 0, meta_global, meta_global_size, meta_local, meta_local_size };
	BandSource src; src.coeffs = coeffs;
	return write_sample(plan, hdr, src, out, cap);
}

// Parse a sample and entropy-decode every band on the host into a pyramid laid out per the plan
// (highpass values already multiplied by their quant, as the reference's FSM decoder delivers them).
int cfhd_amd_decode_bands_host(const uint8_t *sample, size_t size, int pixel_kind, int16_t *coeffs, size_t coeff_elems, int *info /*8 ints*/, int apply_lowpass_bias)
{
	ParsedSample ps;
	int rc = parse_sample(sample, size, &ps);
	if (rc) return rc < 0 ? rc : -30;
	FramePlan plan;
	const int mosaic = ps.encoded_format == ENC_BAYER ? 2 : 1;       // a Bayer sample carries the component plane size; the plan is built from the mosaic's
	if (!build_frame_plan(&plan, ps.width * mosaic, ps.display_height * mosaic, pixel_kind, ps.encoded_format)) return -20;
	if (coeff_elems < plan.coeff_elems) return -21;
	info[0] = ps.width; info[1] = ps.height; info[2] = ps.display_height; info[3] = ps.num_channels; info[4] = ps.precision;
	info[5] = ps.encoded_format; info[6] = ps.prescale_table; info[7] = ps.frame_number;
	memset(coeffs, 0, (size_t)plan.coeff_elems * 2);
	for (int c = 0; c < plan.num_channels; c++) {
		const ParsedBand &lp = ps.lowpass[c];
		const BandDesc &ll = plan.ch[c].band[2][0];
		if (!lp.present || lp.width != ll.width || lp.height != ll.height) return -22;
		const int lowpass_offset = apply_lowpass_bias ? lowpass_bias(plan.precision, ll.width, pixel_kind, c) : 0;
		for (int r = 0; r < ll.height; r++)
			for (int x = 0; x < ll.width; x++) {
				const uint8_t *p = sample + lp.offset + ((size_t)r * ll.width + x) * 2;
				int v = (int16_t)((p[0] << 8) | p[1]);
				v += lowpass_offset;                      // Codec/decoder.c:12240-12290 channeloffset
				coeffs[ll.offset + (size_t)r * ll.pitch + x] = (int16_t)(v > 0x7fff ? 0x7fff : v);
			}
		for (int lv = 0; lv < kNumLevels; lv++)
			for (int b = 1; b < 4; b++) {
				const ParsedBand &pb = ps.high[c][lv][b];
				const BandDesc &bd = plan.ch[c].band[lv][b];
				if (!pb.present || pb.width != bd.width || pb.height != bd.height) return -23;
				rc = vlc_decode_band(sample + pb.offset, pb.bytes, bd.width, bd.height, bd.pitch, pb.quant, pb.codebook, coeffs + bd.offset);
				if (rc) return rc * 100 - (c * 9 + lv * 3 + b);
				if (pb.difference) finish_difference_band(coeffs + bd.offset, bd.width, bd.height, bd.pitch, pb.peak_level ? sample + pb.peak_offset : nullptr,
				                                          pb.peak_level ? size - pb.peak_offset : 0, pb.peak_level);
			}
	}
	return 0;
}

} // extern "C"

// Test hook: the Bayer encode curve the device path uploads (must equal the oracle's restatement entry for entry).
extern "C" void cfhd_amd_bayer_curve(int precision, uint16_t *curve) { cfhd::build_bayer_log90_curve(precision, curve); }
