// cfhd_api_params.h -- part of cfhd_api.cpp (one translation unit: the parts are #included there in this order, they share the handle types of an unnamed namespace).
// Pixel formats, encode parameters and quantizer plans of a prepared encoder (CFHD_PrepareToEncode / CFHD_PrepareEncoderPool), metadata handle of the encoder side, stage profile.


int pixel_kind_of(uint32_t fmt)
{
	if (fmt == FMT_YUY2 || fmt == FMT_YUYV) return PIX_YUY2;
	if (fmt == FMT_2VUY) return PIX_2VUY;
	if (fmt == FMT_RG48) return PIX_RG48;
	if (fmt == FMT_B64A) return PIX_B64A;
	if (fmt == FMT_BYR4) return PIX_BYR4;
	if (fmt == FMT_BYR5) return PIX_BYR5;
	if (fmt == FMT_RG64) return PIX_RG64;
	if (fmt == FMT_YU64) return PIX_YU64;
	if (fmt == FMT_V210) return PIX_V210;
	if (fmt == FMT_RG24) return PIX_RG24;
	if (fmt == FMT_BGRA) return PIX_BGRA;
	if (fmt == FMT_BGRa) return PIX_BGRa;
	if (fmt == FMT_R210) return PIX_R210;
	if (fmt == FMT_DPX0) return PIX_DPX0;
	if (fmt == FMT_AB10 || fmt == FMT_RG30) return PIX_AB10;
	if (fmt == FMT_AR10) return PIX_AR10;
	return PIX_NONE;
}
// COLOR_FORMAT_UYVY = 1 / COLOR_FORMAT_YUYV = 2 / COLOR_FORMAT_BGRA64 (b64a) = 30 / COLOR_FORMAT_RG48 = 120 (Codec/color.h)
int color_format_of(int kind) { return kind == PIX_2VUY ? 1 : (kind == PIX_RG48 ? 120 : (kind == PIX_B64A ? 30 : (kind == PIX_BYR4 ? 104 : (kind == PIX_BYR5 ? 105 : (kind == PIX_RG64 ? 121 : (kind == PIX_YU64 ? 12 : (kind == PIX_V210 ? 10 : (kind == PIX_RG24 ? 7 : (kind == PIX_BGRA ? 32 : (kind == PIX_BGRa ? 9 : (kind == PIX_R210 ? 123 : (kind == PIX_DPX0 ? 128 : (kind == PIX_AB10 ? 125 : (kind == PIX_AR10 ? 124 : 2)))))))))))))); }   // COLOR_FORMAT_* of Codec/color.h
int pixel_bytes_of(int kind) { return kind == PIX_RG24 ? 3 : (kind == PIX_BGRA || kind == PIX_BGRa || (kind >= PIX_R210 && kind <= PIX_AR10)) ? 4 : kind == PIX_RG48 ? 6 : (kind == PIX_B64A || kind == PIX_RG64 ? 8 : (kind == PIX_YU64 || kind == PIX_V210 ? 4 : 2)); }

// ---- metadata handle shared by the encoder-side API (CSampleEncodeMetadata) ----
struct EncMetadata {
	std::mutex lock;
	MetaBlock global, local;
	bool changed = false;
};

// Settings of the ROCm runtime this library is measured with, for a process that did not choose them itself (set when the library is loaded, i.e. before the first
// HIP call of an application that links it; an application that set them keeps its values): sample downloads on the SDMA engines instead of blit kernels that
// compete with the codec's kernels (HSA_ENABLE_SDMA, bench.py: DESIGN.md section 5), kernel arguments written to device memory (HIP_FORCE_DEV_KERNARG: the
// host-fed round trip of tools/cabi_bench 2.4-2.9 k -> 3.6 k fps, profiles/r05_e_*).
// (Not set here: GPU_MAX_HW_QUEUES.  The runtime maps all HIP streams of a process onto 4 hardware queues by default.  For several batches in flight -- a dozen streams -- 16
// queues are worth +8 % (bench.py sets it for itself); the same setting costs the many-thread C ABI case, pool workers + decoder handles, 10-25 %:
// profiles/r05_o_*, r05_p_*.)
// CFHD_AMD_SET_RUNTIME_DEFAULTS=0: leave the process environment alone (an application that shares the process with other HIP users and sets what it wants itself:
// INTEGRATION.md section 4 lists the variables; advisor, round 5).
__attribute__((constructor)) static void cfhd_amd_runtime_defaults()
{
	const char *e = getenv("CFHD_AMD_SET_RUNTIME_DEFAULTS");
	if (e && e[0] == '0') return;
	setenv("HSA_ENABLE_SDMA", "1", 0); setenv("HIP_FORCE_DEV_KERNARG", "1", 0);
}

// CFHD_AMD_PROFILE=1: where the wall time of the synchronous calls goes (printed when the handle is closed)
bool profile_enabled() { static const bool on = [] { const char *e = getenv("CFHD_AMD_PROFILE"); return e && atoi(e) != 0; }(); return on; }
double wall_now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
struct StageProfile {
	double t[6] = {0, 0, 0, 0, 0, 0}; long calls = 0; double last = 0;
	void start() { if (profile_enabled()) last = wall_now(); }
	void mark(int k) { if (profile_enabled()) { const double n = wall_now(); t[k] += n - last; last = n; } }
	void report(const char *what, const char *const *names, int n) const
	{
		if (!profile_enabled() || !calls) return;
		fprintf(stderr, "[cfhd_amd] %s: %ld calls;", what, calls);
		for (int k = 0; k < n; k++) fprintf(stderr, " %s %.1f us;", names[k], 1e6 * t[k] / calls);
		fprintf(stderr, "\n");
	}
};

struct EncodeParams {
	int width = 0, height = 0;
	uint32_t pixel_format = 0;
	int pixel_kind = 0, encoded_format = 0;
	uint32_t flags = 0;
	int quality = 0;
	bool progressive = true;
	int color_space = 2;
	FramePlan plan;
	QuantState qstate = {0, -1, 0};
	int api_encoded = 0, api_quality = 0;         // the encoded format and quality as the caller passed them (quality gets format marks OR-ed in below)
	bool gop = false; GopPlan gplan;              // CFHD_ENCODING_FLAGS_YUV_2FRAME_GOP: two frames per sample (cfhd_gop.h)
	QuantState gstate = {0, -1, 0};               // the quantizer state of the group encoder (rate feedback from the last key sample)
	bool valid = false;
};

int make_params(EncodeParams &p, int w, int h, uint32_t fmt, int encoded, uint32_t flags, int quality)
{
	p.valid = false;
	p.api_encoded = encoded; p.api_quality = quality;
	int kind = pixel_kind_of(fmt);
	if (kind == PIX_NONE) return ERR_BADFORMAT;
	// CFHD_ENCODED_FORMAT_YUV_422 (0) from the packed 4:2:2 formats, CFHD_ENCODED_FORMAT_RGB_444 (1) from RG48; the cross
	// combinations (4:4:4 input subsampled to 4:2:2, ...) go through ConvertLib in the reference and are not built
	// CFHD_ENCODED_FORMAT_RGBA_4444 (2) from b64a
	const bool rgb8 = kind == PIX_RG24 || kind == PIX_BGRA || kind == PIX_BGRa;       // 8-bit RGB(A) input, towards RGB 4:4:4 and YUV 4:2:2 (alpha dropped)
	const bool rgb10 = kind >= PIX_R210 && kind <= PIX_AR10;                             // 10-bit RGB in 32-bit words, to RGB 4:4:4
	// RG64 (16-bit words R, G, B, A; frame.c ConvertRGBA64ToFrame16s): b64a's three encoded formats and marks with the words in another order; its colour format
	// code 121 lies above COLOR_FORMAT_BAYER, so all planes take the full-resolution quantizer tables (RG48's rule, not b64a's; pinned on the reference)
	const bool rg64 = kind == PIX_RG64;
	const bool rgb = kind == PIX_RG48 || kind == PIX_B64A || rg64 || rgb8 || rgb10;
	// CFHD_ENCODED_FORMAT_BAYER (3) from BYR4: default pixel order (red-green) and default encode curve (log 90), i.e. what the
	// reference does without BAYER_FORMAT / ENCODE_CURVE metadata
	// b64a also encodes to RGB 4:4:4 (its default in the reference): the alpha words are dropped, R, G, B as for 4:4:4:4
	// RG48 / b64a encoded as YUV 4:2:2 (rows of TestCFHD's format table): the integer 709 / 601 conversion of frame.c:6731 in the loader of the level-1
	// kernel; the converted frame is quantized as the 4:2:2 frame it has become (derive_quantization).
	// RG24 / BGRA / BGRa encoded as YUV 4:2:2 (the default encoded format of these inputs): frame.c:378 ConvertRGB32to10bitYUVFrame in the loader.
	const bool rgb8_as_422 = rgb8 && encoded == 0;
	const bool deep_rgb_as_422 = ((kind == PIX_RG48 || kind == PIX_B64A || rg64) && encoded == 0) || rgb8_as_422;
	// BGRA / BGRa encoded as RGBA 4:4:4:4 (frame.c:6415 ConvertRGBAtoRGBA64): the alpha byte joins as the fourth plane, curved as b64a's
	const bool rgba8_as_4444 = (kind == PIX_BGRA || kind == PIX_BGRa) && encoded == 2;
	if (!deep_rgb_as_422 && !rgba8_as_4444 && !((kind == PIX_B64A || rg64) && (encoded == 1 || encoded == 2)) && encoded != (kind == PIX_RG48 || rgb8 || rgb10 ? 1 : (kind == PIX_B64A ? 2 : (kind == PIX_BYR4 || kind == PIX_BYR5 ? 3 : 0)))) return ERR_BADFORMAT;
	// CFHD_ENCODING_FLAGS_YUV_INTERLACED: field-based level 1 (encoder.c:2093), built for the packed 4:2:2 formats
	const bool interlaced = (flags & (1u << 0)) != 0;
	if (interlaced && !(kind == PIX_YUY2 || kind == PIX_2VUY)) return ERR_BADFORMAT;
	// CFHD_ENCODING_FLAGS_YUV_2FRAME_GOP (CFHDTypes.h:254, "YUV 4:2:2 only"): two frames per sample through the temporal transform (cfhd_gop.h).
	// With CFHD_ENCODING_FLAGS_YUV_INTERLACED on top, level 1 of both frames is the frame transform (GopPlan::interlaced).
	const bool gop = (flags & (1u << 1)) != 0;
	if (gop && !(kind == PIX_YUY2 || kind == PIX_2VUY)) return ERR_BADFORMAT;
	const int enc = kind == PIX_BYR4 || kind == PIX_BYR5 ? ENC_BAYER : (((kind == PIX_B64A || rg64) && encoded == 2) || rgba8_as_4444 ? ENC_RGBA4444 : (rgb && !deep_rgb_as_422 ? ENC_RGB444 : ENC_YUV422));
	// an encoded format other than the default of the input format marks the quality word (SampleEncoder.cpp:216-219; QUALITY_H 0x0800 in the header)
	if (deep_rgb_as_422 && !rgb8_as_422) quality |= 0x08000000;
	// b64a's default encoded format is RGB 4:4:4; asking for 4:4:4:4 marks the quality word (SampleEncoder.cpp:250-257), which the
	// sample header then carries in QUALITY_H
	if ((kind == PIX_B64A || rg64) && encoded == 2) quality |= 0x20000000;
	// 8-bit RGB sources are marked in the quality word too (encoder.c:2344-2345 ORs 0x1a00000 into it; the header's QUALITY_H reads 0x09a0)
	// (for 8-bit RGB the format that is "other" is RGB 4:4:4: 0x0800 on top of the 0x01a0 of every 8-bit RGB source)
	if (rgb8) quality |= rgb8_as_422 ? 0x01a00000 : (rgba8_as_4444 ? 0x21a00000 : 0x09a00000);
	p.width = w; p.height = h; p.pixel_format = fmt; p.pixel_kind = kind; p.encoded_format = enc; p.flags = flags;
	p.quality = quality; p.progressive = !interlaced;
	const int yuv601 = (flags & (1u << 2)) ? 1 : 2, vsrgb = (flags & (1u << 8)) ? 2 : 1;   // SampleEncoder.cpp:210-212
	p.color_space = ((rgb && !deep_rgb_as_422) || kind == PIX_BYR4 || kind == PIX_BYR5) ? 0 : ((yuv601 == 1 ? 1 : 2) | (vsrgb == 2 ? 4 : 0));           // RGB 4:4:4 samples carry no colour space tag
	if (!build_frame_plan(&p.plan, w, h, kind, enc)) return ERR_BADFORMAT;
	p.plan.color_matrix = (p.color_space & 4 ? 1 : 0) + ((p.color_space & 3) == 1 ? 2 : 0);
	p.plan.interlaced = interlaced;
	p.qstate = {0, -1, 0};
	derive_quantization(&p.plan, quality, p.progressive, 0.0f, &p.qstate);
	p.gop = gop;
	p.gstate = {0, -1, 0};
	if (gop && (!build_gop_plan(&p.gplan, w, h, kind, interlaced) || !derive_gop_quantization(&p.gplan, quality, &p.gstate))) return ERR_BADFORMAT;
	p.valid = true;
	return ERR_OKAY;
}

size_t sample_capacity(const EncodeParams &p) { return (size_t)p.width * p.height * pixel_bytes_of(p.pixel_kind) + 65536; }   // SampleEncoder.cpp:387

// Where the run-length/VLC stage runs: on the GPU by default; CFHD_AMD_ENTROPY=host keeps the reference's arrangement
// (host threads fed by one D2H copy of the quantized bands).  Both produce the same bytes.
bool gpu_entropy_enabled() { const char *e = getenv("CFHD_AMD_ENTROPY"); return !(e && strcmp(e, "host") == 0); }
// CFHD_AMD_ENTROPY=device: (tests) a sample the device stage hands back to the host coder fails the call instead -- proves which stage served a frame or a group
bool gpu_entropy_strict() { const char *e = getenv("CFHD_AMD_ENTROPY"); return e && strcmp(e, "device") == 0; }

// one caller waiting for one frame: its plain buffers are staged in pieces (cfhd_device.hip upload_frame / download_frame)
int sync_stage_pieces() { return 4; }      // (measured with 1 / 2 / 4 / 8 pieces: profiles/r04_j_*, r04_k_*)

int prepare_batch(EncodeBatch &batch, const EncodeParams &p)
{
	if (batch.prepare(p.plan, 1, true)) return ERR_INTERNAL;
	if (gpu_entropy_enabled() && batch.prepare_entropy(sample_capacity(p))) return ERR_INTERNAL;
	return ERR_OKAY;
}

// Encode one frame on one batch slot: upload, forward kernels, entropy kernels, finished sample back (host entropy + syntax with CFHD_AMD_ENTROPY=host).
int encode_one(EncodeBatch &batch, EncodeParams &p, const void *frame, int pitch, uint32_t frame_number,
               MetaBlock global, MetaBlock local, uint8_t *out, size_t cap, size_t *size_out)
{
	int rc;
	meta_remove_hidden(global); meta_remove_hidden(local);
	SampleHeaderInfo hdr = { frame_number, p.pixel_format == FMT_RG30 ? 122 /* COLOR_FORMAT_RG30 */ : color_format_of(p.pixel_kind), p.color_space, p.quality, p.progressive,
	                         global.data(), global.size(), local.data(), local.size() };
	{
		// The one metadata override that changes the sample syntax for 2-D clips (Codec/encoder.c:9043-9046 UpdateEncoderOverrides):
		// TAG_VIDEO_CHANNELS present => ignore_overrides => the channel number tag is written.  More than one video channel is 3-D
		// (two stacked encodes per sample), which is outside the hot path.
		uint32_t sz; unsigned char ty;
		const uint32_t VCHN = CFHD_FOURCC('V', 'C', 'H', 'N');
		const uint8_t *v = meta_find(global.data(), global.size(), VCHN, &sz, &ty);
		if (!v) v = meta_find(local.data(), local.size(), VCHN, &sz, &ty);
		if (v) { uint32_t n; memcpy(&n, v, 4); if (n > 1) return ERR_BADFORMAT; hdr.channel_number_tag = true; }
	}
	// Rate feedback (encoder.c:9442 QuantizationSetQuality + quantize.c:2865 SetTransformQuantization run per frame with the size of the
	// previous sample, encoder.c:9911): FILMSCAN2/3 steer their limiter with it, LOW..HIGH at <= 1080p the bit-rate limiter.  The
	// tables only move when the previous size says so; the device job tables are rewritten only then.
	if (p.qstate.lastgopbitcount) {
		FramePlan next = p.plan;
		derive_quantization(&next, p.quality, p.progressive, 0.0f, &p.qstate);
		bool changed = false;
		for (int c = 0; c < next.num_channels && !changed; c++)
			for (int lv = 0; lv < kNumLevels && !changed; lv++)
				for (int b = 0; b < kNumBands; b++) if (next.ch[c].band[lv][b].quant != p.plan.ch[c].band[lv][b].quant) { changed = true; break; }
		p.plan = next;
		if (changed && batch.update_quant(p.plan)) return ERR_INTERNAL;
	}
	if ((rc = batch.upload_frame(0, frame, pitch))) return ERR_INTERNAL;
	auto host_write = [&]() -> int {
		if ((rc = batch.download_coeffs())) return ERR_INTERNAL;
		if ((rc = batch.wait())) return ERR_INTERNAL;
		BandSource src; src.coeffs = batch.host_coeffs(0);
		size_t n = write_sample(p.plan, hdr, src, out, cap);
		if (!n) return ERR_CODEC_ERROR;
		*size_out = n;
		p.qstate.lastgopbitcount = (int64_t)n * 8;
		return ERR_OKAY;
	};
	// GPU entropy stage: the finished sample comes back, not the coefficients.  A header that does not fit the device template block
	// (several KB of user metadata; the reference takes up to 256 KB) is written by the host writer from the same GPU coefficients.
	if (batch.has_entropy() && batch.entropy().set_frame_header(0, hdr) == 0) {
		if ((rc = batch.launch_forward())) return ERR_INTERNAL;
		if ((rc = batch.entropy().launch())) return ERR_INTERNAL;
		if ((rc = batch.entropy().download())) return ERR_INTERNAL;
		if ((rc = batch.wait())) return ERR_INTERNAL;
		if (!batch.entropy().needs_peak_table(0)) {
			size_t n = batch.entropy().sample_bytes(0);
			if (!n || n > cap) return ERR_CODEC_ERROR;
			memcpy(out, batch.entropy().host_sample(0), n);
			*size_out = n;
			p.qstate.lastgopbitcount = (int64_t)n * 8;
			return ERR_OKAY;
		}
		// an interlaced frame whose field-difference band has more peak values than the entropy stage's positions hold (two million, GpuEntropyEncoder::needs_peak_table):
		// this sample is written by the host writer from the same GPU coefficients
		if (gpu_entropy_strict()) return ERR_INTERNAL;
		return host_write();
	}
	if ((rc = batch.launch_forward())) return ERR_INTERNAL;
	return host_write();
}
