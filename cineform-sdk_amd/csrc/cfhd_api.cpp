// cfhd_api.cpp -- the CFHD_* C ABI (include/cfhd_amd.h) on top of the GPU core.
//
// Mirrors the behaviour of the reference's SDK layer for the hot path:
//   EncoderSDK/CFHDEncoder.cpp + SampleEncoder.cpp (sync encode, metadata handling :744-939),
//   EncoderSDK/CFHDEncoderPool.cpp + EncoderPool.cpp + AsyncEncoder.cpp (async pool, FIFO completion),
//   DecoderSDK/CFHDDecoder.cpp + SampleDecoder.cpp (decode), DecoderSDK/CFHDMetadata.cpp (sample metadata access).
// The reference's CPU thread pool is replaced by HIP-stream frame slots; host threads only submit launches and hand samples back
// (with CFHD_AMD_ENTROPY=host they also run the entropy stage).
#include "../../include/cfhd_amd.h"
#include "cfhd_core.h"
#include "cfhd_bitstream.h"
#include "cfhd_device.h"
#include "cfhd_params.h"
#include "cfhd_metadata.h"
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <vector>
#include <deque>
#include <memory>
#include <mutex>
#include <condition_variable>
#include <thread>
#include <atomic>
#include <random>

using namespace cfhd;

enum {
	ERR_OKAY = 0, ERR_INVALID_ARGUMENT = 1, ERR_OUTOFMEMORY = 2, ERR_BADFORMAT = 3, ERR_BADSCALING = 4, ERR_BADSAMPLE = 5, ERR_INTERNAL = 6,
	ERR_METADATA_END = 9, ERR_UNEXPECTED = 10, ERR_BAD_RESOLUTION = 11, ERR_NOT_FINISHED = 13, ERR_ENCODING_NOT_STARTED = 14, ERR_CODEC_ERROR = 2048,
};

#define FOURCC_BE(a, b, c, d) (((uint32_t)(a) << 24) | ((uint32_t)(b) << 16) | ((uint32_t)(c) << 8) | (uint32_t)(d))
static const uint32_t FMT_YUY2 = FOURCC_BE('Y', 'U', 'Y', '2'), FMT_2VUY = FOURCC_BE('2', 'v', 'u', 'y'), FMT_YUYV = FOURCC_BE('y', 'u', 'y', 'v'),
                      FMT_RG48 = FOURCC_BE('R', 'G', '4', '8'), FMT_B64A = FOURCC_BE('b', '6', '4', 'a'), FMT_BYR4 = FOURCC_BE('B', 'Y', 'R', '4'), FMT_YU64 = FOURCC_BE('Y', 'U', '6', '4'), FMT_V210 = FOURCC_BE('v', '2', '1', '0'), FMT_RG24 = FOURCC_BE('R', 'G', '2', '4'), FMT_BGRA = FOURCC_BE('B', 'G', 'R', 'A'), FMT_BGRa = FOURCC_BE('B', 'G', 'R', 'a'),
                      FMT_R210 = FOURCC_BE('r', '2', '1', '0'), FMT_DPX0 = FOURCC_BE('D', 'P', 'X', '0'), FMT_AB10 = FOURCC_BE('A', 'B', '1', '0'), FMT_AR10 = FOURCC_BE('A', 'R', '1', '0'),
                      FMT_RG30 = FOURCC_BE('R', 'G', '3', '0'), FMT_BYR5 = FOURCC_BE('B', 'Y', 'R', '5'), FMT_RG64 = FOURCC_BE('R', 'G', '6', '4');      // (AJA's name for the AB10 word layout: same pixels, its own colour format code in the sample header)

namespace {

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
__attribute__((constructor)) static void cfhd_amd_runtime_defaults() { setenv("HSA_ENABLE_SDMA", "1", 0); setenv("HIP_FORCE_DEV_KERNARG", "1", 0); }

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

// ---- calls that overlap share launches -----------------------------------------------------------------------------------------------
// The reference's decoder is synchronous per handle and its encoder pool runs one CPU encoder per thread; applications get throughput by
// working on several frames from several threads.  On the GPU one frame per launch sequence leaves the chip mostly idle (a dozen launches
// of kernels that see a single frame), so calls of the same geometry that are in flight at the same time are gathered: every caller stages
// its frame or sample into a slot of a shared batch, one of two dispatcher threads (one per batch, each with its own HIP stream) launches
// whatever has gathered as one multi-frame pass, and every caller copies its own result out.  A lone caller never comes here (the handle's
// own batch of one frame serves it).  CFHD_AMD_DECODE_BATCH=n (default 8) / CFHD_AMD_ENCODE_BATCH=n (default 0 = off) set the slots per batch.
template <class BatchT> struct Gatherer {
	struct Pass {
		BatchT batch;
		int claimed = 0, ready = 0, released = 0, state = 0 /* 0 collecting, 1 running, 2 done */, rc = 0;
		bool bad = false; uint32_t gen = 0;
		std::vector<void *> ptr; std::vector<int> num;      // per slot: what the pass needs from the caller (output buffer + pitch of a decode)
		std::thread worker;
	};
	int slots = 8, device = -1; bool ok = false, dead = false;      // device: the GPU the two batches live on (-1: the process default)
	std::mutex m; std::condition_variable cv_callers, cv_workers;
	Pass g[2];
	std::atomic<int> inflight{0};
	uint32_t launches = 0;
	int (*run_pass)(Pass &, int n, uint32_t launch) = nullptr;

	void start_workers()
	{
		for (Pass &x : g) { x.ptr.assign((size_t)slots, nullptr); x.num.assign((size_t)slots, 0); }
		for (int k = 0; k < 2; k++) g[k].worker = std::thread([this, k] { run(g[k]); });
		ok = true;
	}
	void run(Pass &x)
	{
		(void)device_select(device);                       // the device is selected per thread
		std::unique_lock<std::mutex> lk(m);
		for (;;) {
			cv_workers.wait(lk, [&] { return x.state == 0 && x.claimed > 0 && x.ready == x.claimed; });
			x.state = 1;
			const int n = x.claimed; const bool bad = x.bad; const uint32_t launch = ++launches;
			lk.unlock();
			const int rc = bad ? 0 : run_pass(x, n, launch);
			lk.lock();
			x.rc = rc; x.state = 2;
			cv_callers.notify_all();
		}
	}
	// stage(batch, slot, pass) != 0: this caller's input cannot go through a gathered pass (nobody of the pass is served here then);
	// finish(batch, slot) copies the caller's result out.  Returns 0 when served, 1 when the caller has to take its own path.
	template <class Stage, class Finish> int submit(Stage stage, Finish finish)
	{
		std::unique_lock<std::mutex> lk(m);
		Pass *x = nullptr;
		cv_callers.wait(lk, [&] {
			// join the pass that is gathering; else open one on a free batch
			for (Pass &c : g) if (c.state == 0 && c.claimed > 0 && c.claimed < slots) { x = &c; return true; }
			for (Pass &c : g) if (c.state == 0 && c.claimed == 0) { x = &c; return true; }
			return false;
		});
		const int i = x->claimed++; const uint32_t gen = x->gen;
		lk.unlock();
		const int staged = stage(x->batch, i, *x);            // beside the other callers: parse / copy into the slot's pinned memory
		lk.lock();
		if (staged) x->bad = true;
		x->ready++;
		cv_workers.notify_all();
		cv_callers.wait(lk, [&] { return x->state == 2 && x->gen == gen; });
		const bool failed = x->bad || x->rc != 0;
		lk.unlock();
		if (!failed) finish(x->batch, i);
		lk.lock();
		if (++x->released == x->claimed) { x->claimed = x->ready = x->released = 0; x->bad = false; x->rc = 0; x->state = 0; x->gen++; cv_callers.notify_all(); cv_workers.notify_all(); }
		return failed ? 1 : 0;
	}
};
int gather_slots(const char *env, int dflt) { const char *e = getenv(env); int v = e ? atoi(e) : dflt; return v < 0 ? 0 : (v > 64 ? 64 : v); }


// The encoder side: workers of a pool (or several pools) that encode at the same time.  Only where no frame depends on the previous one:
// qualities whose quantizer follows the size of the last sample (rate feedback, encode_one) keep one launch sequence per frame.
struct EncodeServiceKey {
	int width, height, pixel_kind, encoded_format, quality, color_space; uint32_t flags;
	int device;                                // the GPU the pool's workers were dealt (two pools on different GPUs do not share a service)
	bool operator==(const EncodeServiceKey &o) const { return memcmp(this, &o, sizeof(*this)) == 0; }
};
struct EncodeService : Gatherer<EncodeBatch> {
	EncodeServiceKey key;
	bool start(const EncodeParams &p, int nslots)
	{
		slots = nslots;
		device = key.device;
		struct OnDevice { OnDevice(int d) { device_select(d); } ~OnDevice() { device_select(-1); } } on(device);      // the shared batches live on the GPU of the workers they serve
		for (Pass &x : g) if (x.batch.prepare(p.plan, slots, true) || x.batch.prepare_entropy(sample_capacity(p))) { for (Pass &y : g) y.batch.release(); return false; }   // (a service that cannot be set up holds no HBM)
		run_pass = [](Pass &x, int n, uint32_t) {
			x.batch.set_active(n);
			int rc = x.batch.launch_forward();
			if (!rc) rc = x.batch.entropy().launch();
			if (!rc) rc = x.batch.entropy().download();
			if (!rc) rc = x.batch.wait(); else (void)x.batch.wait();
			for (int i = 0; i < n && !rc; i++) if (!x.batch.entropy().sample_bytes(i) || x.batch.entropy().needs_peak_table(i)) rc = 1;   // overflow / more peak values than the stage places: every caller takes its own path
			return rc;
		};
		start_workers();
		return true;
	}
	int encode(const SampleHeaderInfo &hdr, const void *frame, int pitch, uint8_t *out, size_t cap, size_t *size_out)
	{
		return submit([&](EncodeBatch &b, int i, Pass &) { int rc = b.upload_frame(i, frame, pitch); if (!rc) rc = b.entropy().set_frame_header(i, hdr); return rc; },
		              [&](EncodeBatch &b, int i) { const size_t n = b.entropy().sample_bytes(i); if (n <= cap) { memcpy(out, b.entropy().host_sample(i), n); *size_out = n; } else *size_out = 0; });
	}
};
struct EncodeServices {
	std::mutex m;
	std::vector<EncodeService *> list;         // never freed (see DecodeServices)
	EncodeService *find(const EncodeServiceKey &key)
	{
		std::lock_guard<std::mutex> lk(m);
		for (EncodeService *s : list) if (s->key == key) return s;
		if (list.size() >= 8) return nullptr;
		EncodeService *s = new EncodeService; s->key = key; list.push_back(s);
		return s;
	}
};
EncodeServices &encode_services() { static EncodeServices *s = new EncodeServices; return *s; }
// Pool workers gather their frames into shared passes WHILE DECODERS ARE AT WORK on the GPU, and only then.  Measured on one MI355X at 1080p (profiles/r05_e_*): a pool
// alone runs faster with every worker on its own stream (7.6 k against 6.8 k fps gathered: a shared pass keeps its callers in lock step), but a pool beside eight decoder
// threads -- the round trip through the C ABI -- runs at 2.4-2.9 k fps ungathered and 4.1-4.2 k gathered: three dozen launches per frame from sixteen threads queue up in
// the runtime, a pass of eight frames makes them a dozen.  CFHD_AMD_ENCODE_BATCH=n forces n slots whatever the decoders do (0: never gather).
std::atomic<int> g_decodes_in_flight(0);
std::atomic<long long> g_last_decode_ns(0);
long long mono_ns() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec; }
struct DecodeInFlight { DecodeInFlight() { g_decodes_in_flight.fetch_add(1); } ~DecodeInFlight() { g_last_decode_ns.store(mono_ns()); g_decodes_in_flight.fetch_sub(1); } };
bool encode_gather_forced() { static const bool f = getenv("CFHD_AMD_ENCODE_BATCH") != nullptr; return f; }
bool decoders_at_work() { return g_decodes_in_flight.load() > 0 || mono_ns() - g_last_decode_ns.load() < 5000000ll; }       // a decode call running, or one that ended within the last 5 ms
int encode_gather_slots() { return gather_slots("CFHD_AMD_ENCODE_BATCH", 8); }        // (read when a pool starts)
// true when the quantizer tables of a sequence never move: FILMSCAN1 (and anything above 1080p for LOW..HIGH) -- decided by asking the
// derivation itself whether a large previous sample would change them
bool quantizer_is_static(const EncodeParams &p)
{
	FramePlan probe = p.plan; QuantState st = p.qstate;
	st.lastgopbitcount = (int64_t)p.width * p.height * 64;               // an absurdly large previous sample
	derive_quantization(&probe, p.quality, p.progressive, 0.0f, &st);
	for (int c = 0; c < probe.num_channels; c++)
		for (int lv = 0; lv < kNumLevels; lv++)
			for (int b = 0; b < kNumBands; b++) if (probe.ch[c].band[lv][b].quant != p.plan.ch[c].band[lv][b].quant) return false;
	return true;
}

// encode_one for a caller that may share its launches with others encoding the same geometry right now
int encode_one_gathered(EncodeBatch &own, EncodeParams &p, const void *frame, int pitch, uint32_t frame_number,
                        MetaBlock global, MetaBlock local, uint8_t *out, size_t cap, size_t *size_out, EncodeService *svc)
{
	if (svc) {
		struct InFlight { std::atomic<int> &n; int before; InFlight(std::atomic<int> &c) : n(c), before(c.fetch_add(1)) {} ~InFlight() { n.fetch_sub(1); } } mark(svc->inflight);
		if (mark.before > 0 && (encode_gather_forced() || decoders_at_work())) {
			bool usable;
			{
				std::lock_guard<std::mutex> lk(svc->m);
				if (!svc->ok && !svc->dead) { if (!svc->start(p, encode_gather_slots())) svc->dead = true; }
				usable = svc->ok;
			}
			MetaBlock g2 = global, l2 = local;
			meta_remove_hidden(g2); meta_remove_hidden(l2);
			SampleHeaderInfo hdr = { frame_number, p.pixel_format == FMT_RG30 ? 122 : color_format_of(p.pixel_kind), p.color_space, p.quality, p.progressive, g2.data(), g2.size(), l2.data(), l2.size() };
			uint32_t sz; unsigned char ty;
			const uint32_t VCHN = CFHD_FOURCC('V', 'C', 'H', 'N');
			const bool vchn = meta_find(g2.data(), g2.size(), VCHN, &sz, &ty) || meta_find(l2.data(), l2.size(), VCHN, &sz, &ty);       // rare syntax switch: encode_one knows it
			if (usable && !vchn && svc->encode(hdr, frame, pitch, out, cap, size_out) == 0 && *size_out) return ERR_OKAY;
		}
		return encode_one(own, p, frame, pitch, frame_number, global, local, out, cap, size_out);
	}
	return encode_one(own, p, frame, pitch, frame_number, global, local, out, cap, size_out);
}

// Handles carry their kind in their first word: the reference's own harness closes an encoder POOL with CFHD_CloseEncoder on its error path (Example/TestCFHD.cpp:1044)
enum : uint32_t { kEncoderMagic = 0x43464845u /* 'CFHE' */, kPoolMagic = 0x43464850u /* 'CFHP' */ };
struct Encoder {
	uint32_t magic = kEncoderMagic;
	EncodeParams params;
	MetaState meta;
	EncodeBatch batch;
	bool batch_ready = false;
	GopBatch gop_batch; bool gop_ready = false; uint32_t gop_calls = 0;      // two-frame groups: calls since CFHD_PrepareToEncode
	uint32_t frame_number = 0;
	std::vector<uint8_t> sample; size_t sample_size = 0;
	StageProfile prof;
};

// ---- devices ----
// Worker i of an encoder pool / the i-th decoder handle of the process on a node with several GPUs (cfhd_core.h unit_device): -1 = the process default.
int device_of_unit(int i)
{
	const char *pinned = getenv("CFHD_AMD_DEVICE");
	if (!pinned) pinned = getenv("LOCAL_RANK");
	return unit_device(i, device_count(), pinned, getenv("CFHD_AMD_POOL_DEVICES"));
}

// ---- async pool ----
struct SampleBuffer { std::vector<uint8_t> data; size_t size = 0; };

struct PoolJob {
	uint32_t frame_number = 0;
	const void *frame = nullptr; intptr_t pitch = 0;
	MetaBlock global, local;
	std::unique_ptr<SampleBuffer> sample;
	int error = 0;
	bool finished = false;
};

struct PoolWorker {
	int device = -1;                          // the GPU this worker's batch lives on (-1: the process default)
	EncodeBatch batch;
	EncodeParams params;                      // this worker's encoder state (quantizer feedback is per encoder: each CAsyncEncoder owns an ENCODER)
	uint32_t encoded = 0;                     // per-"encoder" frame counter (the reference numbers frames per CAsyncEncoder)
	std::thread thread;
	std::deque<std::shared_ptr<PoolJob>> inbox;
};

struct EncoderPool {
	uint32_t magic = kPoolMagic;
	int nworkers = 1, queue_len = 1;
	EncodeParams params;
	MetaState meta;                           // pool-wide metadata (attached with CFHD_AttachEncoderPoolMetadata)
	std::vector<std::unique_ptr<PoolWorker>> workers;
	std::mutex m; std::condition_variable cv_work, cv_done;
	std::deque<std::shared_ptr<PoolJob>> fifo;               // submission order
	bool started = false, stopping = false;
	int next_worker = 0;
	EncodeService *service = nullptr;          // set by CFHD_StartEncoderPool when the sequence has no frame-to-frame dependency

	void worker_loop(PoolWorker *w)
	{
		device_select(w->device);                  // (per thread: everything this worker launches goes to its own GPU)
		for (;;) {
			std::shared_ptr<PoolJob> job;
			{
				std::unique_lock<std::mutex> lk(m);
				cv_work.wait(lk, [&] { return stopping || !w->inbox.empty(); });
				if (w->inbox.empty()) return;
				job = w->inbox.front(); w->inbox.pop_front();
			}
			job->sample.reset(new SampleBuffer);
			job->sample->data.resize(sample_capacity(params));
			job->error = encode_one_gathered(w->batch, w->params, job->frame, (int)job->pitch, ++w->encoded, job->global, job->local,
			                                 job->sample->data.data(), job->sample->data.size(), &job->sample->size, service);
			{
				std::lock_guard<std::mutex> lk(m);
				job->finished = true;
			}
			cv_done.notify_all();
		}
	}
	void stop()
	{
		{ std::lock_guard<std::mutex> lk(m); stopping = true; }
		cv_work.notify_all();
		for (auto &w : workers) if (w->thread.joinable()) w->thread.join();
		started = false;
	}
};

// ---- decoder ----
struct Decoder {
	int device = -1;                          // the GPU this handle decodes on (dealt round robin when the process owns several, -1: the process default)
	ParsedSample header; bool prepared = false, half = false;
	uint32_t out_format = 0; int out_kind = 0;
	FramePlan plan;
	DecodeBatch batch; bool batch_ready = false;
	uint32_t frames_decoded = 0;
	StageProfile prof;
	struct DecodeService *service = nullptr; bool service_interlaced = false;
	// samples of two-frame groups (cfhd_gop.h): the group decodes both frames, the P-frame sample behind it hands out the second one
	bool gop = false, gop_ready = false, gop_second = false; GopPlan gplan; GopBatch gop_batch;
};

struct DecMetadata { std::vector<uint8_t> block; size_t cursor = 0; };


struct DecodeServiceKey {
	int width, height, display_height, encoded_format, precision, prescale[3], out_kind, device; bool half, interlaced;
	bool operator==(const DecodeServiceKey &o) const
	{
		return device == o.device && width == o.width && height == o.height && display_height == o.display_height && encoded_format == o.encoded_format && precision == o.precision &&
		       prescale[0] == o.prescale[0] && prescale[1] == o.prescale[1] && prescale[2] == o.prescale[2] && out_kind == o.out_kind && half == o.half && interlaced == o.interlaced;
	}
};
struct DecodeService : Gatherer<DecodeBatch> {
	DecodeServiceKey key;
	bool start(const FramePlan &plan, int out_kind, bool half, bool interlaced, int nslots)
	{
		slots = nslots;
		const size_t cap = (size_t)plan.width * plan.height * pixel_bytes_of(out_kind) + 65536;
		device = key.device;
		struct OnDevice { OnDevice(int d) { device_select(d); } ~OnDevice() { device_select(-1); } } on(device);      // the batches are prepared on the service's GPU
		for (Pass &x : g) {
			x.batch.set_interlaced(interlaced);
			if (x.batch.prepare(plan, slots, out_kind, true, half) || x.batch.prepare_entropy(cap) || !x.batch.entropy().chunk_indexed()) { for (Pass &y : g) y.batch.release(); return false; }
		}
		run_pass = [](Pass &x, int n, uint32_t launch) {
			x.batch.set_active(n);
			int rc = x.batch.launch_entropy();
			if (!rc) rc = x.batch.launch_inverse(0x2545F491u * launch);
			for (int i = 0; i < n && !rc; i++) rc = x.batch.download_frame(i, x.ptr[i], x.num[i]);
			if (!rc) rc = x.batch.wait(); else (void)x.batch.wait();
			if (!rc && x.batch.entropy().check()) rc = 1;      // a damaged sample somewhere in the pass: every caller decodes alone and gets its own verdict
			return rc;
		};
		start_workers();
		return true;
	}
	int decode(const uint8_t *sample, size_t size, void *out, int pitch)
	{
		return submit([&](DecodeBatch &b, int i, Pass &x) { x.ptr[i] = out; x.num[i] = pitch; return b.entropy().set_sample_host(i, sample, size); },
		              [&](DecodeBatch &b, int i) { b.finish_frame(i, out, pitch); });       // every caller copies its own frame out of the pinned staging
	}
};
struct DecodeServices {
	std::mutex m;
	std::vector<DecodeService *> list;         // never freed: the dispatcher threads and their HIP objects live as long as the process
	DecodeService *find(const DecodeServiceKey &key)
	{
		std::lock_guard<std::mutex> lk(m);
		for (DecodeService *s : list) if (s->key == key) return s;
		if (list.size() >= 8) return nullptr;          // (a service holds two batches in HBM and pinned memory for good: a process that decodes more geometries than this at once does without)
		DecodeService *s = new DecodeService; s->key = key; list.push_back(s);
		return s;
	}
};
DecodeServices &decode_services() { static DecodeServices *s = new DecodeServices; return *s; }
int decode_gather_slots() { static const int n = gather_slots("CFHD_AMD_DECODE_BATCH", 8); return n; }

void plan_from_sample(const ParsedSample &ps, int out_kind, FramePlan *plan, bool *ok)
{
	const int quad = ps.encoded_format == ENC_BAYER ? 2 : 1;      // (build_frame_plan takes the mosaic's size)
	*ok = build_frame_plan(plan, quad * ps.width, quad * ps.display_height, out_kind, ps.encoded_format);
	if (!*ok) return;
	plan->precision = ps.precision;
	// for outputs that convert YUV to RGB: 601 or 709 by the sample's colour space tag, always the computer-systems range -- probed on the reference decoder: a
	// sample encoded with CFHD_ENCODING_FLAGS_YUV_VSRGB decodes with the CG matrix all the same (26 dB against its source instead of 31)
	plan->color_matrix = (ps.color_space & 3) == 1 ? 2 : 0;
	if (ps.prescale_table) for (int i = 0; i < kNumLevels; i++) plan->prescale[i] = (ps.prescale_table >> (14 - 2 * i)) & 3;
	else { plan->prescale[0] = 0; plan->prescale[1] = ps.precision >= 10 ? 2 : 0; plan->prescale[2] = ps.precision == 12 ? 2 : 0; }
	if (plan->height != ps.height) *ok = false;
}

} // namespace

namespace cfhd {
int front_end_params(int width, int height, uint32_t pixel_format, int encoded_format, uint32_t encoding_flags, int quality, FrontEndParams *out)
{
	EncodeParams p;
	const int rc = make_params(p, width, height, pixel_format, encoded_format, encoding_flags, quality);
	if (rc) return rc;
	out->pixel_kind = p.pixel_kind; out->encoded_format = p.encoded_format; out->pixel_bytes = pixel_bytes_of(p.pixel_kind);
	out->color_format = p.pixel_format == FMT_RG30 ? 122 : color_format_of(p.pixel_kind); out->color_space = p.color_space; out->quality = p.quality; out->progressive = p.progressive;
	out->plan = p.plan; out->static_quantizer = quantizer_is_static(p);
	return 0;
}
}

extern "C" {

int cfhd_amd_device_count(void) { return device_count(); }
void cfhd_amd_set_clip_guid(const unsigned char guid[16]) { meta_fix_guid(guid); }
const char *cfhd_amd_last_error(void) { return device_last_error(); }
// Extension: a caller that reuses its frame / output buffers may page-lock them once; CFHD_EncodeSample, the encoder pool and
// CFHD_DecodeSample then DMA between them and HBM directly instead of staging through the library's own pinned memory.  The caller
// keeps the buffer alive and unregisters it before freeing it.  Buffers that were never registered work as before.
int cfhd_amd_register_host_buffer(void *buffer, size_t bytes) { return host_buffer_register(buffer, bytes); }
int cfhd_amd_unregister_host_buffer(void *buffer) { return host_buffer_unregister(buffer); }

// =============================================================================================
// Synchronous encoder
// =============================================================================================
CFHD_Error CFHD_OpenEncoder(CFHD_EncoderRef *out, CFHD_ALLOCATOR *)
{
	if (!out) return ERR_INVALID_ARGUMENT;
	Encoder *e = new (std::nothrow) Encoder;
	if (!e) return ERR_OUTOFMEMORY;
	*out = e;
	return ERR_OKAY;
}

CFHD_Error CFHD_GetInputFormats(CFHD_EncoderRef ref, CFHD_PixelFormat *arr, int len, int *count)
{
	if (!ref || !arr) return ERR_INVALID_ARGUMENT;
	const uint32_t fmts[] = { FMT_YUY2, FMT_2VUY, FMT_RG48, FMT_B64A, FMT_BYR4, FMT_YU64, FMT_V210, FMT_RG24, FMT_BGRA, FMT_BGRa, FMT_R210, FMT_DPX0, FMT_AB10, FMT_AR10, FMT_RG30, FMT_BYR5, FMT_RG64 };
	int n = 0;
	for (; n < (int)(sizeof(fmts) / sizeof(fmts[0])) && n < len; n++) arr[n] = fmts[n];
	if (count) *count = n;
	return ERR_OKAY;
}

CFHD_Error CFHD_PrepareToEncode(CFHD_EncoderRef ref, int w, int h, CFHD_PixelFormat fmt, CFHD_EncodedFormat encoded,
                                CFHD_EncodingFlags flags, CFHD_EncodingQuality quality)
{
	CallerDevice caller_device;                      // (the caller's current HIP device is put back on the way out)
	if (!ref) return ERR_INVALID_ARGUMENT;
	Encoder *e = (Encoder *)ref;
	const int want_encoded = e->params.encoded_format == ENC_RGB444 ? 1 : (e->params.encoded_format == ENC_RGBA4444 ? 2 : (e->params.encoded_format == ENC_BAYER ? 3 : 0));
	if (e->params.valid && e->params.width == w && e->params.height == h && e->params.pixel_format == fmt && e->params.flags == (uint32_t)flags &&
	    want_encoded == (int)encoded) {
		// "just changing quality" (SampleEncoder.cpp:322-327); a different encoded format or other flags take the full path below
		e->params.quality = (int)((0xffff0000u & (uint32_t)e->params.quality) | (0xffffu & (uint32_t)quality));
		derive_quantization(&e->params.plan, e->params.quality, e->params.progressive, 0.0f, &e->params.qstate);
		e->batch_ready = false;
		// (a group encoder deals its tables on the call that opens the next group, from e->params.quality: nothing to do here)
		return ERR_OKAY;
	}
	int rc = make_params(e->params, w, h, fmt, encoded, flags, quality);
	if (rc) return rc;
	e->batch_ready = false;
	e->frame_number = 0;
	e->sample.assign(sample_capacity(e->params), 0);
	e->sample_size = 0;
	e->gop_calls = 0; e->gop_ready = false;
	return ERR_OKAY;
}

CFHD_Error CFHD_SetEncodeLicense(CFHD_EncoderRef ref, unsigned char *) { return ref ? ERR_OKAY : ERR_INVALID_ARGUMENT; }
CFHD_Error CFHD_SetEncodeLicense2(CFHD_EncoderRef ref, unsigned char *, uint32_t *level) { if (level) *level = 31; return ref ? ERR_OKAY : ERR_INVALID_ARGUMENT; }

CFHD_Error CFHD_EncodeSample(CFHD_EncoderRef ref, void *frame, int pitch)
{
	CallerDevice caller_device;                      // (the caller's current HIP device is put back on the way out)
	if (!ref || !frame) return ERR_INVALID_ARGUMENT;
	Encoder *e = (Encoder *)ref;
	if (!e->params.valid) return ERR_CODEC_ERROR;
	e->meta.handle();
	if (e->params.gop) {
		// Two frames per sample (encoder.c:3282-3380): the first call of a sequence answers with the sequence header, the call that completes a
		// pair with the group, the calls in between with the header of the group's second frame.  Frame numbers: group g carries 2 g + 1,
		// and so does the P-frame header behind it (pinned on the reference's samples).
		if (!e->gop_ready) {
			if (e->gop_batch.prepare(e->params.gplan, false, e->params.pixel_kind)) return ERR_INTERNAL;
			e->gop_ready = true;
			e->sample.assign(2 * sample_capacity(e->params), 0);
			if (gpu_entropy_enabled() && e->gop_batch.prepare_entropy(e->sample.size())) return ERR_INTERNAL;
		}
		const uint32_t n = e->gop_calls++;
		// Rate feedback (encoder.c:2880-2905): every call re-derives the subband tables from the size of the last key sample (the FILMSCAN2/3 limiter moves), the
		// call that opens a group also runs the bit-rate limiter and deals the divisors to the group's wavelets -- both frames of the group are quantized with them
		if (!(n & 1u)) {
			GopPlan next = e->params.gplan;
			if (!derive_gop_quantization(&next, e->params.quality, &e->params.gstate, 0.0f, true)) return ERR_INTERNAL;
			bool changed = next.midpoint_prequant != e->params.gplan.midpoint_prequant;
			for (int c = 0; c < 3 && !changed; c++)
				for (int k = 0; k < kGopWavelets && !changed; k++) changed = memcmp(next.ch[c].w[k].quant, e->params.gplan.ch[c].w[k].quant, sizeof(next.ch[c].w[k].quant)) != 0;
			if (changed) { e->params.gplan = next; e->gop_batch.set_plan(next); }
		} else derive_gop_quantization(&e->params.gplan, e->params.quality, &e->params.gstate, 0.0f, false);
		if (e->gop_batch.upload_frame((int)(n & 1u), frame, pitch)) return ERR_INTERNAL;
		size_t bytes;
		if (!(n & 1u)) {
			if (e->gop_batch.wait()) return ERR_INTERNAL;              // the caller's frame is borrowed for the call only: it is in pinned memory now
			bytes = n == 0 ? write_sequence_header(e->params.gplan, color_format_of(e->params.pixel_kind), e->sample.data(), e->sample.size())
			               : write_pframe_sample(e->params.gplan, n - 1, e->sample.data(), e->sample.size());
		} else {
			MetaBlock global = e->meta.global, local = e->meta.local;
			meta_remove_hidden(global); meta_remove_hidden(local);
			SampleHeaderInfo hdr = { n, color_format_of(e->params.pixel_kind), e->params.color_space, e->params.quality, !e->params.gplan.interlaced, global.data(), global.size(), local.data(), local.size() };
			if (e->gop_batch.launch_forward()) return ERR_INTERNAL;
			bytes = 0;
			// GPU entropy stage: the finished group sample comes back, not the pyramid.  The host writer takes over (from the same GPU coefficients) when the header
			// does not fit the device template block, and for a sample so large that the reference would have zeroed bands of the frame wavelets (encoder.c:8332:
			// that depends on the bytes written so far, which the device stage only knows when it is done)
			if (e->gop_batch.has_entropy() && e->gop_batch.entropy().set_frame_header(0, hdr) == 0) {
				if (e->gop_batch.entropy().launch() || e->gop_batch.entropy().download() || e->gop_batch.wait()) return ERR_INTERNAL;
				const size_t nb = e->gop_batch.entropy().sample_bytes(0);
				// (an interlaced group with more peak values in a difference-coded band than the entropy stage's positions hold: the host writer's, as for interlaced intra frames)
				if (nb && nb <= e->sample.size() && !gop_sample_may_zero_bands(e->params.gplan, nb) && !e->gop_batch.entropy().needs_peak_table(0)) { memcpy(e->sample.data(), e->gop_batch.entropy().host_sample(0), nb); bytes = nb; }
			}
			if (!bytes) {
				if (gpu_entropy_strict()) return ERR_INTERNAL;
				if (e->gop_batch.download_coeffs() || e->gop_batch.wait()) return ERR_INTERNAL;
				bytes = write_group_sample(e->params.gplan, hdr, e->gop_batch.host_coeffs(), e->sample.data(), e->sample.size());
			}
		}
		e->meta.local.clear();
		if (!bytes) return ERR_CODEC_ERROR;
		e->sample_size = bytes;
		if (n == 0 || (n & 1u)) e->params.gstate.lastgopbitcount = (int64_t)bytes * 8;      // key samples: the sequence header and the groups (encoder.c:3414)
		return ERR_OKAY;
	}
	if (!e->batch_ready) {
		e->batch.set_stage_pieces(sync_stage_pieces());
		if (prepare_batch(e->batch, e->params)) return ERR_INTERNAL;
		e->batch_ready = true;
	}
	e->prof.start();
	int rc = encode_one(e->batch, e->params, frame, pitch, ++e->frame_number, e->meta.global, e->meta.local,
	                    e->sample.data(), e->sample.size(), &e->sample_size);
	e->prof.mark(0); e->prof.calls++;
	e->meta.local.clear();                                                // FreeLocalMetadata (CFHDEncoder.cpp:351)
	return rc;
}

CFHD_Error CFHD_GetSampleData(CFHD_EncoderRef ref, void **data, size_t *size)
{
	if (!ref || !data || !size) return ERR_INVALID_ARGUMENT;
	Encoder *e = (Encoder *)ref;
	*data = e->sample.data(); *size = e->sample_size;
	return ERR_OKAY;
}

CFHD_Error CFHD_ReleaseEncoderPool(CFHD_EncoderPoolRef ref);
CFHD_Error CFHD_CloseEncoder(CFHD_EncoderRef ref)
{
	CallerDevice caller_device;                      // (the caller's current HIP device is put back on the way out)
	if (!ref) return ERR_INVALID_ARGUMENT;
	if (*(const uint32_t *)ref == kPoolMagic) return CFHD_ReleaseEncoderPool((CFHD_EncoderPoolRef)ref);      // (what the reference's harness does on its error path)
	if (*(const uint32_t *)ref != kEncoderMagic) return ERR_INVALID_ARGUMENT;
	static const char *const names[] = { "encode_one" };
	((Encoder *)ref)->prof.report("CFHD_EncodeSample", names, 1);
	delete (Encoder *)ref;
	return ERR_OKAY;
}

// =============================================================================================
// Encoder metadata
// =============================================================================================
CFHD_Error CFHD_MetadataOpen(CFHD_MetadataRef *out)
{
	if (!out) return ERR_INVALID_ARGUMENT;
	*out = new (std::nothrow) EncMetadata;
	return *out ? ERR_OKAY : ERR_OUTOFMEMORY;
}

CFHD_Error CFHD_MetadataAdd(CFHD_MetadataRef ref, uint32_t tag, CFHD_MetadataType type, size_t size, uint32_t *data, bool local)
{
	if (!ref || tag == 0 || size == 0 || !data) return ERR_INVALID_ARGUMENT;
	EncMetadata *m = (EncMetadata *)ref;
	static const char ctypes[] = { 0, 'c', 'L', 'S', 'B', 'f', 'd', 'G', 'x', 'H', 0, 'h', 0 };   // CFHDEncoderMetadata.cpp:196-236
	unsigned char ctype = (type >= 0 && type < (int)sizeof(ctypes)) ? (unsigned char)ctypes[type] : 0;
	if (!ctype) return ERR_INVALID_ARGUMENT;
	std::lock_guard<std::mutex> lk(m->lock);
	m->changed = true;
	if (m->global.empty() && tag != MTAG_CLIP_GUID && !local) { unsigned char g[16]; meta_new_guid(g); meta_add(m->global, MTAG_CLIP_GUID, 'G', 16, g); }
	return meta_add(local ? m->local : m->global, tag, ctype, (uint32_t)size, data) ? ERR_OKAY : ERR_UNEXPECTED;
}

CFHD_Error CFHD_MetadataAttach(CFHD_EncoderRef eref, CFHD_MetadataRef mref)
{
	if (!eref || !mref) return ERR_INVALID_ARGUMENT;
	Encoder *e = (Encoder *)eref; EncMetadata *m = (EncMetadata *)mref;
	std::lock_guard<std::mutex> lk(m->lock);
	if (m->changed) {
		e->meta.global = m->global;                                       // MergeMetadata (SampleEncoder.cpp:962)
		e->meta.local = m->local;
		m->local.clear();
		m->changed = false;
	}
	return ERR_OKAY;
}

CFHD_Error CFHD_MetadataClose(CFHD_MetadataRef ref)
{
	if (!ref) return ERR_INVALID_ARGUMENT;
	delete (EncMetadata *)ref;
	return ERR_OKAY;
}

// =============================================================================================
// Asynchronous encoder pool
// =============================================================================================
CFHD_Error CFHD_CreateEncoderPool(CFHD_EncoderPoolRef *out, int threads, int queue_len, CFHD_ALLOCATOR *)
{
	if (!out) return ERR_INVALID_ARGUMENT;
	EncoderPool *p = new (std::nothrow) EncoderPool;
	if (!p) return ERR_OUTOFMEMORY;
	p->nworkers = threads > 0 ? threads : 1;
	p->queue_len = queue_len > 0 ? queue_len : p->nworkers;
	*out = p;
	return ERR_OKAY;
}

CFHD_Error CFHD_GetAsyncInputFormats(CFHD_EncoderPoolRef ref, CFHD_PixelFormat *arr, int len, int *count)
{
	if (!ref) return ERR_INVALID_ARGUMENT;
	return CFHD_GetInputFormats((CFHD_EncoderRef)ref, arr, len, count);
}

CFHD_Error CFHD_PrepareEncoderPool(CFHD_EncoderPoolRef ref, uint_least16_t w, uint_least16_t h, CFHD_PixelFormat fmt,
                                   CFHD_EncodedFormat encoded, CFHD_EncodingFlags flags, CFHD_EncodingQuality quality)
{
	CallerDevice caller_device;                      // (the caller's current HIP device is put back on the way out)
	if (!ref) return ERR_INVALID_ARGUMENT;
	EncoderPool *p = (EncoderPool *)ref;
	if (p->started) {
		// A pool that is encoding takes the quality of its NEXT frames from this call and nothing else (EncoderSDK/EncoderPool.cpp:129-132 SetNextFrameQuality; the
		// reference's own harness calls this once per loop turn until its first sample comes back, Example/TestCFHD.cpp:860-897).  Frames already submitted keep the
		// tables they were submitted under: the call waits for them, then every worker takes the new tables.
		if ((int)quality == p->params.api_quality) return ERR_OKAY;
		EncodeParams np;
		const int rc = make_params(np, p->params.width, p->params.height, p->params.pixel_format, p->params.api_encoded, p->params.flags, quality);
		if (rc) return rc;
		std::unique_lock<std::mutex> lk(p->m);
		p->cv_done.wait(lk, [&] { for (auto &j : p->fifo) if (!j->finished) return false; return true; });
		p->params = np; p->service = nullptr;           // (shared passes are keyed by quality: the workers go on alone)
		for (auto &wk : p->workers) {
			device_select(wk->device);
			wk->params = np;
			const int urc = wk->batch.update_quant(np.plan);
			device_select(-1);
			if (urc) return ERR_INTERNAL;
		}
		return ERR_OKAY;
	}
	if ((uint32_t)flags & 2u) return ERR_BADFORMAT;      // two-frame groups are sequential (a frame pair per sample): the synchronous encoder serves them
	return make_params(p->params, w, h, fmt, encoded, flags, quality);
}

CFHD_Error CFHD_SetEncoderPoolLicense(CFHD_EncoderPoolRef ref, unsigned char *) { return ref ? ERR_OKAY : ERR_INVALID_ARGUMENT; }
CFHD_Error CFHD_SetEncoderPoolLicense2(CFHD_EncoderPoolRef ref, unsigned char *, uint32_t *level) { if (level) *level = 31; return ref ? ERR_OKAY : ERR_INVALID_ARGUMENT; }

CFHD_Error CFHD_AttachEncoderPoolMetadata(CFHD_EncoderPoolRef ref, CFHD_MetadataRef mref)
{
	if (!ref || !mref) return ERR_INVALID_ARGUMENT;
	EncoderPool *p = (EncoderPool *)ref; EncMetadata *m = (EncMetadata *)mref;
	std::lock_guard<std::mutex> lk(m->lock);
	std::lock_guard<std::mutex> lk2(p->m);
	p->meta.global = m->global; p->meta.local = m->local;
	return ERR_OKAY;
}

CFHD_Error CFHD_StartEncoderPool(CFHD_EncoderPoolRef ref)
{
	CallerDevice caller_device;                      // (the caller's current HIP device is put back on the way out)
	if (!ref) return ERR_INVALID_ARGUMENT;
	EncoderPool *p = (EncoderPool *)ref;
	if (!p->params.valid) return ERR_ENCODING_NOT_STARTED;
	if (p->started) return ERR_UNEXPECTED;               // (EncoderSDK/EncoderPool.cpp:187-189; the pool keeps running)
	p->stopping = false;
	p->workers.clear();
	for (int i = 0; i < p->nworkers; i++) {
		std::unique_ptr<PoolWorker> w(new PoolWorker);
		// the workers spread over the GPUs of the node round robin (one process per GPU -- CFHD_AMD_DEVICE / LOCAL_RANK set -- keeps them all on
		// its own); every worker owns its stream, tables and scratch on its device; delivery stays in submission order (the FIFO below)
		w->device = device_of_unit(i);
		device_select(w->device);
		const int prc = prepare_batch(w->batch, p->params);
		device_select(-1);
		if (prc) return ERR_INTERNAL;
		w->params = p->params;
		p->workers.push_back(std::move(w));
	}
	p->service = nullptr;
	bool one_device = true;
	for (auto &w : p->workers) if (w->batch.device() != p->workers[0]->batch.device()) one_device = false;
	if (one_device && encode_gather_slots() > 1 && p->nworkers > 1 && gpu_entropy_enabled() && quantizer_is_static(p->params)) {
		EncodeServiceKey key; memset(&key, 0, sizeof(key));
		key.width = p->params.width; key.height = p->params.height; key.pixel_kind = p->params.pixel_kind; key.encoded_format = p->params.encoded_format;
		key.quality = p->params.quality; key.color_space = p->params.color_space; key.flags = p->params.flags;
		key.device = p->workers[0]->batch.device();
		p->service = encode_services().find(key);
	}
	for (auto &w : p->workers) { PoolWorker *pw = w.get(); pw->thread = std::thread([p, pw] { p->worker_loop(pw); }); }
	p->started = true;
	return ERR_OKAY;
}

CFHD_Error CFHD_StopEncoderPool(CFHD_EncoderPoolRef ref)
{
	CallerDevice caller_device;                      // (the caller's current HIP device is put back on the way out)
	if (!ref) return ERR_INVALID_ARGUMENT;
	((EncoderPool *)ref)->stop();
	return ERR_OKAY;
}

CFHD_Error CFHD_EncodeAsyncSample(CFHD_EncoderPoolRef ref, uint32_t frame_number, void *frame, intptr_t pitch, CFHD_MetadataRef mref)
{
	CallerDevice caller_device;                      // (the caller's current HIP device is put back on the way out)
	if (!ref || !frame) return ERR_INVALID_ARGUMENT;
	EncoderPool *p = (EncoderPool *)ref;
	if (!p->started) return ERR_ENCODING_NOT_STARTED;
	std::shared_ptr<PoolJob> job(new PoolJob);
	job->frame_number = frame_number; job->frame = frame; job->pitch = pitch;    // the frame is borrowed, not copied (EncoderPool.cpp:262)
	{
		std::unique_lock<std::mutex> lk(p->m);
		// Bounded queue: block the submitter while jobQueueLength jobs are pending (MessageQueue semantics).
		p->cv_done.wait(lk, [&] { size_t pending = 0; for (auto &j : p->fifo) if (!j->finished) pending++; return pending < (size_t)p->queue_len + p->nworkers; });
		if (mref) {
			EncMetadata *m = (EncMetadata *)mref;
			std::lock_guard<std::mutex> lk2(m->lock);
			p->meta.global = m->global; p->meta.local = m->local; m->local.clear();
		}
		p->meta.handle();
		job->global = p->meta.global; job->local = p->meta.local;
		p->meta.local.clear();
		p->fifo.push_back(job);
		PoolWorker *w = p->workers[p->next_worker].get();
		p->next_worker = (p->next_worker + 1) % p->nworkers;                      // round robin on every (key) frame, EncoderPool.cpp:281-291
		w->inbox.push_back(job);
	}
	p->cv_work.notify_all();
	return ERR_OKAY;
}

static CFHD_Error pool_pop(EncoderPool *p, uint32_t *frame_number, CFHD_SampleBufferRef *out, bool wait)
{
	std::unique_lock<std::mutex> lk(p->m);
	if (p->fifo.empty()) return ERR_UNEXPECTED;
	if (!p->fifo.front()->finished) {
		if (!wait) return ERR_NOT_FINISHED;
		p->cv_done.wait(lk, [&] { return p->fifo.front()->finished; });
	}
	std::shared_ptr<PoolJob> job = p->fifo.front();
	p->fifo.pop_front();
	lk.unlock();
	p->cv_done.notify_all();
	if (frame_number) *frame_number = job->frame_number;
	if (job->error) return job->error;
	if (out) *out = job->sample.release();
	return ERR_OKAY;
}

CFHD_Error CFHD_WaitForSample(CFHD_EncoderPoolRef ref, uint32_t *frame_number, CFHD_SampleBufferRef *out)
{
	if (!ref) return ERR_INVALID_ARGUMENT;
	return pool_pop((EncoderPool *)ref, frame_number, out, true);
}

CFHD_Error CFHD_TestForSample(CFHD_EncoderPoolRef ref, uint32_t *frame_number, CFHD_SampleBufferRef *out)
{
	if (!ref) return ERR_INVALID_ARGUMENT;
	return pool_pop((EncoderPool *)ref, frame_number, out, false);
}

CFHD_Error CFHD_GetEncodedSample(CFHD_SampleBufferRef ref, void **data, size_t *size)
{
	if (!ref || !data || !size) return ERR_INVALID_ARGUMENT;
	SampleBuffer *s = (SampleBuffer *)ref;
	*data = s->data.data(); *size = s->size;
	return ERR_OKAY;
}

CFHD_Error CFHD_ReleaseSampleBuffer(CFHD_EncoderPoolRef, CFHD_SampleBufferRef ref)
{
	if (!ref) return ERR_INVALID_ARGUMENT;
	delete (SampleBuffer *)ref;
	return ERR_OKAY;
}

CFHD_Error CFHD_ReleaseEncoderPool(CFHD_EncoderPoolRef ref)
{
	CallerDevice caller_device;                      // (the caller's current HIP device is put back on the way out)
	if (!ref) return ERR_INVALID_ARGUMENT;
	EncoderPool *p = (EncoderPool *)ref;
	p->stop();
	delete p;
	return ERR_OKAY;
}

// =============================================================================================
// Decoder
// =============================================================================================
CFHD_Error CFHD_OpenDecoder(CFHD_DecoderRef *out, CFHD_ALLOCATOR *)
{
	if (!out) return ERR_INVALID_ARGUMENT;
	*out = new (std::nothrow) Decoder;
	// a process that owns several GPUs deals its decoder handles over them round robin (one process per GPU keeps them on its own)
	static std::atomic<int> handles{0};
	if (*out) ((Decoder *)*out)->device = device_of_unit(handles.fetch_add(1));
	return *out ? ERR_OKAY : ERR_OUTOFMEMORY;
}

// The output formats this decoder offers for the given sample (all of them without one): what CFHD_PrepareToDecode accepts for its encoded format.
CFHD_Error CFHD_GetOutputFormats(CFHD_DecoderRef ref, void *sample, size_t size, CFHD_PixelFormat *arr, int len, int *count)
{
	if (!ref || !arr) return ERR_INVALID_ARGUMENT;
	ParsedSample ps;
	const bool known = sample && parse_sample((const uint8_t *)sample, size, &ps) >= 0;
	uint32_t fmts[32]; int total = 0;
	auto add = [&](uint32_t f) { for (int i = 0; i < total; i++) if (fmts[i] == f) return; fmts[total++] = f; };      // (without a sample: every format once)
	if (!known || ps.encoded_format == ENC_YUV422) { add(FMT_YUY2); add(FMT_2VUY); add(FMT_YU64); add(FMT_V210); add(FMT_RG24); }
	if (!known || ps.encoded_format == ENC_RGB444) { add(FMT_RG48); add(FMT_RG24); add(FMT_BGRA); add(FMT_BGRa); add(FMT_R210); add(FMT_DPX0); add(FMT_AB10); add(FMT_AR10); add(FMT_RG30); add(FMT_B64A); }
	if (!known || ps.encoded_format == ENC_RGBA4444) { add(FMT_B64A); add(FMT_BGRA); add(FMT_BGRa); add(FMT_RG48); }
	if (!known || ps.encoded_format == ENC_BAYER) add(FMT_BYR4);
	int n = 0;
	for (; n < total && n < len; n++) arr[n] = fmts[n];
	if (count) *count = n;
	return ERR_OKAY;
}

CFHD_Error CFHD_GetSampleInfo(CFHD_DecoderRef ref, void *sample, size_t size, CFHD_SampleInfoTag tag, void *value, size_t buffer_size)
{
	if (!ref || !sample || !value || buffer_size < 4) return ERR_INVALID_ARGUMENT;
	ParsedSample ps;
	if (parse_sample((const uint8_t *)sample, size, &ps) < 0) return ERR_BADSAMPLE;
	int32_t v = 0;
	switch (tag) {
	case 0: v = 1; break;                                 // CFHD_SAMPLE_INFO_CHANNELS (video channels: 2D)
	case 1: v = ps.encoded_format == ENC_BAYER ? 2 * ps.width : ps.width; break;   // CFHD_SAMPLE_DISPLAY_WIDTH (Bayer samples carry the component plane size, decoder.c:2617)
	case 2: v = 0; break;                                 // CFHD_SAMPLE_DISPLAY_HEIGHT: the reference answers 0 for every sample (its parser keeps the tag in a local, decoder.c:2156, and SAMPLE_HEADER::display_height stays cleared); callers take the height from CFHD_PrepareToDecode
	case 3: v = 1; break;                                 // CFHD_SAMPLE_KEY_FRAME (intra only)
	case 4: v = ps.progressive; break;                    // CFHD_SAMPLE_PROGRESSIVE
	// CFHD_SAMPLE_ENCODED_FORMAT: the public CFHD_EncodedFormat enum (YUV_422 0, RGB_444 1, RGBA_4444 2, BAYER 3), not the bitstream's
	// ENCODED_FORMAT code (SampleDecoder.cpp:821-840)
	case 5: v = ps.encoded_format == ENC_RGB444 ? 1 : (ps.encoded_format == ENC_RGBA4444 ? 2 : (ps.encoded_format == ENC_BAYER ? 3 : 0)); break;
	case 6: v = (10 << 16) | (1 << 8) | 0; break;         // CFHD_SAMPLE_SDK_VERSION
	case 7: v = ((ps.version >> 12) << 16) | (((ps.version >> 8) & 0xf) << 8) | (ps.version & 0xff); break;   // CFHD_SAMPLE_ENCODE_VERSION
	default: return ERR_INVALID_ARGUMENT;
	}
	memcpy(value, &v, 4);
	return ERR_OKAY;
}

CFHD_Error CFHD_PrepareToDecode(CFHD_DecoderRef ref, int, int, CFHD_PixelFormat fmt, CFHD_DecodedResolution resolution, CFHD_DecodingFlags,
                                void *sample, size_t size, int *aw, int *ah, CFHD_PixelFormat *af)
{
	CallerDevice caller_device;                      // (the caller's current HIP device is put back on the way out)
	if (!ref || !sample) return ERR_INVALID_ARGUMENT;
	Decoder *d = (Decoder *)ref;
	{
		// a stream of two-frame groups starts with a sequence header, or is entered at a group (or at the header of a group's second frame)
		const uint8_t *s8 = (const uint8_t *)sample;
		const int first_tag = size >= 4 ? (int16_t)((s8[0] << 8) | s8[1]) : 0, sample_type = size >= 4 ? ((s8[2] << 8) | s8[3]) : 0;
		ParsedGroup pg;
		const bool group_stream = first_tag == TAG_SAMPLE && (sample_type == 7 || sample_type == 2 || sample_type == 1);
		if (group_stream) {
			// (only the header tags matter here: the caller may pass the first 512 bytes of the sample)
			(void)parse_group_sample(s8, size < 160 ? size : 160, &pg);
			if (pg.width <= 0 || pg.height <= 0) return ERR_BADSAMPLE;
			const int kind = pixel_kind_of(fmt);
			if ((kind != PIX_YUY2 && kind != PIX_2VUY) || (resolution != 1 && resolution != 0)) return ERR_BADFORMAT;      // packed 8-bit 4:2:2 at full resolution
			const int display = pg.display_height ? pg.display_height : pg.height;
			if (!build_gop_plan(&d->gplan, pg.width, display, kind)) return ERR_BADFORMAT;
			d->gop = true; d->gop_ready = false; d->gop_second = false;
			d->out_format = fmt; d->out_kind = kind; d->half = false; d->prepared = true;
			d->plan = FramePlan(); d->plan.width = pg.width; d->plan.height = d->gplan.height; d->plan.display_height = display;
			if (aw) *aw = pg.width;
			if (ah) *ah = display;
			if (af) *af = fmt;
			return ERR_OKAY;
		}
		d->gop = false;
	}
	if (parse_sample((const uint8_t *)sample, size, &d->header) < 0) return ERR_BADSAMPLE;
	// CFHD_DECODED_RESOLUTION_FULL (1; 0 = unknown is taken as full) and _HALF (2): the level-1 lowpass planes shown as the picture
	// (decoder.c:14124, :26752).  Quarter / thumbnail resolutions are not built.
	if (resolution != 1 && resolution != 0 && resolution != 2) return ERR_BAD_RESOLUTION;
	const bool half = resolution == 2;
	const int encf = d->header.encoded_format;

	if ((encf != ENC_YUV422 && encf != ENC_RGB444 && encf != ENC_RGBA4444 && encf != ENC_BAYER) || d->header.transform_type != 0) return ERR_BADFORMAT;
	int kind = pixel_kind_of(fmt);
	if (kind == PIX_NONE) return ERR_BADFORMAT;
	// 4:2:2 samples decode to the packed 4:2:2 formats, RGB 4:4:4 samples to RG48 (wavelet.c:4947), RGBA 4:4:4:4 samples to b64a
	// (bayer.c:11916 Row16uFull2OutputFormat); colour conversions between the families (ConvertLib / the colour part of the
	// active-metadata pipeline in the reference) are not built
	// ... and to YU64 (16-bit words Y0 C1 Y1 C2; the reference's planar 16-bit row route, full resolution, progressive samples)
	if (kind == PIX_YU64 && encf != ENC_YUV422) return ERR_BADFORMAT;      // (half resolution: frame.c:11146 ConvertLowpass16sToYUV64, k_half_yu64)
	// ... and RGB 4:4:4 samples to the 8-bit pixels RG24 / BGRA / BGRa (the RG48 reconstruction reduced with the reference's four-bit dither; full resolution)
	const bool rgb8 = kind == PIX_RG24 || kind == PIX_BGRA || kind == PIX_BGRa;
	// ... and RGBA 4:4:4:4 samples to BGRA / BGRa (no dither there: (12-bit component + 2) >> 4, the alpha expanded from that rounded value)
	const bool rgba8 = (kind == PIX_BGRA || kind == PIX_BGRa) && encf == ENC_RGBA4444;
	// ... and 4:2:2 samples to RG24: the YU64 rows through the reference's scalar colour conversion with its 15-bit dither (DecodeBatch / k_yu64_to_rgb24)
	const bool rgb24_of_422 = kind == PIX_RG24 && encf == ENC_YUV422 && d->header.width >= 128;      // (half resolution: frame.c:8504, k_half_rgb24)
	// ... and 4:2:2 samples to BGRA / BGRa (the reference's fused horizontal pass + 8-bit colour conversion, spatial.c:29577: k_inv_yuv422_rgb32) and to RG48 / b64a (its
	// 16-bit rows + RGB2YUV.c:1760: k_yu64_to_rgb16) -- the last four rows of TestCFHD's table; full resolution, progressive
	// (half resolution: the level-1 lowpass planes through frame.c:8504's RGB32 branch -- its SSE2 loop, so half widths that are multiples of 16 -- and frame.c:9567
	// ConvertLowpass16sYUVtoRGB48: k_half_rgb24's other modes)
	const bool rgb32_of_422 = (kind == PIX_BGRA || kind == PIX_BGRa) && encf == ENC_YUV422 && d->header.width >= 32 && (!half || (d->header.width / 2) % 16 == 0);
	const bool rgb16_of_422 = (kind == PIX_RG48 || kind == PIX_B64A) && encf == ENC_YUV422 && d->header.width >= (half ? 32 : 128);
	// (half resolution -- frame.c:7150 ConvertLowpassRGB444ToRGB -- for the outputs of RGB 4:4:4 samples: 8-bit, 10-bit, b64a; k_half_rgb)
	if (rgb8 && ((encf != ENC_RGB444 && !rgba8 && !rgb24_of_422 && !rgb32_of_422) || (half && encf != ENC_RGB444 && !rgba8 && !rgb24_of_422 && !rgb32_of_422) || d->header.width < 32)) return ERR_BADFORMAT;
	// ... and to the 10-bit RGB words r210 / DPX0 / AB10 / AR10 ((value before the final >> 1, + 3) >> 3 per component: a model fitted on the reference
	// decoder and pinned word for word on the CPU, equal to the reference decoder on the GPU)
	const bool rgb10 = kind >= PIX_R210 && kind <= PIX_AR10;
	if (rgb10 && (encf != ENC_RGB444 || d->header.width < 32)) return ERR_BADFORMAT;
	// ... and 4:2:2 samples to v210 (the YU64 words >> 6, three to a 32-bit word: DecodeBatch / k_yu64_to_v210; widths of whole six-pixel groups)
	if (kind == PIX_V210 && (encf != ENC_YUV422 || (half ? d->header.width / 2 : d->header.width) % 6 || d->header.width < 128)) return ERR_BADFORMAT;      // (half resolution: frame.c:12139 ConvertLowpass16s10bitToV210 = the half-resolution YU64 words >> 6)
	// ... and Bayer samples to BYR4: the raw mosaic, no demosaic (the four planes as 16-bit rows, recombined per quad and sent through the reference's linear-restore
	// table: DecodeBatch / k_bayer_to_byr4; full resolution)
	const bool byr4_of_bayer = kind == PIX_BYR4 && encf == ENC_BAYER && !half && d->header.width >= 32;
	if ((encf == ENC_BAYER) != byr4_of_bayer) return ERR_BADFORMAT;
	if ((kind == PIX_BYR4 && !byr4_of_bayer) || kind == PIX_BYR5 || kind == PIX_RG64 || (kind >= PIX_R210 && kind <= PIX_AR10 && !rgb10)) return ERR_BADFORMAT;     // encoder inputs only
	// ... and RGB 4:4:4 samples to b64a (the RG48 words behind a constant alpha word 0xfff0, full resolution: what TestCFHD's b64a -> RGB 4:4:4 row decodes to)
	const bool b64a_of_444 = kind == PIX_B64A && encf == ENC_RGB444;
	// ... and RGBA 4:4:4:4 samples to RG48 (the RG48 route on planes G, R, B, the alpha plane left behind; full and half resolution)
	const bool rg48_of_4444 = kind == PIX_RG48 && encf == ENC_RGBA4444;
	if ((kind == PIX_RG48 || kind == PIX_B64A) && encf == ENC_YUV422 && !rgb16_of_422) return ERR_BADFORMAT;
	if ((encf == ENC_RGB444) != ((kind == PIX_RG48 && !rg48_of_4444 && !rgb16_of_422) || (rgb8 && !rgba8 && !rgb24_of_422 && !rgb32_of_422) || rgb10 || b64a_of_444) ||
	    (encf == ENC_RGBA4444) != ((kind == PIX_B64A && !b64a_of_444 && !rgb16_of_422) || rgba8 || rg48_of_4444)) return ERR_BADFORMAT;
	if (kind == PIX_YU64 && d->header.width < 128) return ERR_BADFORMAT;      // (the tail-column rule of the 16-bit rows is restated for chroma bands of 16 columns and more)
	bool ok;
	plan_from_sample(d->header, kind, &d->plan, &ok);
	if (!ok) return ERR_BADSAMPLE;
	d->out_format = fmt; d->out_kind = kind; d->prepared = true; d->batch_ready = false; d->half = half;
	d->service = nullptr;                                 // (looked up again for the new geometry by the next concurrent decode)
	const int quad = encf == ENC_BAYER ? 2 : 1;         // (Bayer samples carry the size of their component planes)
	if (aw) *aw = half ? d->header.width / 2 : quad * d->header.width;
	if (ah) *ah = half ? d->header.display_height / 2 : quad * d->header.display_height;
	if (af) *af = fmt;
	return ERR_OKAY;
}

static int pixel_size_of(uint32_t fmt)
{
	switch (fmt) {
	case FOURCC_BE('Y', 'U', 'Y', '2'): case FOURCC_BE('2', 'v', 'u', 'y'): case FOURCC_BE('y', 'u', 'y', 'v'):
	case FOURCC_BE('B', 'Y', 'R', '2'): case FOURCC_BE('B', 'Y', 'R', '4'): return 2;
	case FOURCC_BE('R', 'G', '2', '4'): return 3;
	case FOURCC_BE('B', 'G', 'R', 'A'): case FOURCC_BE('B', 'G', 'R', 'a'): case FOURCC_BE('r', '2', '1', '0'): case FOURCC_BE('D', 'P', 'X', '0'):
	case FOURCC_BE('R', 'G', '3', '0'): case FOURCC_BE('A', 'B', '1', '0'): case FOURCC_BE('A', 'R', '1', '0'): case FOURCC_BE('Y', 'U', '6', '4'): return 4;
	case FOURCC_BE('R', 'G', '4', '8'): case FOURCC_BE('W', 'P', '1', '3'): return 6;
	case FOURCC_BE('b', '6', '4', 'a'): case FOURCC_BE('R', 'G', '6', '4'): case FOURCC_BE('W', '1', '3', 'A'): return 8;
	default: return 0;
	}
}

CFHD_Error CFHD_GetPixelSize(CFHD_PixelFormat fmt, uint32_t *out)
{
	if (!out) return ERR_INVALID_ARGUMENT;
	*out = (uint32_t)pixel_size_of(fmt);
	return ERR_OKAY;
}

CFHD_Error CFHD_GetImagePitch(uint32_t width, CFHD_PixelFormat fmt, int32_t *out)
{
	if (!out) return ERR_INVALID_ARGUMENT;
	if (fmt == FMT_V210) { *out = (int32_t)((width + 47u) / 48u * 128u); return ERR_OKAY; }      // six pixels in 16 bytes, rows of whole 48-pixel groups (as the reference answers)
	*out = (int32_t)(((width * (uint32_t)pixel_size_of(fmt)) + 15u) & ~15u);              // SampleDecoder.cpp:290-305
	return ERR_OKAY;
}

CFHD_Error CFHD_GetImageSize(uint32_t width, uint32_t height, CFHD_PixelFormat fmt, CFHD_VideoSelect videoselect, CFHD_Stereo3DType stereotype, uint32_t *out)
{
	if (!out) return ERR_INVALID_ARGUMENT;
	int32_t pitch; CFHD_GetImagePitch(width, fmt, &pitch);
	uint32_t size = (uint32_t)pitch * height;
	if (stereotype == 0 && videoselect == 3) size *= 2;
	*out = size;
	return ERR_OKAY;
}

static CFHD_Error decode_on_handle(Decoder *d, const ParsedSample &ps, const uint8_t *s, size_t size, void *out, int32_t pitch, bool interlaced);

// Samples of a stream of two-frame groups (decoder.c:11180 DecodeSampleGroup, :11426 DecodeSampleFrame): the sequence header changes nothing, the
// group is decoded whole -- run-length / VLC stage on the host, the inverse transforms on the GPU (GopBatch) -- and gives its first frame, the
// P-frame sample behind it gives the second.
static CFHD_Error decode_group_sample(Decoder *d, const uint8_t *s, size_t size, void *out, int32_t pitch)
{
	const GopPlan &gp = d->gplan;
	auto fail_zero = [&](int err) {
		const int rowbytes = packed_frame_pitch(d->out_kind, gp.width);
		for (int r = 0; r < gp.display_height; r++) memset((uint8_t *)out + (ptrdiff_t)r * pitch, 0, (size_t)rowbytes);
		return err;
	};
	ParsedGroup pg;
	const int rc = parse_group_sample(s, size, &pg);
	if (rc < 0) return fail_zero(ERR_BADSAMPLE);
	if (pg.sample_type == 7) return ERR_OKAY;                            // sequence header: no picture (the reference leaves the buffer alone too)
	if (pg.sample_type == 1) {                                           // the second frame of the last group
		if (!d->gop_second) return fail_zero(ERR_BADSAMPLE);
		d->gop_batch.finish_frame(1, out, pitch);
		d->gop_second = false;
		return ERR_OKAY;
	}
	if (pg.sample_type != 2 || pg.width != gp.width || pg.height != gp.height || pg.precision != 10) return fail_zero(ERR_BADSAMPLE);
	// Groups of interlaced frames (YUV_INTERLACED | 2FRAME_GOP) carry no SAMPLE_FLAGS tag (decoder.c:13397 sets `progressive` only from the tag): frame transform at
	// level 1 of both frames, the band 2 of both frame wavelets difference coded in code set 18 (subbands 12 and 15) -- and nowhere else
	const bool interlaced = !pg.progressive;
	for (int c = 0; c < 3; c++) for (int k = 0; k < kGopWavelets; k++) for (int b = 0; b < 4; b++)
		if (pg.band[c][k][b].present && pg.band[c][k][b].difference != (interlaced && k < 2 && b == 2)) return fail_zero(ERR_BADSAMPLE);
	if (d->gop_ready && d->gplan.interlaced != interlaced) d->gop_ready = false;
	d->gplan.interlaced = interlaced;
	if (!d->gop_ready) {
		device_select(d->device);
		const int prc = d->gop_batch.prepare(gp, true, d->out_kind);
		device_select(-1);
		if (prc) return ERR_INTERNAL;
		d->gop_ready = true;
	}
	const uint32_t dither_seed = 0x2545F491u * ++d->frames_decoded;
	// GPU entropy stage (the default): the sample goes to HBM, every coded band to one workgroup; a sample the device stage does not serve (launch < 0: geometry
	// the kernels do not take, a raw band with a divisor) is decoded below by the host coder instead
	if (gpu_entropy_enabled() && d->gop_batch.launch_entropy_decode(s, size, pg, (size_t)gp.width * gp.display_height * 8 + 131072) == 0) {
		if (d->gop_batch.launch_inverse(dither_seed, true)) return ERR_INTERNAL;
		if (d->gop_batch.download_frame(0, nullptr, 0) || d->gop_batch.download_frame(1, nullptr, 0) || d->gop_batch.wait()) return ERR_INTERNAL;
		if (d->gop_batch.entropy_decode_errors()) return fail_zero(ERR_BADSAMPLE);
		d->gop_batch.finish_frame(0, out, pitch);
		d->gop_second = true;
		return ERR_OKAY;
	}
	if (gpu_entropy_strict()) return fail_zero(ERR_INTERNAL);
	int16_t *coeffs = d->gop_batch.host_coeffs_rw();
	memset(coeffs, 0, gp.coeff_elems * 2);
	for (int c = 0; c < 3; c++) {
		const GopChannel &ch = gp.ch[c];
		const ParsedBand &lp = pg.lowpass[c];
		const GopWavelet &top = ch.w[5];
		if (!lp.present || lp.width != top.width || lp.height != top.height) return fail_zero(ERR_BADSAMPLE);
		// the bias the reference adds to the lowpass band while unpacking it: twice the intra frame's for a group (decoder.c:12265 `num_frames == 2 ? 48 : 24`)
		const int bias = 2 * lowpass_bias(10, top.width, d->out_kind);
		for (int r = 0; r < top.height; r++) {
			const uint8_t *p = s + lp.offset + (size_t)r * top.width * 2;
			int16_t *dst = coeffs + top.offset[0] + (size_t)r * top.pitch;
			// (a band of odd width is read 16 unsigned bits at a time, one of even width as pairs of signed words: decoder.c:12240-12290, as k_dec_lowpass does)
			for (int x = 0; x < top.width; x++) { int v = (int16_t)((p[2 * x] << 8) | p[2 * x + 1]); if (top.width & 1) v = (int)(uint16_t)v; v += bias; dst[x] = (int16_t)(v > 0x7fff ? 0x7fff : v); }
		}
		static const int coded[5] = { 5, 4, 3, 1, 0 };
		for (int k : coded) {
			const GopWavelet &wv = ch.w[k];
			for (int b = (k == 3 ? 0 : 1); b < 4; b++) {
				const ParsedBand &pb = pg.band[c][k][b];
				if (!pb.present || pb.width != wv.width || pb.height != wv.height) return fail_zero(ERR_BADSAMPLE);
				int16_t *dst = coeffs + wv.offset[b];
				if (pb.codebook < 0) {                                       // raw 16-bit words (the lowpass band of the temporal highpass wavelet)
					if ((size_t)pb.bytes < (size_t)wv.width * wv.height * 2) return fail_zero(ERR_BADSAMPLE);
					for (int r = 0; r < wv.height; r++) {
						const uint8_t *p = s + pb.offset + (size_t)r * wv.width * 2;
						for (int x = 0; x < wv.width; x++) dst[(size_t)r * wv.pitch + x] = (int16_t)(((p[2 * x] << 8) | p[2 * x + 1]) * pb.quant);
					}
				} else if (vlc_decode_band(s + pb.offset, pb.bytes, wv.width, wv.height, wv.pitch, pb.quant, pb.codebook, dst)) return fail_zero(ERR_BADSAMPLE);
				// interlaced groups: peak values, then every row becomes its running sum (decoder.c:19809, :20822)
				if (pb.difference) finish_difference_band(dst, wv.width, wv.height, wv.pitch, pb.peak_level ? s + pb.peak_offset : nullptr, pb.peak_level ? size - pb.peak_offset : 0, pb.peak_level);
			}
		}
	}
	if (d->gop_batch.launch_inverse(dither_seed)) return ERR_INTERNAL;
	if (d->gop_batch.download_frame(0, nullptr, 0) || d->gop_batch.download_frame(1, nullptr, 0) || d->gop_batch.wait()) return ERR_INTERNAL;
	d->gop_batch.finish_frame(0, out, pitch);
	d->gop_second = true;
	return ERR_OKAY;
}

CFHD_Error CFHD_DecodeSample(CFHD_DecoderRef ref, void *sample, size_t size, void *out, int32_t pitch)
{
	CallerDevice caller_device;                      // (the caller's current HIP device is put back on the way out)
	DecodeInFlight decode_in_flight;                 // (encoder pools of the process gather their frames while decoders are at work: encode_one_gathered)
	if (!ref || !sample || !out) return ERR_INVALID_ARGUMENT;
	Decoder *d = (Decoder *)ref;
	if (!d->prepared) return ERR_UNEXPECTED;
	const uint8_t *s = (const uint8_t *)sample;
	if (d->gop) return decode_group_sample(d, s, size, out, pitch);
	ParsedSample ps;
	auto fail_zero = [&](int err) {                                               // decode failure zero-fills the output (decoder.c:11850-11859)
		const int quad = d->plan.encoded_format == ENC_BAYER ? 2 : 1;      // (the plan of a Bayer sample counts photosite quads)
		const int rowbytes = packed_frame_pitch(d->out_kind, d->half ? d->plan.width / 2 : quad * d->plan.width), rows = d->half ? d->plan.display_height / 2 : quad * d->plan.display_height;
		for (int r = 0; r < rows; r++) memset((uint8_t *)out + (ptrdiff_t)r * pitch, 0, (size_t)rowbytes);
		return err;
	};
	if (parse_sample(s, size, &ps) != 0) return fail_zero(ERR_BADSAMPLE);
	if (ps.width != d->header.width || ps.display_height != d->header.display_height || ps.encoded_format != d->header.encoded_format ||
	    ps.num_channels != d->plan.num_channels) return fail_zero(ERR_BADSAMPLE);
	// interlaced samples (known only now: the SAMPLE_FLAGS tag lies behind the 512 bytes CFHD_PrepareToDecode sees): 4:2:2, full resolution through the
	// inverse frame transform, half resolution from the level-1 lowpass planes like any other sample (the reference's output is the same model)
	const bool interlaced = !ps.progressive;
	// (YU64 / v210 output of interlaced samples: at half resolution only -- the level-1 lowpass planes, as for progressive samples; RG24 takes another route there: not built)
	// (likewise the 8-bit / 16-bit RGB(A) pictures and the 10-bit RGB words of an interlaced sample: half resolution only -- at full resolution DecodeBatch::launch_inverse has no
	// inverse frame transform into planes for them, and refusing here keeps the contract of every other unsupported combination: BADFORMAT, zeroed picture, nothing queued)
	const bool planes_out = d->out_kind == PIX_YU64 || d->out_kind == PIX_V210 || d->out_kind == PIX_BGRA || d->out_kind == PIX_BGRa || d->out_kind == PIX_RG48 || d->out_kind == PIX_B64A ||
	                        (d->out_kind >= PIX_R210 && d->out_kind <= PIX_AR10);
	if (interlaced && (ps.encoded_format != ENC_YUV422 || (planes_out && !d->half) || d->out_kind == PIX_RG24)) return fail_zero(ERR_BADFORMAT);
	if (interlaced && !d->half && ps.width > 8192) return fail_zero(ERR_BADFORMAT);        // k_dec_undiff serves rows of up to 4096 coefficients (cfhd_dec_kernels.h DXU_MAX): an unsupported size, not a bad sample
	// another call of this geometry in flight right now: decode together with it (see DecodeService)
	if (decode_gather_slots() > 1 && gpu_entropy_enabled() && size <= (size_t)d->plan.width * d->plan.height * pixel_bytes_of(d->out_kind) + 65536) {
		if (!d->service || d->service_interlaced != interlaced) {
			DecodeServiceKey key; memset(&key, 0, sizeof(key));
			key.width = d->plan.width; key.height = d->plan.height; key.display_height = d->plan.display_height; key.encoded_format = d->plan.encoded_format;
			key.precision = d->plan.precision; for (int k = 0; k < 3; k++) key.prescale[k] = d->plan.prescale[k];
			key.out_kind = d->out_kind; key.half = d->half; key.interlaced = interlaced; key.device = d->device;
			d->service = decode_services().find(key); d->service_interlaced = interlaced;
		}
		DecodeService *svc = d->service;
		if (!svc) return decode_on_handle(d, ps, s, size, out, pitch, interlaced);
		struct InFlight { std::atomic<int> &n; int before; InFlight(std::atomic<int> &c) : n(c), before(c.fetch_add(1)) {} ~InFlight() { n.fetch_sub(1); } } mark(svc->inflight);
		if (mark.before > 0) {
			bool usable;
			{
				std::lock_guard<std::mutex> lk(svc->m);
				if (!svc->ok && !svc->dead) { if (!svc->start(d->plan, d->out_kind, d->half, interlaced, decode_gather_slots())) svc->dead = true; }   // (could not be set up: never tried again)
				usable = svc->ok;
			}
			if (usable && svc->decode(s, size, out, pitch) == 0) return ERR_OKAY;
			// not decoded there (damaged sample in the pass, device trouble): this handle's own path gives this sample its own verdict
		}
		return decode_on_handle(d, ps, s, size, out, pitch, interlaced);
	}
	return decode_on_handle(d, ps, s, size, out, pitch, interlaced);
}

// One sample on the handle's own batch of one frame (the only path of a caller that decodes alone).
static CFHD_Error decode_on_handle(Decoder *d, const ParsedSample &ps, const uint8_t *s, size_t size, void *out, int32_t pitch, bool interlaced)
{
	auto fail_zero = [&](int err) {                                               // decode failure zero-fills the output (decoder.c:11850-11859)
		const int quad = d->plan.encoded_format == ENC_BAYER ? 2 : 1;      // (the plan of a Bayer sample counts photosite quads)
		const int rowbytes = packed_frame_pitch(d->out_kind, d->half ? d->plan.width / 2 : quad * d->plan.width), rows = d->half ? d->plan.display_height / 2 : quad * d->plan.display_height;
		for (int r = 0; r < rows; r++) memset((uint8_t *)out + (ptrdiff_t)r * pitch, 0, (size_t)rowbytes);
		return err;
	};
	if (d->batch_ready && d->batch.interlaced() != interlaced) d->batch_ready = false;
	if (!d->batch_ready) {
		d->batch.set_interlaced(interlaced);
		device_select(d->device);                      // the handle's GPU (the batch remembers it: later calls may come from any thread)
		d->batch.set_stage_pieces(sync_stage_pieces());
		int prc = d->batch.prepare(d->plan, 1, d->out_kind, true, d->half);
		if (!prc && gpu_entropy_enabled()) prc = d->batch.prepare_entropy((size_t)d->plan.width * d->plan.height * pixel_bytes_of(d->out_kind) + 65536);
		device_select(-1);
		if (prc) return ERR_INTERNAL;
		d->batch_ready = true;
	}
	if (d->batch.has_entropy() && size <= (size_t)d->plan.width * d->plan.height * pixel_bytes_of(d->out_kind) + 65536) {
		// GPU entropy decoder: ship the sample bytes, the pyramid is rebuilt in HBM
		d->prof.start();
		if (d->batch.entropy().set_sample_host(0, s, size)) return fail_zero(ERR_BADSAMPLE);
		d->prof.mark(0);
		if (d->batch.launch_entropy()) return ERR_INTERNAL;
		if (d->batch.launch_inverse(0x2545F491u * ++d->frames_decoded)) return ERR_INTERNAL;
		if (d->batch.download_frame(0, out, pitch)) return ERR_INTERNAL;
		d->prof.mark(1);
		// the picture is copied out piece by piece behind the DMA of each piece (finish_frame waits on their events); whether the sample decoded cleanly is known
		// behind it -- a damaged one has its output zeroed as before
		if (d->batch.staged_in_pieces()) {
			if (d->batch.finish_frame(0, out, pitch)) return ERR_INTERNAL;
			d->prof.mark(2);
			if (d->batch.wait()) return ERR_INTERNAL;
			if (d->batch.entropy().check()) return fail_zero(ERR_BADSAMPLE);
		} else {
			if (d->batch.wait()) return ERR_INTERNAL;
			d->prof.mark(2);
			if (d->batch.entropy().check()) return fail_zero(ERR_BADSAMPLE);
			d->batch.finish_frame(0, out, pitch);
		}
		d->prof.mark(3); d->prof.calls++;
		return ERR_OKAY;
	}
	// Entropy decode on the host into the pinned coefficient staging (dequantized values, as the reference's FSM delivers them).
	d->batch.clear_host_coeffs(0);
	int16_t *coeffs = d->batch.host_coeffs(0);
	const FramePlan &plan = d->plan;
	// The reference biases the lowpass band while unpacking it (Codec/decoder.c:12240-12290 "channeloffset"):
	// see lowpass_bias().
	for (int c = 0; c < plan.num_channels; c++) {
		const ParsedBand &lp = ps.lowpass[c];
		const BandDesc &ll = plan.ch[c].band[2][0];
		if (!lp.present || lp.width != ll.width || lp.height != ll.height) return fail_zero(ERR_BADSAMPLE);
		const int lowpass_offset = lowpass_bias(plan.precision, ll.width, d->out_kind, c);
		for (int r = 0; r < ll.height; r++) {
			const uint8_t *p = s + lp.offset + (size_t)r * ll.width * 2;
			int16_t *dst = coeffs + ll.offset + (size_t)r * ll.pitch;
			for (int x = 0; x < ll.width; x++) {
				int v = (int16_t)((p[2 * x] << 8) | p[2 * x + 1]);
				if (ll.width & 1) v = (int)(uint16_t)v;           // (odd width: 16 unsigned bits at a time, decoder.c:12240-12290; the same rule as k_dec_lowpass)
				v += lowpass_offset;
				dst[x] = (int16_t)(v > 0x7fff ? 0x7fff : v);
			}
		}
		for (int lv = 0; lv < kNumLevels; lv++)
			for (int b = 1; b < 4; b++) {
				const ParsedBand &pb = ps.high[c][lv][b];
				const BandDesc &bd = plan.ch[c].band[lv][b];
				if (!pb.present || pb.width != bd.width || pb.height != bd.height) return fail_zero(ERR_BADSAMPLE);
				if (vlc_decode_band(s + pb.offset, pb.bytes, bd.width, bd.height, bd.pitch, pb.quant, pb.codebook, coeffs + bd.offset)) return fail_zero(ERR_BADSAMPLE);
				if (pb.difference) {
					if (pb.peak_level && (size_t)pb.peak_offset + 2 > size) return fail_zero(ERR_BADSAMPLE);
					finish_difference_band(coeffs + bd.offset, bd.width, bd.height, bd.pitch, pb.peak_level ? s + pb.peak_offset : nullptr, pb.peak_level ? size - pb.peak_offset : 0, pb.peak_level);
				}
			}
	}
	if (d->batch.upload_coeffs()) return ERR_INTERNAL;
	if (d->batch.launch_inverse(0x2545F491u * ++d->frames_decoded)) return ERR_INTERNAL;
	if (d->batch.download_frame(0, out, pitch)) return ERR_INTERNAL;
	if (d->batch.wait()) return ERR_INTERNAL;
	d->batch.finish_frame(0, out, pitch);
	return ERR_OKAY;
}

CFHD_Error CFHD_SetLicense(CFHD_DecoderRef ref, const unsigned char *) { return ref ? ERR_OKAY : ERR_INVALID_ARGUMENT; }
CFHD_Error CFHD_SetActiveMetadata(CFHD_DecoderRef ref, CFHD_MetadataRef, unsigned int, CFHD_MetadataType, void *, unsigned int)
{
	// Active-metadata image development (colour, 3D, burn-ins) is outside the hot path; the tags are accepted and ignored.
	return ref ? ERR_OKAY : ERR_INVALID_ARGUMENT;
}
CFHD_Error CFHD_ClearActiveMetadata(CFHD_DecoderRef ref, CFHD_MetadataRef) { return ref ? ERR_OKAY : ERR_INVALID_ARGUMENT; }
// 1/8 x 1/8 thumbnail straight from the raw lowpass bands of the sample, no decode and no GPU involved (host code as in the reference:
// Codec/thumbnail.c:65 GenerateThumbnail).  Output: 10-bit RGB, one big-endian dword per pixel, r << 22 | g << 12 | b << 2 ("DPX0").
//   4:2:2: per pixel pair, y = (lowpass >> 4 & 0x3ff) - 64, Cr / Cb = (lowpass >> 4 & 0x3ff) - 512 from channels 1 / 2, then the fixed-point
//          709 matrix of thumbnail.c:205-222;  4:4:4(:4): the G, R, B lowpass values >> 4.  Bayer samples are not built.
static int thumbnail_from_sample(const void *sample, size_t size, void *out, size_t out_size, size_t *rw, size_t *rh, size_t *rsize)
{
	if (!sample) return ERR_INVALID_ARGUMENT;
	ParsedSample ps;
	if (parse_sample((const uint8_t *)sample, size, &ps) != 0) return ERR_BADSAMPLE;
	const int enc = ps.encoded_format;
	if (enc != ENC_YUV422 && enc != ENC_RGB444 && enc != ENC_RGBA4444) return ERR_BADFORMAT;
	const int w = (ps.width + 7) / 8, h = (ps.height + 7) / 8;
	for (int c = 0; c < 3; c++) {
		const ParsedBand &lp = ps.lowpass[c];
		const int cw = (enc == ENC_YUV422 && c) ? w / 2 : w;
		if (!lp.present || lp.width != cw || lp.height != h || (size_t)lp.offset + (size_t)cw * h * 2 > size) return ERR_BADSAMPLE;
	}
	if (!out) {                                          // size query (thumbnail.c:30 GetThumbnailInfo)
		if (rw) *rw = (size_t)w;
		if (rh) *rh = (size_t)h;
		if (rsize) *rsize = (size_t)w * h * 4;
		return ERR_OKAY;
	}
	if ((w & 1) || out_size < (size_t)w * h * 4) return ERR_INVALID_ARGUMENT;
	const uint8_t *s = (const uint8_t *)sample;
	auto be16 = [&](const ParsedBand &b, size_t i) { const uint8_t *p = s + b.offset + 2 * i; return (int)((p[0] << 8) | p[1]); };
	auto clamp10 = [](int v) { return v < 0 ? 0 : (v > 0x3ff ? 0x3ff : v); };
	auto put = [&](size_t i, int r, int g, int b) {
		const uint32_t rgb = ((uint32_t)r << 22) | ((uint32_t)g << 12) | ((uint32_t)b << 2);
		uint8_t *o = (uint8_t *)out + 4 * i;
		o[0] = (uint8_t)(rgb >> 24); o[1] = (uint8_t)(rgb >> 16); o[2] = (uint8_t)(rgb >> 8); o[3] = (uint8_t)rgb;
	};
	const size_t n = (size_t)w * h;
	if (enc == ENC_YUV422) {
		for (size_t i = 0; i < n; i += 2) {
			const int cr = ((be16(ps.lowpass[1], i / 2) >> 4) & 0x3ff) - 0x200, cb = ((be16(ps.lowpass[2], i / 2) >> 4) & 0x3ff) - 0x200;
			for (int k = 0; k < 2; k++) {
				const int y = ((be16(ps.lowpass[0], i + k) >> 4) & 0x3ff) - 64;
				put(i + k, clamp10((1192 * y + 1836 * cr) >> 10), clamp10((1192 * y - 547 * cr - 218 * cb) >> 10), clamp10((1192 * y + 2166 * cb) >> 10));
			}
		}
	} else {
		for (size_t i = 0; i < n; i++)
			put(i, (be16(ps.lowpass[1], i) >> 4) & 0x3ff, (be16(ps.lowpass[0], i) >> 4) & 0x3ff, (be16(ps.lowpass[2], i) >> 4) & 0x3ff);
	}
	if (rw) *rw = (size_t)w;
	if (rh) *rh = (size_t)h;
	if (rsize) *rsize = n * 4;
	return ERR_OKAY;
}

CFHD_Error CFHD_GetThumbnail(CFHD_DecoderRef ref, void *sample, size_t size, void *out, size_t out_size, uint32_t, size_t *rw, size_t *rh, size_t *rsize)
{
	if (!ref || !sample || !out) return ERR_INVALID_ARGUMENT;
	return thumbnail_from_sample(sample, size, out, out_size, rw, rh, rsize);
}

// EncoderSDK/CFHDEncoder.cpp:593: the same thumbnail, asked of an encoder for a sample it produced.
CFHD_Error CFHD_GetEncodeThumbnail(CFHD_EncoderRef ref, void *sample, size_t size, void *out, size_t out_size, uint32_t, size_t *rw, size_t *rh, size_t *rsize)
{
	if (!ref || !sample || !out) return ERR_INVALID_ARGUMENT;
	return thumbnail_from_sample(sample, size, out, out_size, rw, rh, rsize);
}

// EncoderSDK/CFHDEncoderPool.cpp:620: thumbnail of a sample buffer of the encoder pool; without an output buffer only the dimensions.
CFHD_Error CFHD_GetSampleThumbnail(CFHD_SampleBufferRef ref, void *out, size_t out_size, uint32_t, uint_least16_t *rw, uint_least16_t *rh,
                                   CFHD_PixelFormat *fmt, size_t *rsize)
{
	if (!ref) return ERR_INVALID_ARGUMENT;
	SampleBuffer *sb = (SampleBuffer *)ref;
	size_t w = 0, h = 0, n = 0;
	const int rc = thumbnail_from_sample(sb->data.data(), sb->size, out_size ? out : nullptr, out_size, &w, &h, &n);
	if (rc != ERR_OKAY) return ERR_CODEC_ERROR;
	if (rw) *rw = (uint_least16_t)w;
	if (rh) *rh = (uint_least16_t)h;
	if (fmt) *fmt = FOURCC_BE('D', 'P', 'X', '0');
	if (rsize) *rsize = n;
	return ERR_OKAY;
}

// DecoderSDK/CFHDDecoder.cpp:443 (obsoleted there by CFHD_GetSampleInfo): encoded format, field type and frame size of a sample.  The
// reference's CFHD_SampleHeader is a class of exactly these four ints (Common/CFHDSampleHeader.h:32).
CFHD_Error CFHD_ParseSampleHeader(void *sample, size_t size, CFHD_SampleHeader *hdr)
{
	if (!sample || !hdr) return ERR_INVALID_ARGUMENT;
	ParsedSample ps;
	if (parse_sample((const uint8_t *)sample, size, &ps) < 0) return ERR_BADSAMPLE;
	hdr->encoded_format = ps.encoded_format == ENC_RGB444 ? 1 : (ps.encoded_format == ENC_RGBA4444 ? 2 : (ps.encoded_format == ENC_BAYER ? 3 : 0));
	// CSampleDecoder::FieldType (SampleDecoder.cpp:1886): 1 progressive; interlaced: 2 upper field first unless the optional interlaced
	// flags say interlaced (bit 0) without field-1-first (bit 1), then 3 lower field first
	if (ps.progressive || ps.encoded_format == ENC_BAYER) hdr->field_type = 1;
	else hdr->field_type = ((ps.interlaced_flags & 1) && !(ps.interlaced_flags & 2)) ? 3 : 2;
	hdr->width = ps.width; hdr->height = ps.display_height;
	return ERR_OKAY;
}

CFHD_Error CFHD_CloseDecoder(CFHD_DecoderRef ref)
{
	CallerDevice caller_device;                      // (the caller's current HIP device is put back on the way out)
	if (!ref) return ERR_INVALID_ARGUMENT;
	static const char *const names[] = { "parse+stage", "submit", "gpu+copies", "copy out" };
	((Decoder *)ref)->prof.report("CFHD_DecodeSample", names, 4);
	delete (Decoder *)ref;
	return ERR_OKAY;
}

// =============================================================================================
// Decoder-side metadata access
// =============================================================================================
CFHD_Error CFHD_OpenMetadata(CFHD_MetadataRef *out)
{
	if (!out) return ERR_INVALID_ARGUMENT;
	*out = new (std::nothrow) DecMetadata;
	return *out ? ERR_OKAY : ERR_OUTOFMEMORY;
}

CFHD_Error CFHD_InitSampleMetadata(CFHD_MetadataRef ref, CFHD_MetadataTrack, void *sample, size_t size)
{
	if (!ref || !sample) return ERR_INVALID_ARGUMENT;
	DecMetadata *m = (DecMetadata *)ref;
	ParsedSample ps;
	m->block.clear(); m->cursor = 0;
	if (parse_sample((const uint8_t *)sample, size, &ps) < 0) return ERR_BADSAMPLE;
	if (ps.metadata_bytes && ps.metadata_offset + ps.metadata_bytes <= size)
		m->block.assign((const uint8_t *)sample + ps.metadata_offset, (const uint8_t *)sample + ps.metadata_offset + ps.metadata_bytes);
	return ERR_OKAY;
}

static int api_type_of(unsigned char c)
{
	switch (c) { case 'c': return 1; case 'L': return 2; case 'S': return 3; case 'B': return 4; case 'f': return 5; case 'd': return 6;
	case 'G': return 7; case 'x': return 8; case 'H': return 9; case 'h': return 11; default: return 0; }
}

CFHD_Error CFHD_ReadMetadata(CFHD_MetadataRef ref, unsigned int *tag, CFHD_MetadataType *type, void **data, CFHD_MetadataSize *size)
{
	if (!ref || !tag || !type || !data || !size) return ERR_INVALID_ARGUMENT;
	DecMetadata *m = (DecMetadata *)ref;
	if (m->cursor + 8 > m->block.size()) return ERR_METADATA_END;
	uint32_t t, ts; memcpy(&t, &m->block[m->cursor], 4); memcpy(&ts, &m->block[m->cursor + 4], 4);
	uint32_t len = ts & 0xffffff;
	if (t == 0 || m->cursor + 8 + len > m->block.size()) return ERR_METADATA_END;
	*tag = t; *type = api_type_of((unsigned char)(ts >> 24)); *data = &m->block[m->cursor + 8]; *size = (CFHD_MetadataSize)len;
	m->cursor += 8 + ((len + 3) & ~3u);
	return ERR_OKAY;
}

CFHD_Error CFHD_FindMetadata(CFHD_MetadataRef ref, unsigned int tag, CFHD_MetadataType *type, void **data, CFHD_MetadataSize *size)
{
	if (!ref || !type || !data || !size) return ERR_INVALID_ARGUMENT;
	DecMetadata *m = (DecMetadata *)ref;
	uint32_t len; unsigned char ty;
	const uint8_t *p = meta_find(m->block.data(), m->block.size(), tag, &len, &ty);
	if (!p) return ERR_METADATA_END;
	*type = api_type_of(ty); *data = (void *)p; *size = (CFHD_MetadataSize)len;
	return ERR_OKAY;
}

CFHD_Error CFHD_CloseMetadata(CFHD_MetadataRef ref)
{
	if (!ref) return ERR_INVALID_ARGUMENT;
	delete (DecMetadata *)ref;
	return ERR_OKAY;
}

} // extern "C"
